#!/usr/bin/env python
"""Warm vs cold (rotating > 256 MiB of feature buffers) timing of the fused scorer+loss kernel:
python scripts/probe_cold.py [--workload c2] [--nbuf 5] [--kinds hinge,ndcg2]

Warm = the same batch every launch (71 MB at C2: sits in the 256 MiB Infinity Cache);
cold = NBUF distinct batches in rotation, so every launch streams its features from HBM."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import WORKLOADS, synth  # noqa: E402
from pytorchltr_amd import _C  # noqa: E402


def graph_time(fns, replays=10):
    """us per launch of the launches in `fns`, captured back-to-back in one hipGraph."""
    def many():
        for f in fns:
            f()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        many()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        many()
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(replays):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / (len(fns) * replays)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c2")
    ap.add_argument("--nbuf", type=int, default=5)
    ap.add_argument("--kinds", default="")
    ap.add_argument("--B", type=int, default=0)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    B, L, F, kind0 = WORKLOADS[args.workload]
    if args.B:
        B = args.B
    kinds = args.kinds.split(",") if args.kinds else [kind0]
    lib = _C.lib()
    W = (torch.rand(F, device=dev) * 2 - 1) / F ** 0.5
    bias = torch.zeros(1, device=dev)
    lossv = torch.empty(B, device=dev)
    part = torch.empty(lib.ltr_linear_workspace_bytes(B, L, F) // 4 + 64, device=dev)
    dWb = torch.zeros(F + 3, device=dev)
    for full in (False, True):
        bufs = []
        for i in range(args.nbuf):
            _, rel, n, X = synth(B, L, F, 100 + i, dev)
            if full:
                n = torch.full_like(n, L)
            bufs.append((X, rel, n))
        for kind in kinds:
            k = getattr(_C, kind.upper())

            def mk(buf, with_reduce):
                X, rel, n = buf

                def f():
                    st = torch.cuda.current_stream().cuda_stream
                    _C.check(lib.ltr_linear_partials_f32(k, 1.0, X.data_ptr(), W.data_ptr(), bias.data_ptr(),
                                                         rel.data_ptr(), 0, n.data_ptr(), B, L, F, lossv.data_ptr(),
                                                         None, part.data_ptr(), st))
                    if with_reduce:
                        _C.check(lib.ltr_linear_reduce_loss_f32(part.data_ptr(), None, lossv.data_ptr(), B, F,
                                                                dWb.data_ptr(), dWb.data_ptr() + 4 * F,
                                                                dWb.data_ptr() + 4 * (F + 1), st))
                return f
            res = {"workload": args.workload, "B": B, "kind": kind, "full_lists": full}
            for with_reduce in (False, True):
                warm = graph_time([mk(bufs[0], with_reduce)] * 20)
                cold = graph_time([mk(bufs[i % args.nbuf], with_reduce) for i in range(4 * args.nbuf)])
                key = "step" if with_reduce else "kernel"
                res[key + "_warm_us"] = round(warm, 2)
                res[key + "_cold_us"] = round(cold, 2)
            rows = sum(int(b[2].clamp(max=L).sum()) for b in bufs) / args.nbuf
            moved = rows * (4 * F + 8) + B * 8 + 4 * (F + 1) + B * 4 + 4 * (F + 1) * B
            padded = B * (4 * L * F + 8 * L + 8 + 4 + 4 * (F + 1)) + 4 * (F + 1)
            res["moved_MB"] = round(moved / 1e6, 2)
            res["padded_MB"] = round(padded / 1e6, 2)
            res["frac_moved_cold"] = round(moved / (res["kernel_cold_us"] * 1e-6) / 8e12, 3)
            res["frac_padded_cold"] = round(padded / (res["kernel_cold_us"] * 1e-6) / 8e12, 3)
            print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
