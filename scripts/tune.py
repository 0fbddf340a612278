#!/usr/bin/env python
"""Time kernel variants on the GPU: python scripts/tune.py build/variants/*.so [--workload c2]"""
import argparse
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import WORKLOADS, synth  # noqa: E402
from _benchutil import time_events, time_launches  # noqa: E402
from pytorchltr_amd import _C  # noqa: E402

KIND = {"hinge": 0, "dcg_hinge": 1, "logistic": 2, "arp1": 3, "arp2": 4, "ndcg1": 5, "ndcg2": 6}


def load(path):
    h = ctypes.CDLL(path)
    for name, (restype, argtypes) in _C.SIGNATURES.items():
        fn = getattr(h, name)
        fn.restype = restype
        fn.argtypes = argtypes
    return h


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--workload", default="c2")
    ap.add_argument("--full", action="store_true")
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--loss-cfgs", default="")
    ap.add_argument("--trace", action="store_true", help="libs built with -DLTR_TRACE: per-phase cycle stamps")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    B, L, F, kind = WORKLOADS[args.workload]
    scores, rel, n, X = synth(B, L, F, 0, dev)
    if args.full:
        n = torch.full_like(n, L)
    W = torch.randn(F, device=dev) * 0.1
    bias = torch.zeros(1, device=dev)
    loss = torch.empty(B, device=dev)
    ds = torch.empty(B, L, device=dev)
    dW = torch.empty(F, device=dev)
    db = torch.empty(1, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    k = KIND[kind]
    ref = None
    trbuf = torch.zeros(B * 8, dtype=torch.int64, device=dev) if args.trace else None
    so_ptr = trbuf.data_ptr() if args.trace else None
    for path in args.libs:
        lib = load(path)
        part = torch.empty(lib.ltr_linear_workspace_bytes(B, L, F) // 4 + 64, device=dev)

        def cs():
            return torch.cuda.current_stream().cuda_stream

        def fused():
            rc = lib.ltr_linear_partials_f32(k, 1.0, X.data_ptr(), W.data_ptr(), bias.data_ptr(),
                                             rel.data_ptr(), 0, n.data_ptr(), B, L, F, loss.data_ptr(),
                                             so_ptr, part.data_ptr(), cs())
            assert rc == 0, rc

        def reduce():
            rc = lib.ltr_linear_reduce_f32(part.data_ptr(), None, B, F, dW.data_ptr(), db.data_ptr(), cs())
            assert rc == 0, rc

        def lossk():
            rc = lib.ltr_pairwise_loss_f32(k, 1.0, scores.data_ptr(), rel.data_ptr(), 0, n.data_ptr(),
                                           B, L, loss.data_ptr(), ds.data_ptr(), cs())
            assert rc == 0, rc

        out = [os.path.basename(path)]
        for name, fn in (("fused", fused), ("reduce", reduce), ("loss", lossk)):
            for _ in range(10):
                fn()
            avg, med, mn = time_events(fn, args.iters)
            bat, _ = time_launches(fn, per_graph=20, replays=10)
            out.append("%s %.2f us (evt-pair %.2f)" % (name, bat, avg))
        fused(); reduce()
        torch.cuda.synchronize()
        chk = (float(loss.double().sum()), float(dW.double().abs().sum()))
        if ref is None:
            ref = chk
        out.append("chk %.6g %.6g%s" % (chk[0], chk[1], "" if abs(chk[0] - ref[0]) < 1e-3 * abs(ref[0]) + 1e-6 and abs(chk[1] - ref[1]) < 1e-3 * abs(ref[1]) + 1e-6 else "  MISMATCH"))
        if args.trace:
            tr = trbuf
            for _ in range(3):
                rc = lib.ltr_linear_partials_f32(k, 1.0, X.data_ptr(), W.data_ptr(), bias.data_ptr(),
                                                 rel.data_ptr(), 0, n.data_ptr(), B, L, F, loss.data_ptr(),
                                                 tr.data_ptr(), part.data_ptr(), st)
                assert rc == 0
            torch.cuda.synchronize()
            t = tr.cpu().view(B, 8).double()
            t0 = t[:, 0].min()
            names = ["dma+labels", "gemv", "pair", "gfin", "dW", ]
            d = t[:, 1:6] - t[:, 0:5]
            print("  trace (cycles): kernel span %.0f | per-query mean " % (t[:, 5].max() - t0) +
                  " ".join("%s %.0f" % (nm, d[:, i].mean()) for i, nm in enumerate(names)) +
                  " | total/query mean %.0f max %.0f" % ((t[:, 5] - t[:, 0]).mean(), (t[:, 5] - t[:, 0]).max()))
            print("  entry -> first stamp (scheduling): mean %.0f max %.0f cycles" % ((t[:, 0] - t[:, 6]).mean(), (t[:, 0] - t[:, 6]).max()))
            nbv = n.cpu().double()
            big = nbv > 100
            print("  queries with n>100: " + " ".join("%s %.0f" % (nm, d[big, i].mean()) for i, nm in enumerate(names)))
            blk = torch.arange(B)
            per_block_end = torch.zeros(int(blk.max()) + 1).double()
            per_block_cnt = torch.zeros(int(blk.max()) + 1)
            for i in range(B):
                per_block_end[blk[i]] = max(per_block_end[blk[i]], t[i, 5] - t0)
                per_block_cnt[blk[i]] += 1
            print("  block end cycles: min %.0f mean %.0f max %.0f | queries per block min %d max %d; first-start spread %.0f" % (
                per_block_end.min(), per_block_end.mean(), per_block_end.max(), per_block_cnt.min(), per_block_cnt.max(),
                (t[:, 0].sort().values[min(B, 512) - 1] - t0)))
        for cfg in [c for c in args.loss_cfgs.split(";") if c]:
            o, d, m = (int(v) for v in cfg.split(","))

            def lossc():
                rc = lib.ltr_pairwise_loss_f32_cfg(k, 1.0, scores.data_ptr(), rel.data_ptr(), 0, n.data_ptr(),
                                                   B, L, loss.data_ptr(), ds.data_ptr(), o, d, m, cs())
                assert rc == 0, rc
            for _ in range(10):
                lossc()
            bat, _ = time_launches(lossc, per_graph=20, replays=10)
            out.append("loss[%s] %.2f" % (cfg, bat))
        print(" | ".join(out), flush=True)


if __name__ == "__main__":
    main()
