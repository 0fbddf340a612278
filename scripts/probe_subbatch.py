#!/usr/bin/env python
"""Does the general fused kernel (features read twice) run faster in sub-batches whose features fit
the 256 MiB Infinity Cache?  python scripts/probe_subbatch.py [--workload c5] [--chunks 1,2,4]"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import WORKLOADS, synth  # noqa: E402
from probe_cold import graph_time  # noqa: E402
from pytorchltr_amd import _C  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="c5")
ap.add_argument("--chunks", default="1,2,4")
ap.add_argument("--nbuf", type=int, default=2)
args = ap.parse_args()
dev = torch.device("cuda:0")
B, L, F, kind = WORKLOADS[args.workload]
lib = _C.lib()
k = getattr(_C, kind.upper())
W = (torch.rand(F, device=dev) * 2 - 1) / F ** 0.5
bias = torch.zeros(1, device=dev)
lossv = torch.empty(B, device=dev)
PF = (F + 4) & ~3
part = torch.empty(lib.ltr_linear_workspace_bytes(B, L, F) // 4 + 64, device=dev)
bufs = []
for i in range(args.nbuf):
    _, rel, n, X = synth(B, L, F, 100 + i, dev)
    bufs.append((X, rel, n))


def mk(buf, chunks):
    X, rel, n = buf
    Bc = B // chunks

    def f():
        st = torch.cuda.current_stream().cuda_stream
        for c in range(chunks):
            o = c * Bc
            _C.check(lib.ltr_linear_partials_f32(
                k, 1.0, X.data_ptr() + 4 * o * L * F, W.data_ptr(), bias.data_ptr(),
                rel.data_ptr() + 8 * o * L, 0, n.data_ptr() + 8 * o, Bc, L, F,
                lossv.data_ptr() + 4 * o, None, part.data_ptr() + 4 * o * PF, st))
    return f


for chunks in [int(c) for c in args.chunks.split(",")]:
    fns = [mk(bufs[i % args.nbuf], chunks) for i in range(2 * args.nbuf)]
    us = graph_time(fns)
    print(json.dumps({"workload": args.workload, "chunks": chunks, "queries_per_chunk": B // chunks,
                      "plan": lib.ltr_linear_fused_plan(k, B // chunks, L, F), "us_per_batch": round(us, 1)}), flush=True)
