#!/usr/bin/env python
"""Fused scorer+loss kernel time per loss kind for one or more builds:
python scripts/time_fused_kinds.py lib1.so lib2.so [--workload c2] [--full]"""
import argparse
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import WORKLOADS, synth  # noqa: E402
from _benchutil import time_launches  # noqa: E402
from pytorchltr_amd import _C  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+")
ap.add_argument("--workload", default="c2")
ap.add_argument("--full", action="store_true")
ap.add_argument("--kinds", default="hinge,dcg_hinge,logistic,arp1,arp2,ndcg1,ndcg2")
args = ap.parse_args()
dev = torch.device("cuda:0")
B, L, F, _ = WORKLOADS[args.workload]
_, rel, n, X = synth(B, L, F, 0, dev)
if args.full:
    n = torch.full_like(n, L)
W = (torch.rand(F, device=dev) * 2 - 1) / F ** 0.5
bias = torch.zeros(1, device=dev)
loss = torch.empty(B, device=dev)
for path in args.libs:
    h = ctypes.CDLL(path)
    for name, (restype, argtypes) in _C.SIGNATURES.items():
        fn = getattr(h, name)
        fn.restype = restype
        fn.argtypes = argtypes
    part = torch.empty(h.ltr_linear_workspace_bytes(B, L, F) // 4 + 64, device=dev)
    out = [os.path.basename(path)]
    for kind in args.kinds.split(","):
        k = getattr(_C, kind.upper())

        def f():
            rc = h.ltr_linear_partials_f32(k, 1.0, X.data_ptr(), W.data_ptr(), bias.data_ptr(), rel.data_ptr(), 0,
                                           n.data_ptr(), B, L, F, loss.data_ptr(), None, part.data_ptr(),
                                           torch.cuda.current_stream().cuda_stream)
            assert rc == 0, rc
        for _ in range(5):
            f()
        t, _ = time_launches(f, per_graph=20, replays=10)
        out.append("%s %.2f (loss %.6g)" % (kind, t, float(loss.double().sum())))
    print(" | ".join(out), flush=True)
