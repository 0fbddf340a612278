#!/usr/bin/env python
"""Launch only the fused scorer+loss kernel a few times (for rocprofv3 --pmc passes):
python scripts/run_kernel_only.py [lib.so] [--workload c2] [--full] [--kind hinge] [--iters 20]"""
import argparse
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import WORKLOADS, synth  # noqa: E402
from pytorchltr_amd import _C  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("lib", nargs="?", default=_C.LIB_PATH)
ap.add_argument("--workload", default="c2")
ap.add_argument("--kind", default="")
ap.add_argument("--full", action="store_true")
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--nbuf", type=int, default=5)
args = ap.parse_args()
h = ctypes.CDLL(args.lib)
for name, (restype, argtypes) in _C.SIGNATURES.items():
    fn = getattr(h, name)
    fn.restype = restype
    fn.argtypes = argtypes
dev = torch.device("cuda:0")
B, L, F, kind = WORKLOADS[args.workload]
kind = args.kind or kind
k = getattr(_C, kind.upper())
bufs = []
for i in range(args.nbuf):
    _, rel, n, X = synth(B, L, F, 100 + i, dev)
    if args.full:
        n = torch.full_like(n, L)
    bufs.append((X, rel, n))
W = (torch.rand(F, device=dev) * 2 - 1) / F ** 0.5
bias = torch.zeros(1, device=dev)
loss = torch.empty(B, device=dev)
part = torch.empty(h.ltr_linear_workspace_bytes(B, L, F) // 4 + 64, device=dev)
st = torch.cuda.current_stream().cuda_stream
for it in range(args.iters):
    X, rel, n = bufs[it % args.nbuf]
    rc = h.ltr_linear_partials_f32(k, 1.0, X.data_ptr(), W.data_ptr(), bias.data_ptr(), rel.data_ptr(), 0,
                                   n.data_ptr(), B, L, F, loss.data_ptr(), None, part.data_ptr(), st)
    assert rc == 0, rc
torch.cuda.synchronize()
print("done", float(loss.sum()))
