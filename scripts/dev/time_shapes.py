#!/usr/bin/env python
"""Fused scorer + loss kernel time for a list of shapes and kinds on whatever plan the dispatcher picks
(LTR_DISABLE_PARTS=1 / LTR_PARTS_ALL=1 in the environment steer it):
python scripts/dev/time_shapes.py --kinds ndcg2,hinge 512x512x700 1024x512x700"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from bench import synth, nbuf_for, time_launches  # noqa: E402
from pytorchltr_amd import _C  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("shapes", nargs="+")
ap.add_argument("--kinds", default="hinge,ndcg2")
ap.add_argument("--full", action="store_true")
args = ap.parse_args()
dev = torch.device("cuda:0")
h = _C.lib()
for shp in args.shapes:
    B, L, F = (int(v) for v in shp.split("x"))
    # cold: as many distinct batches in rotation as exceed the Infinity Cache (bench.py's rule)
    nbuf = nbuf_for(B, L, F)
    bat = []
    for i in range(nbuf):
        _, rel, n, X = synth(B, L, F, i, dev)
        if args.full:
            n = torch.full_like(n, L)
        bat.append((rel, n, X))
    g = torch.Generator().manual_seed(1)
    W = ((torch.rand(F, generator=g) * 2 - 1) / F ** 0.5).to(dev)
    bias = torch.zeros(1, device=dev)
    loss = torch.empty(B, device=dev)
    part = torch.empty(h.ltr_linear_workspace_bytes(B, L, F) // 4 + 64, device=dev)
    out = ["%s x%d" % (shp, nbuf)]
    for kind in args.kinds.split(","):
        k = getattr(_C, kind.upper())

        def f(i):
            rel, n, X = bat[i % nbuf]
            rc = h.ltr_linear_partials_f32(k, 1.0, X.data_ptr(), W.data_ptr(), bias.data_ptr(), rel.data_ptr(), 0,
                                           n.data_ptr(), B, L, F, loss.data_ptr(), None, part.data_ptr(),
                                           torch.cuda.current_stream().cuda_stream)
            assert rc == 0, rc
        for i in range(5):
            f(i)
        t, _ = time_launches(f, nbuf, rounds=max(2, 20 // nbuf), replays=10)
        out.append("%s plan %d %.2f us (loss %.6g)" % (kind, h.ltr_linear_fused_plan(k, B, L, F), t, float(loss.double().sum())))
    print(" | ".join(out), flush=True)
