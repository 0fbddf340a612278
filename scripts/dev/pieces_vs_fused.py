"""FusedLinearLoss: the pieces path (scorer + split-query loss + weight-gradient kernels) against the one-kernel fused
path, module forward + backward replayed as a hipGraph, cold rotation.  python scripts/dev/pieces_vs_fused.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from pytorchltr_amd import _C, fused
dev = torch.device("cuda:0")
for kind, B, L, F in [("ndcg2", 80, 700, 512), ("ndcg2", 64, 512, 700), ("ndcg1", 100, 1000, 700), ("logistic", 100, 1000, 700),
                      ("hinge", 100, 1000, 700), ("ndcg2", 80, 700, 64), ("ndcg2", 120, 1000, 512), ("hinge", 120, 600, 700)]:
    nbuf = bench.nbuf_for(B, L, F)
    bat = bench.make_batches(B, L, F, nbuf, 0, dev)
    out = ["%s %dx%dx%d plan %d" % (kind, B, L, F, _C.lib().ltr_linear_fused_plan(getattr(_C, kind.upper()), B, L, F))]
    for force in (True, False):
        m = fused.FusedLinearLoss(F, kind).to(dev)
        fused._pieces_cache[(m.kind, B, L, F)] = force
        def step(i):
            b = bat[i % nbuf]
            m.weight.grad = None; m.bias.grad = None
            m(b["X"], b["rel"], b["n"]).mean().backward()
        for i in range(3):
            step(i)
        us, graphed = bench.time_launches(step, nbuf, rounds=2, replays=5)
        out.append("%s %.1f us%s" % ("pieces" if force else "fused ", us, "" if graphed else " (eager)"))
    fused._pieces_cache.clear()
    print(" | ".join(out), flush=True)
