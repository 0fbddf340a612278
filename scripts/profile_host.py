#!/usr/bin/env python
"""Host-side (Python) overhead of the eager drop-in path: cProfile of
`loss_fn(scores, y, n).mean().backward()` and of the fused modules."""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from pytorchltr_amd.loss import PairwiseHingeLoss  # noqa: E402
from pytorchltr_amd.evaluation import ndcg  # noqa: E402

dev = torch.device("cuda", 0)
scores, y, n, X = bench.synth(1024, 128, 136, 0, dev)
sc = scores.clone().requires_grad_(True)
loss_fn = PairwiseHingeLoss()


def step():
    sc.grad = None
    loss_fn(sc, y, n).mean().backward()


def metric():
    ndcg(scores, y, n, k=10)


from pytorchltr_amd.fused import FusedLinearLoss, LinearScorer  # noqa: E402
fused = FusedLinearLoss(136, "hinge").to(dev)
scorer = LinearScorer(136).to(dev)


def fused_step():
    fused.zero_grad(set_to_none=True)
    fused(X, y, n).mean().backward()


def scorer_step():
    scorer.zero_grad(set_to_none=True)
    loss_fn(scorer(X, n), y, n).mean().backward()


CASES = {"loss": ("loss step", step), "ndcg": ("ndcg@10", metric), "fused": ("FusedLinearLoss step", fused_step),
         "scorer": ("LinearScorer + loss step", scorer_step)}
want = sys.argv[1:] or ["loss", "ndcg"]
for name, fn in (CASES[w] for w in want):
    for _ in range(200):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2000):
        fn()
    torch.cuda.synchronize()
    print("%s: %.1f us/iter eager" % (name, (time.perf_counter() - t0) / 2000 * 1e6))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(2000):
        fn()
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(22)
