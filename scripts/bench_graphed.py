#!/usr/bin/env python
"""Eager vs GraphedStep wall time per training step (forward + backward + optimizer)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from pytorchltr_amd.fused import FusedLinearLoss, FusedMLPLoss  # noqa: E402
from pytorchltr_amd.graphed import GraphedStep  # noqa: E402
from pytorchltr_amd.loss import PairwiseHingeLoss  # noqa: E402

dev = torch.device("cuda", 0)


def run(name, B, L, F, make):
    _, y, n, X = bench.synth(B, L, F, 0, dev)
    model, closure = make(F)
    opt = torch.optim.Adam(model.parameters(), lr=0.01, capturable=True)

    def eager():
        opt.zero_grad(set_to_none=True)
        loss = closure(model, X, y, n)
        loss.backward()
        opt.step()
    for _ in range(20):
        eager()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(300):
        eager()
    torch.cuda.synchronize(); te = (time.perf_counter() - t0) / 300 * 1e6
    step = GraphedStep(model, opt, lambda xs, ys, nn: closure(model, xs, ys, nn), (X, y, n))
    for _ in range(20):
        step(X, y, n)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(1000):
        step(X, y, n)
    torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / 1000 * 1e6
    print("%-34s B=%d L=%d F=%d | eager %.1f us/step | graphed %.1f us/step | x%.1f" % (name, B, L, F, te, tg, te / tg), flush=True)


loss_fn = PairwiseHingeLoss()
for B, L in ((16, 20), (1024, 128)):
    run("nn.Linear + PairwiseHingeLoss", B, L, 136,
        lambda F: (torch.nn.Linear(F, 1).to(dev), lambda m, xs, ys, n: loss_fn(m(xs), ys, n).mean()))
    run("FusedLinearLoss", B, L, 136,
        lambda F: (FusedLinearLoss(F, "hinge").to(dev), lambda m, xs, ys, n: m(xs, ys, n).mean()))
    run("FusedMLPLoss (136-50-10-1)", B, L, 136,
        lambda F: (FusedMLPLoss(F, "hinge").to(dev), lambda m, xs, ys, n: m(xs, ys, n)))
