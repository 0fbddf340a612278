#!/usr/bin/env python
"""Host-side cost of the eager drop-in call, piece by piece, next to the floor of PyTorch's own
eager autograd for a graph of the same depth (a native op in place of the loss):
python scripts/host_breakdown.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from pytorchltr_amd.loss import PairwiseHingeLoss  # noqa: E402
from pytorchltr_amd.fused import FusedLinearLoss  # noqa: E402

dev = torch.device("cuda", 0)
scores, y, n, X = bench.synth(1024, 128, 136, 0, dev)
sc = scores.clone().requires_grad_(True)
loss_fn = PairwiseHingeLoss()
fused = FusedLinearLoss(136, "hinge").to(dev)


def timeit(fn, iters=3000):
    for _ in range(300):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6


def native_step():
    sc.grad = None
    (sc * 2.0).sum(1).mean().backward()


def native_fwd():
    (sc * 2.0).sum(1).mean()


def loss_fwd_nograd():
    with torch.no_grad():
        loss_fn(sc, y, n)


def loss_fwd():
    loss_fn(sc, y, n)


def loss_fwd_mean():
    loss_fn(sc, y, n).mean()


def loss_step():
    sc.grad = None
    loss_fn(sc, y, n).mean().backward()


def loss_step_sum():
    sc.grad = None
    loss_fn(sc, y, n).sum().backward()


g1 = torch.full((1024,), 1.0 / 1024, device=dev)


def loss_step_explicit_grad():
    sc.grad = None
    loss_fn(sc, y, n).backward(g1)


def fused_step():
    fused.weight.grad = None
    fused.bias.grad = None
    fused(X, y, n).mean().backward()


def fused_fwd():
    fused(X, y, n)


for name, fn in [("native (x*2).sum(1).mean() forward", native_fwd), ("native ... .backward()", native_step),
                 ("loss forward, no_grad", loss_fwd_nograd), ("loss forward (grad)", loss_fwd),
                 ("loss forward + mean", loss_fwd_mean), ("loss .mean().backward()", loss_step),
                 ("loss .sum().backward()", loss_step_sum), ("loss .backward(explicit grad)", loss_step_explicit_grad),
                 ("fused forward", fused_fwd), ("fused .mean().backward()", fused_step)]:
    print("%-40s %7.1f us" % (name, timeit(fn)), flush=True)
