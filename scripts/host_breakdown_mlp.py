#!/usr/bin/env python
"""Host-side cost of the eager FusedMLPLoss step (forward + backward), piece by piece, next to the
guide's own composition (three nn.Linear + ReLU + loss module) and to a native autograd graph with
six parameter leaves:  python scripts/host_breakdown_mlp.py"""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from pytorchltr_amd import fused  # noqa: E402
from pytorchltr_amd.fused import FusedMLPLoss  # noqa: E402

dev = torch.device("cuda", 0)
scores, y, n, X = bench.synth(1024, 128, 136, 0, dev)
torch.manual_seed(0)
m = FusedMLPLoss(136, "hinge").to(dev)
ps = list(m.parameters())


def timeit(fn, iters=1000):
    for _ in range(100):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6


def module_step():
    for p in ps:
        p.grad = None
    m(X, y, n).backward()


def module_fwd():
    m(X, y, n)


def direct_step():
    fused.mlp_loss_step(X, m._params(), y, n, loss="hinge")


def native_six_leaves():
    for p in ps:
        p.grad = None
    t = ps[0].sum() + ps[1].sum() + ps[2].sum() + ps[3].sum() + ps[4].sum() + ps[5].sum()
    t.backward()


for name, fn in (("module fwd+bwd", module_step), ("module fwd only", module_fwd),
                 ("mlp_loss_step (no autograd)", direct_step), ("native graph, 6 leaves", native_six_leaves)):
    print("%-32s %8.1f us" % (name, timeit(fn)))
pr = cProfile.Profile()
pr.enable()
for _ in range(500):
    module_step()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)

# backward alone (the graph kept): engine + the custom backward + six AccumulateGrad nodes
out = m(X, y, n)


def bwd_keep_grads():
    out.backward(retain_graph=True)


def bwd_fresh_grads():
    for p in ps:
        p.grad = None
    out.backward(retain_graph=True)


def clear_grads():
    for p in ps:
        p.grad = None


print("%-32s %8.1f us" % ("backward only, grads accumulate", timeit(bwd_keep_grads)))
print("%-32s %8.1f us" % ("backward only, grads = None first", timeit(bwd_fresh_grads)))
print("%-32s %8.1f us" % ("p.grad = None x 6", timeit(clear_grads)))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(20):
        module_step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=25, max_name_column_width=60))
