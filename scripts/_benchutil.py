"""Timing helpers shared by the tuning scripts (same-batch launches; bench.py itself rotates over
several batches, see its docstring)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

try_graph = bench.try_graph


def time_events(fn, iters):
    return bench.time_events(lambda i: fn(), iters)


def time_launches(fn, per_graph=20, replays=10):
    """Average launch duration (us): `per_graph` back-to-back launches of fn() in one hipGraph."""
    return bench.time_launches(lambda i: fn(), 1, rounds=per_graph, replays=replays)


def time_wall(fn, steps, barrier):
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    barrier()
    return time.perf_counter() - t0
