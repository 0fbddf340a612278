"""Practical streaming floor on this box: time plain reads of the bench tensors (event pairs and
graph-batched), to calibrate what 36/71 MB cost at best."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from _benchutil import time_events, time_launches
dev = torch.device("cuda:0")
for mb in (4, 18, 36, 71, 142, 568):
    n = mb * 1024 * 1024 // 4
    x = torch.randn(n, device=dev)
    y = torch.empty_like(x)
    for name, fn in (("sum", lambda: x.sum()), ("copy", lambda: y.copy_(x)), ("fill", lambda: y.fill_(1.0))):
        for _ in range(5):
            fn()
        avg, med, mn = time_events(fn, 50)
        bavg, graphed = time_launches(fn, per_graph=10, replays=5)
        moved = mb * (2 if name == "copy" else 1)
        print("%4d MB %-5s event-pair avg %.2f us min %.2f | batched %.2f us (%s) -> %.2f TB/s" % (
            mb, name, avg, mn, bavg, "graph" if graphed else "eager", moved / 1e6 * 1.048576 / (bavg * 1e-6) / 1e3 * 1e-3 * 1e3))
