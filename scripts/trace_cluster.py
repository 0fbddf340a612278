#!/usr/bin/env python
"""Per-phase wall-clock trace of the cluster kernel (ltr_cluster.inc) on a -DLTR_TRACE build:

    scripts/build_variants.sh cltr:"-DLTR_TRACE"
    python scripts/trace_cluster.py B L F [kind]

Workgroups stamp the 100 MHz wall clock (identical on every XCD) into the score-output buffer."""
import ctypes
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synth  # noqa: E402
from pytorchltr_amd import _C  # noqa: E402

dev = torch.device("cuda:0")
lib = ctypes.CDLL(os.path.join(ROOT, "build", "variants", "libltr_cltr.so"))
for name, (res, argt) in _C.SIGNATURES.items():
    getattr(lib, name).restype = res
    getattr(lib, name).argtypes = argt
B, L, F = (int(v) for v in sys.argv[1:4])
kind = int(sys.argv[4]) if len(sys.argv) > 4 else 0
scores, rel, n, X = synth(B, L, F, 0, dev)
W = torch.randn(F, device=dev) * 0.1
bias = torch.randn(1, device=dev)
loss = torch.empty(B, device=dev)
part = torch.empty(lib.ltr_linear_workspace_bytes(B, L, F) // 4 + 64, device=dev)
tr = torch.zeros(B * 16 * 8, dtype=torch.int64, device=dev)


def launch():
    rc = lib.ltr_linear_partials_f32(kind, 1.0, X.data_ptr(), W.data_ptr(), bias.data_ptr(), rel.data_ptr(), 0,
                                     n.data_ptr(), B, L, F, loss.data_ptr(), tr.data_ptr(), part.data_ptr(),
                                     torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc


for _ in range(3):
    launch()
torch.cuda.synchronize()
t0h = time.perf_counter()
for _ in range(10):
    launch()
torch.cuda.synchronize()
print("host-timed per call: %.1f us" % ((time.perf_counter() - t0h) / 10 * 1e6))
tr.zero_()
launch()
torch.cuda.synchronize()
t = tr.cpu().view(-1, 8)
ex = t[t[:, 7] == -1]
live = (t[:, 0] != 0) & (t[:, 7] != -1)
t = t[live].double()
t0 = min(float(t[:, 0].min()), float(ex[:, 0].min()) if ex.shape[0] else 1e300)
names = ["load+scores", "wait1", "pair", "wait2", "dW", "final"]
d = (t[:, 1:7] - t[:, 0:6]) / 100.0
print("workgroups with rows: %d, leaving at once: %d; span %.1f us" % (t.shape[0], ex.shape[0], (t[:, 6].max() - t0) / 100.0))
print(" | ".join("%s mean %.1f max %.1f" % (nm, d[:, i].mean(), d[:, i].max()) for i, nm in enumerate(names)))
q = torch.tensor([0.1, 0.5, 0.9, 1.0], dtype=torch.float64)
print("start (us) p10/p50/p90/max:", [round(float(v), 1) for v in torch.quantile((t[:, 0] - t0) / 100.0, q)],
      " end:", [round(float(v), 1) for v in torch.quantile((t[:, 6] - t0) / 100.0, q)])
if ex.shape[0]:
    print("early-exit stamps (us) p50/max:", [round(float(v), 1) for v in torch.quantile((ex[:, 0].double() - t0) / 100.0, torch.tensor([0.5, 1.0], dtype=torch.float64))])
