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
lib = ctypes.CDLL(os.environ.get("LTR_TRACE_LIB") or os.path.join(ROOT, "build", "variants", "libltr_cltr.so"))
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
tr = torch.zeros((B + 7) // 8 * 8 * 16 * 16, dtype=torch.int64, device=dev)     # (the grid is padded to groups of eight queries)


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
END = 10
t = tr.cpu().view(-1, 16)
ex = t[t[:, 15] == -1]
live = (t[:, 0] != 0) & (t[:, 15] != -1)
t = t[live].double()
t0 = min(float(t[:, 0].min()), float(ex[:, 0].min()) if ex.shape[0] else 1e300)
# stamps: 0 start, 1 scores, 2 published (by runs: local ranks in between), 3 after wait 1, [by runs: 4 runs scattered,
# 5 tables scanned, 6 searches done,] 7 pair pass / runs done, 8 after wait 2, 9 dW share published, 10 end
byruns = bool((t[:, 4] != 0).any())
marks = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10] if byruns else [0, 1, 2, 3, 7, 8, 9, 10]
names = {1: "load+scores", 2: "local ranks+publish", 3: "wait1", 4: "read+scatter", 5: "scan", 6: "searches", 7: "reduce" if byruns else "pair",
         8: "wait2", 9: "dW", 10: "final"}
print("workgroups with rows: %d, leaving at once: %d; span %.1f us%s" % (t.shape[0], ex.shape[0], (t[:, END].max() - t0) / 100.0, " (hinge by sorted runs)" if byruns else ""))
print(" | ".join("%s mean %.2f max %.2f" % (names[b_], ((t[:, b_] - t[:, a_]) / 100.0).mean(), ((t[:, b_] - t[:, a_]) / 100.0).max()) for a_, b_ in zip(marks[:-1], marks[1:])))
q = torch.tensor([0.1, 0.5, 0.9, 1.0], dtype=torch.float64)
print("start (us) p10/p50/p90/max:", [round(float(v), 1) for v in torch.quantile((t[:, 0] - t0) / 100.0, q)],
      " end:", [round(float(v), 1) for v in torch.quantile((t[:, END] - t0) / 100.0, q)])
if ex.shape[0]:
    print("early-exit stamps (us) p50/max:", [round(float(v), 1) for v in torch.quantile((ex[:, 0].double() - t0) / 100.0, torch.tensor([0.5, 1.0], dtype=torch.float64))])

# ---- back-to-back: four launches in one hipGraph, each with its own trace buffer: spans and the gaps between them ----
from bench import try_graph  # noqa: E402
K = 6
trs = [torch.zeros_like(tr) for _ in range(K)]


def many():
    for j in range(K):
        rc = lib.ltr_linear_partials_f32(kind, 1.0, X.data_ptr(), W.data_ptr(), bias.data_ptr(), rel.data_ptr(), 0,
                                         n.data_ptr(), B, L, F, loss.data_ptr(), trs[j].data_ptr(), part.data_ptr(),
                                         torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc


replay = try_graph(many, warm=1)
if replay is not None:
    for t_ in trs:
        t_.zero_()
    torch.cuda.synchronize()
    replay()
    torch.cuda.synchronize()
    first, last, lastlive = [], [], []
    for t_ in trs:
        v = t_.cpu().view(-1, 16)
        started = v[:, 0] != 0
        livem = started & (v[:, 15] != -1)
        first.append(int(v[started, 0].min()))
        last.append(int(v[livem, END].max()))
    print("back-to-back (graph): span us", [round((last[j] - first[j]) / 100.0, 1) for j in range(K)],
          " gap last end -> next first start us", [round((first[j + 1] - last[j]) / 100.0, 1) for j in range(K - 1)],
          " period us", [round((first[j + 1] - first[j]) / 100.0, 1) for j in range(K - 1)])
