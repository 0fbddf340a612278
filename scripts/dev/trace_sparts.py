#!/usr/bin/env python
"""Per-phase wall-clock trace of the parts kernel's SYM mode on a -DLTR_TRACE -DLTR_TRACE_WALL build:
   python scripts/dev/trace_sparts.py B L F kind [lib]"""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import synth
from pytorchltr_amd import _C
dev = torch.device("cuda:0")
B, L, F = (int(v) for v in sys.argv[1:4])
kind = int(sys.argv[4]) if len(sys.argv) > 4 else 0
lib = ctypes.CDLL(sys.argv[5] if len(sys.argv) > 5 else os.path.join(ROOT, "build", "variants", "libltr_ptrace.so"))
for name, (res, argt) in _C.SIGNATURES.items():
    if hasattr(lib, name):
        getattr(lib, name).restype = res
        getattr(lib, name).argtypes = argt
assert lib.ltr_linear_fused_plan(kind, B, L, F) == 4
scores, rel, n, X = synth(B, L, F, 0, dev)
W = torch.randn(F, device=dev) * 0.1
bias = torch.randn(1, device=dev)
loss = torch.empty(B, device=dev)
part = torch.empty(lib.ltr_linear_workspace_bytes(B, L, F) // 4 + 64, device=dev)
NT = B * 40
tr = torch.zeros(NT * 16, dtype=torch.int64, device=dev)
def launch():
    rc = lib.ltr_linear_partials_f32(kind, 1.0, X.data_ptr(), W.data_ptr(), bias.data_ptr(), rel.data_ptr(), 0,
                                     n.data_ptr(), B, L, F, loss.data_ptr(), tr.data_ptr(), part.data_ptr(),
                                     torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc
for _ in range(3):
    launch()
torch.cuda.synchronize()
tr.zero_()
launch()
torch.cuda.synchronize()
t = tr.cpu().view(-1, 16)
t = t[t[:, 0] != 0]
tt = t.double()
ent = t[t[:, 15] != 0][:, 15].double()
t0 = float(ent.min()) if ent.numel() else float(tt[:, 0].min())
P = (t[:, 8] >> 8) & 0xff
sym = tt[:, 12] > 0
# stamp order in SYM mode: 0 1 2 3 4 12 13 14 5 6 7
order = [0, 1, 2, 3, 4, 12, 13, 14, 5, 6, 7]
names = ["decode+issue", "labels+dots", "fold+publish", "poll scores", "pair share", "fold+publish slices", "poll slices", "gloc", "dW", "finish"]
print("parts: %d  span %.1f us (first entry -> last end); workgroups %d" % (t.shape[0], (tt[:, 7].max() - t0) / 100.0, len(set(t[:, 9].tolist()))))
if ent.numel():
    f = tt[t[:, 15] != 0]
    print("prologue: entry spread %.2f us; entry -> first part start mean %.2f max %.2f" % (
        (ent.max() - ent.min()) / 100.0, float(((f[:, 0] - f[:, 15]) / 100).mean()), float(((f[:, 0] - f[:, 15]) / 100).max())))
def show(mask, label):
    x = tt[mask]
    if x.shape[0] == 0:
        return
    segs = []
    for i in range(len(order) - 1):
        a, b = x[:, order[i]], x[:, order[i + 1]]
        ok = (a > 0) & (b > 0)
        d = ((b - a)[ok]) / 100.0
        segs.append("%s %.2f/%.2f" % (names[i], float(d.mean()) if d.numel() else 0.0, float(d.max()) if d.numel() else 0.0))
    tot = (x[:, 7] - x[:, 0]) / 100.0
    print("%s: %d parts, duration mean %.2f max %.2f | " % (label, x.shape[0], tot.mean(), tot.max()) + " | ".join(segs))
show(torch.ones_like(P, dtype=torch.bool), "all")
for pv in sorted(set(P.tolist())):
    show(P == pv, "P=%d" % pv)
last = tt[:, 7].max()
print("end-of-kernel tail: last part end at %.1f us; parts ending in the last 5 us: %d" % ((last - t0) / 100.0, int((tt[:, 7] > last - 500).sum())))
