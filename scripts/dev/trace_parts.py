#!/usr/bin/env python
"""Per-phase wall-clock trace of the parts kernel on a -DLTR_TRACE -DLTR_TRACE_WALL build of the linear TU:
   python scripts/dev/trace_parts.py B L F kind [lib]"""
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
t0 = float(tt[:, 0].min())
ent = t[t[:, 15] != 0][:, 15].double()
if ent.numel():
    t0 = min(t0, float(ent.min()))
names = ["decode+issue", "labels+dots", "fold+publish", "poll", "pair", "gloc+dW", "finish"]
d = (tt[:, 1:8] - tt[:, 0:7]) / 100.0
print("parts: %d  span %.1f us  (first entry -> last end)" % (t.shape[0], (tt[:, 7].max() - t0) / 100.0))
print("prologue (entry -> first part start) mean %.2f max %.2f us" % tuple(
    float(v) for v in [((tt[t[:, 15] != 0][:, 0] - ent) / 100.0).mean(), ((tt[t[:, 15] != 0][:, 0] - ent) / 100.0).max()]))
print(" | ".join("%s %.2f/%.2f" % (nm, d[:, i].mean(), d[:, i].max()) for i, nm in enumerate(names)), "(mean/max us)")
tot = (tt[:, 7] - tt[:, 0]) / 100.0
print("part duration mean %.2f max %.2f; sum over parts / slots = %.1f us" % (tot.mean(), tot.max(), tot.sum() / max(1, len(set(t[:, 9].tolist())))))
rows = (t[:, 8] & 0xff)
P = (t[:, 8] >> 8) & 0xff
for pv in sorted(set(P.tolist())):
    m = P == pv
    print("  P=%d: %d parts, dur %.2f, poll %.2f, pair %.2f, load %.2f" % (pv, int(m.sum()), tot[m].mean(), d[m, 3].mean(), d[m, 4].mean(), d[m, 1].mean()))
e = ((ent - t0) / 100.0).sort().values
q = torch.tensor([0.0, 0.25, 0.5, 0.75, 0.9, 1.0], dtype=torch.float64)
print("workgroups with a first part: %d; entry times (us) p0/25/50/75/90/100:" % ent.numel(), [round(float(v), 1) for v in torch.quantile(e, q)])
import collections
hw = collections.Counter()

f = t[t[:, 15] != 0].double()
print("prologue phases (us, mean): n+ballots %.2f | prefix+table+order %.2f | ctrl %.2f | to first stamp %.2f" % (
    float(((f[:, 10] - f[:, 15]) / 100).mean()), float(((f[:, 11] - f[:, 10]) / 100).mean()),
    float(((f[:, 12] - f[:, 11]) / 100).mean()), float(((f[:, 0] - f[:, 12]) / 100).mean())))

rk = t[(t[:, 14] > t[:, 4]) & (t[:, 14] < t[:, 5]) & (t[:, 10] > t[:, 4])].double()
if rk.shape[0]:
    names = ["S1 qualify", "S2-3 hist+scan", "S4 scatter", "S5 bucket sort+agg", "S6 scans", "S7 rows+sum"]
    seg = [rk[:, 10] - rk[:, 4], rk[:, 11] - rk[:, 10], rk[:, 12] - rk[:, 11], rk[:, 13] - rk[:, 12], rk[:, 14] - rk[:, 13], rk[:, 5] - rk[:, 14]]
    print("ranked parts %d: " % rk.shape[0] + " | ".join("%s %.2f" % (n_, float(v.mean()) / 100) for n_, v in zip(names, seg)))
