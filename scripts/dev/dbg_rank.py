import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import ltr_oracle as O
from tests.conftest import synth
from pytorchltr_amd import _C
dev = torch.device("cuda:0")
B, L, F, kind = 512, 1000, 220, "hinge"
s, y, n, X, W, b = synth(B, L, 5, F=F)
lib = _C.lib()
k = O.KINDS[kind]
print("plan", lib.ltr_linear_fused_plan(k, B, L, F))
Xd, Wd, bd, yd, nd = X.to(dev), W.to(dev), b.to(dev), y.to(dev), n.to(dev)
ws = torch.empty(lib.ltr_linear_workspace_bytes(B, L, F) // 4 + 64, device=dev)
loss = torch.empty(B, device=dev)
rc = lib.ltr_linear_partials_f32(k, 1.0, Xd.data_ptr(), Wd.data_ptr(), bd.data_ptr(), yd.data_ptr(), 0, nd.data_ptr(), B, L, F,
                                 loss.data_ptr(), None, ws.data_ptr(), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
want_l, sc, want_dW, want_db = O.linear_pairwise(kind, X.numpy(), W.numpy(), float(b[0]), y.numpy(), n.numpy(), np.full(B, 1.0 / B))
got = loss.cpu().numpy()
bad = np.nonzero(~np.isclose(got, want_l, rtol=5e-4, atol=1e-5))[0]
print("bad rows", bad, "n", n.numpy()[bad], "got", got[bad], "want", want_l[bad])
PF = (F + 4) & ~3
part = ws[:B * PF].reshape(B, PF).cpu().numpy()
for r in bad[:2]:
    # per-query gradient check through the partial row: dW_b = sum_l g_l x_l ; recover g via least squares is overkill: compare the bias partial (sum g) and a few columns
    nn = int(n[r])
    _, gs = O.pairwise_loss(kind, sc[r:r+1], y.numpy()[r:r+1], n.numpy()[r:r+1])
    wantrow = gs[0, :nn] @ X.numpy()[r, :nn, :]
    print("row", r, "n", nn, "max dW diff", np.abs(part[r, :F] - wantrow).max(), "of", np.abs(wantrow).max(), "db got/want", part[r, F], gs[0].sum())
    srow = sc[r, :nn]
    print("score min/max", srow.min(), srow.max(), "labels", np.bincount(y.numpy()[r, :nn].astype(int)))
