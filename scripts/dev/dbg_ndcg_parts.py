import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import ltr_oracle as O
from tests.conftest import synth
from pytorchltr_amd import _C
from pytorchltr_amd.fused import linear_loss_step
dev = torch.device("cuda:0")
for kind, B, L, F in [("ndcg2", 128, 1000, 136), ("ndcg2", 256, 1000, 220)]:
    s, y, n, X, W, b = synth(B, L, 4, F=F)
    print(kind, B, L, F, "plan", _C.lib().ltr_linear_fused_plan(getattr(_C, kind.upper()), B, L, F))
    loss, dW, db = linear_loss_step(X.to(dev), W.to(dev), b.to(dev), y.to(dev), n.to(dev), loss=kind)
    want_l, _, want_dW, want_db = O.linear_pairwise(kind, X.numpy(), W.numpy(), float(b[0]), y.numpy(), n.numpy(), np.full(B, 1.0 / B))
    got = loss.cpu().numpy()
    rel = (got - want_l) / np.maximum(1e-5, np.abs(want_l))
    print(" rel err percentiles (signed)", np.percentile(rel, [0, 1, 10, 50, 90, 99, 100]))
    i = int(np.argmax(np.abs(rel)))
    print("   worst b", i, "n", int(n[i]), "got", got[i], "want", want_l[i])
    print("   b 54 n", int(n[54]), "got", got[54], "want", want_l[54])
