import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import ltr_oracle as O
from tests.conftest import synth
from pytorchltr_amd import _C
from pytorchltr_amd.fused import linear_loss_step
dev = torch.device("cuda:0")
for kind, B, L, F in [("ndcg2", 200, 256, 136), ("ndcg1", 200, 200, 136), ("ndcg2", 100, 162, 220), ("ndcg1", 64, 256, 100), ("ndcg2", 50, 285, 136)]:
    s, y, n, X, W, b = synth(B, L, 4, F=F)
    n[:4] = torch.tensor([0, 1, L, L - 1])
    print(kind, B, L, F, "plan", _C.lib().ltr_linear_fused_plan(getattr(_C, kind.upper()), B, L, F))
    loss, dW, db = linear_loss_step(X.to(dev), W.to(dev), b.to(dev), y.to(dev), n.to(dev), loss=kind)
    want_l, _, want_dW, want_db = O.linear_pairwise(kind, X.numpy(), W.numpy(), float(b[0]), y.numpy(), n.numpy(), np.full(B, 1.0 / B))
    got = loss.cpu().numpy()
    rel = np.abs(got - want_l) / np.maximum(1e-5, np.abs(want_l))
    print("  max rel loss err", rel.max(), " dW err", np.max(np.abs(dW.cpu().numpy() - want_dW)), "scale", np.max(np.abs(want_dW)), "db", abs(float(db[0]) - want_db))
