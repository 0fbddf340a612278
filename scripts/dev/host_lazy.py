"""Host cost of the lazy drop-in path, piece by piece (C2 shape)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pytorchltr_amd import loss as L_
from pytorchltr_amd.fused import FusedLinearLoss, LazyScores, use_linear_scorer, _LinearLossFunction
from pytorchltr_amd import _C
from tests.conftest import synth
dev = torch.device("cuda:0")
B, L, F = 1024, 128, 136
s, y, n, X, W, b = synth(B, L, 0, F=F)
X, y, n = X.to(dev), y.to(dev), n.to(dev)
model = use_linear_scorer(torch.nn.Linear(F, 1).to(dev))
loss_fn = L_.PairwiseHingeLoss()
fused = FusedLinearLoss(F, "hinge").to(dev)
def t(name, fn, reps=2000):
    for _ in range(50): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); print("%-46s %.1f us" % (name, (time.perf_counter() - t0) / reps * 1e6), flush=True)
t("model(X) -> LazyScores", lambda: model(X))
t("LazyScores(...) alone", lambda: LazyScores(X, model.weight, model.bias, None))
sc = model(X)
t("loss_fn(lazy)", lambda: loss_fn(model(X), y, n))
t("_LinearLossFunction.apply", lambda: _LinearLossFunction.apply(X, model.weight, model.bias, y, n, _C.HINGE, 1.0, False))
t("fused module forward", lambda: fused(X, y, n))
def full_lazy():
    model.weight.grad = None; model.bias.grad = None
    loss_fn(model(X), y, n).mean().backward()
def full_fused():
    fused.weight.grad = None; fused.bias.grad = None
    fused(X, y, n).mean().backward()
t("lazy: fwd + mean + backward", full_lazy)
t("fused module: fwd + mean + backward", full_fused)
t("lazy again", full_lazy)
