#!/usr/bin/env python
"""Randomised parity fuzz on the GPU: loss kernel (plain and workspace entry points) and the fused
Linear scorer + loss step against the fp64 C oracle, over random (kind, B, L, F, list-length
pattern, sigma).  Test infrastructure (uses oracle/); not part of the test tiers because it runs
for minutes (not collected by pytest):   python tests/fuzz_parity.py SEED SECONDS

Known benign report: "hinge ... grad ok False" on large shapes -- a pair whose margin is within one
fp32 ulp of 0 flips activity against the fp64 oracle."""
import sys, time, random
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from oracle import ltr_oracle as O
from pytorchltr_amd import _C
from pytorchltr_amd._autograd import pairwise_loss_and_grad
from pytorchltr_amd.fused import linear_loss_step
dev = torch.device("cuda:0")
KINDS = list(O.KINDS)
rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
fails = 0; cases = 0; t_end = time.time() + float(sys.argv[2]) if len(sys.argv) > 2 else time.time() + 120
def tol_loss(L): return (5e-4 if L > 256 else 2e-5)
while time.time() < t_end:
    mode = rnd.choice(["loss", "loss_ws", "fused"])
    kind = rnd.choice(KINDS)
    B = rnd.choice([1, 2, 7, 33, 64, 100, 257, 300, 600, 1024, 1100, 2048])
    L = rnd.choice([1, 2, 5, 17, 64, 65, 128, 129, 200, 256, 257, 300, 512, 700, 1000, 1025, 1100, 1251, 1500, 2048, 2049])
    if B * L > 700 * 1000: continue
    F = rnd.choice([4, 8, 12, 64, 136, 220])
    g = torch.Generator().manual_seed(rnd.randrange(1 << 30))
    s = torch.randn(B, L, generator=g); y = torch.randint(0, 5, (B, L), generator=g)
    pat = rnd.randrange(4)
    n = torch.randint(0, L + 1, (B,), generator=g)
    if pat == 1: n = torch.full((B,), L)
    if pat == 2: n = torch.where(torch.rand(B, generator=g) < 0.5, torch.full((B,), L), n)
    sigma = rnd.choice([1.0, 0.5, 2.0])
    cases += 1
    try:
        if mode in ("loss", "loss_ws"):
            want_l, want_g = O.pairwise_loss(kind, s.numpy(), y.numpy(), n.numpy(), sigma=sigma)
            loss, ds = pairwise_loss_and_grad(s.to(dev), y.to(dev), n.to(dev), getattr(_C, kind.upper()), sigma, cfg=("split" if mode == "loss_ws" else None))
            got_l = loss.cpu().numpy().astype(np.float64); got_g = ds.cpu().numpy().astype(np.float64)
            okl = np.all(np.abs(got_l - want_l) <= 2e-6 + tol_loss(L) * np.abs(want_l))
            scale = np.max(np.abs(want_g), axis=1, keepdims=True)
            okg = np.all(np.abs(got_g - want_g) <= 2e-5 * scale + 2e-6)
            if not (okl and okg and np.all(np.isfinite(got_l))):
                fails += 1; print("FAIL", mode, kind, B, L, "pat", pat, "sigma", sigma, "loss ok", okl, "grad ok", okg, flush=True)
        else:
            if B * L * F > 40e6 or (L > 300 and B > 300): continue
            X = torch.randn(B, L, F, generator=g); W = (torch.rand(F, generator=g) * 2 - 1) / F ** 0.5; b = torch.rand(1, generator=g) - 0.5
            gout = np.full(B, 1.0 / B)
            want_l, want_s, want_dW, want_db = O.linear_pairwise(kind, X.numpy(), W.numpy(), float(b[0]), y.numpy(), n.numpy(), gout, sigma=sigma)
            from pytorchltr_amd import loss as L_
            mod = {"hinge": L_.PairwiseHingeLoss, "dcg_hinge": L_.PairwiseDCGHingeLoss, "logistic": L_.PairwiseLogisticLoss, "arp1": L_.LambdaARPLoss1, "arp2": L_.LambdaARPLoss2, "ndcg1": L_.LambdaNDCGLoss1, "ndcg2": L_.LambdaNDCGLoss2}[kind]
            lm = mod() if kind in ("hinge", "dcg_hinge") else mod(sigma)
            loss, dW, db = linear_loss_step(X.to(dev), W.to(dev), b.to(dev), y.to(dev), n.to(dev), loss=lm)
            got_l = loss.cpu().numpy().astype(np.float64)
            ndcgk = kind.startswith("ndcg")
            okl = np.all(np.abs(got_l - want_l) <= 1e-5 + (3e-3 if ndcgk else tol_loss(L)) * np.abs(want_l))
            tol = (2e-3 if ndcgk else 2e-4) * max(1.0, float(np.max(np.abs(want_dW))))
            okw = np.max(np.abs(dW.cpu().numpy() - want_dW)) < tol and abs(float(db.cpu()[0]) - want_db) < tol
            if ndcgk and okw and not okl:
                # a rank-dependent loss on fp32 scores: two documents whose fp64 scores differ by less than an fp32 ulp may swap
                # ranks -- judge the losses on the scores the GPU computed
                sc = linear_loss_step(X.to(dev), W.to(dev), b.to(dev), y.to(dev), n.to(dev), loss=lm, return_scores=True)[-1]
                want_l2 = O.pairwise_loss(kind, sc.cpu().numpy().reshape(B, L), y.numpy(), n.numpy(), sigma=sigma)[0]
                okl = np.all(np.abs(got_l - want_l2) <= 1e-5 + 3e-3 * np.abs(want_l2))
                if okl: print("note fused", kind, B, L, F, "rank flips against the fp64 scores; equal on the GPU's scores", flush=True)
            if not (okl and okw and np.all(np.isfinite(got_l))):
                fails += 1; print("FAIL fused", kind, B, L, F, "pat", pat, "sigma", sigma, "loss ok", okl, "dW ok", okw, "plan", _C.lib().ltr_linear_fused_plan(getattr(_C, kind.upper()), B, L, F), flush=True)
    except Exception as exc:
        fails += 1; print("EXC", mode, kind, B, L, F, repr(exc)[:200], flush=True)
print("cases %d fails %d" % (cases, fails))
