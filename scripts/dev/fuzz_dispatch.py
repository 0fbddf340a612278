#!/usr/bin/env python
"""Round-4 fuzz of the fused Linear scorer + loss step over the regimes whose dispatch changed (wide rows on the parts kernel
incl. short lists and the NDCG kinds, the 19- / 24-sweep register tiles, the NDCG kinds on the cluster kernel): random
(kind, B, L, F, list-length pattern) against the fp64 C oracle; the plan the dispatcher picked is reported.
Test infrastructure (uses oracle/), not collected by pytest:   python scripts/dev/fuzz_dispatch.py SEED SECONDS"""
import sys, time, random, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import ltr_oracle as O
from pytorchltr_amd import _C
from pytorchltr_amd.fused import linear_loss_step
dev = torch.device("cuda:0")
KINDS = list(O.KINDS)
rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
t_end = time.time() + (float(sys.argv[2]) if len(sys.argv) > 2 else 120)
cases = bad = 0
plans = {}
while time.time() < t_end:
    regime = rnd.choice(["wide", "wide_short", "rt1024", "ndcg_cluster", "ndcg_parts", "hinge_cluster", "hinge_cluster"])
    kind = rnd.choice(KINDS)
    if regime == "wide":
        B, L, F = rnd.choice([65, 130, 200, 257, 300, 384, 520]), rnd.choice([300, 400, 512, 640, 768, 1000]), rnd.choice([448, 512, 576, 640, 700])
    elif regime == "wide_short":
        B, L, F = rnd.choice([512, 600, 1024]), rnd.choice([100, 128, 200, 256]), rnd.choice([640, 700, 764])
    elif regime == "rt1024":
        B, L, F = rnd.choice([3, 64, 257, 600]), rnd.choice([129, 150, 172, 181, 200, 216, 217, 255, 256]), rnd.choice([100, 120, 136, 160, 220, 300])
    elif regime == "hinge_cluster":      # round 5: the hinge kinds by sorted runs (light members, also on large batches of the longest lists)
        kind = rnd.choice(["hinge", "dcg_hinge"])
        B, L, F = rnd.choice([1, 7, 33, 64, 100, 200, 256, 300, 384]), rnd.choice([257, 300, 512, 700, 999, 1000, 1001, 1024]), rnd.choice([16, 64, 136, 220, 384, 520, 700])
    elif regime == "ndcg_cluster":
        kind = rnd.choice(["ndcg1", "ndcg2"])
        B, L, F = rnd.choice([5, 33, 64, 100, 200, 256, 380]), rnd.choice([257, 300, 512, 700, 1000, 1024]), rnd.choice([16, 64, 136, 220, 384])
    else:
        kind = rnd.choice(["ndcg1", "ndcg2"])
        B, L, F = rnd.choice([70, 128, 190, 256, 400]), rnd.choice([400, 512, 600, 768, 1000]), rnd.choice([448, 512, 640, 700, 764])
    if B * L * F > 110_000_000:
        B = max(1, 110_000_000 // (L * F))
    g = torch.Generator().manual_seed(rnd.randrange(1 << 30))
    y = torch.randint(0, 5, (B, L), generator=g)
    X = torch.randn(B, L, F, generator=g)
    W = (torch.rand(F, generator=g) * 2 - 1) / F ** 0.5
    b = torch.randn(1, generator=g) * 0.1
    pat = rnd.randrange(4)
    n = torch.randint(0, L + 1, (B,), generator=g)
    if pat == 1: n = torch.full((B,), L)
    if pat == 2: n = torch.where(torch.rand(B, generator=g) < 0.5, torch.full((B,), L), n)
    if pat == 3: n = torch.clamp(n, max=max(1, L // 3))
    gout = torch.rand(B, generator=g) + 0.5 if rnd.random() < 0.3 else None
    plan = _C.lib().ltr_linear_fused_plan(getattr(_C, kind.upper()), B, L, F)
    plans[(regime, plan)] = plans.get((regime, plan), 0) + 1
    cases += 1
    if os.environ.get("FUZZ_VERBOSE"):
        print("CASE", cases, regime, kind, (B, L, F), "pat", pat, "gout", gout is not None, "plan", plan, flush=True)
    try:
        outs = []
        for rep in range(2):
            loss, dW, db = linear_loss_step(X.to(dev), W.to(dev), b.to(dev), y.to(dev), n.to(dev), loss=kind, grad_out=None if gout is None else gout.to(dev))
            outs.append((loss.cpu().numpy(), dW.cpu().numpy(), db.cpu().numpy()))
        _C.device_status()
        want_l, _, want_dW, want_db = O.linear_pairwise(kind, X.numpy(), W.numpy(), float(b[0]), y.numpy(), n.numpy(),
                                                        np.full(B, 1.0 / B) if gout is None else gout.numpy().astype(np.float64))
        loss, dW, db = outs[0]
        rtol = 5e-4 if L > 256 else 2e-5
        okr = np.isclose(loss, want_l, rtol=rtol, atol=1e-5)
        if kind in ("ndcg1", "ndcg2"):
            loss_ok = okr.mean() >= 0.97 and np.allclose(loss, want_l, rtol=5e-3, atol=1e-5)
        else:
            loss_ok = bool(okr.all())
        scale = max(1.0, float(np.max(np.abs(want_dW))))
        tol = (2e-4 if L > 256 else 2e-5) * scale * (8 if kind in ("ndcg1", "ndcg2") else 1)       # (rank swaps of nearly tied fp32 scores against the fp64 oracle move dW too)
        g_ok = np.max(np.abs(dW - want_dW)) < tol and abs(float(db[0]) - want_db) < tol
        same = all(np.array_equal(a, c) for a, c in zip(outs[0], outs[1]))
        if not (loss_ok and g_ok and same and np.all(np.isfinite(loss))):
            bad += 1
            print("BAD", regime, kind, (B, L, F), "pat", pat, "gout", gout is not None, "plan", plan, "loss_ok", loss_ok, "bad rows", int((~okr).sum()),
                  "dW err", float(np.max(np.abs(dW - want_dW))), "tol", tol, "bit-identical", same, flush=True)
    except Exception as exc:
        bad += 1
        print("EXC", regime, kind, (B, L, F), pat, repr(exc)[:300], flush=True)
print("cases", cases, "bad", bad, "plans", sorted(plans.items()))
