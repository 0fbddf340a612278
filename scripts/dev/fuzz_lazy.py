#!/usr/bin/env python
"""Fuzz of the lazy entry points (round 6): random (kind, B, L, F, list-length pattern); four steps + flushes through
ltr_linear_sgd_lazy_step_f32 against ltr_linear_sgd_step_f32 (bit for bit up to 1024 queries, to rounding and run-to-run
identical beyond), and the same batches through pytorchltr_amd.optim.SGD against torch.optim.SGD.
Test infrastructure, not collected by pytest:   python scripts/dev/fuzz_lazy.py SEED SECONDS"""
import sys, time, random, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pytorchltr_amd import _C
from pytorchltr_amd import loss as L_
from pytorchltr_amd.fused import use_linear_scorer
from pytorchltr_amd.optim import SGD
dev = torch.device("cuda:0")
lib = _C.lib()
KINDS = ["hinge", "dcg_hinge", "logistic", "arp1", "arp2", "ndcg1", "ndcg2"]
CLS = {"hinge": L_.PairwiseHingeLoss, "dcg_hinge": L_.PairwiseDCGHingeLoss, "logistic": L_.PairwiseLogisticLoss, "arp1": L_.LambdaARPLoss1,
       "arp2": L_.LambdaARPLoss2, "ndcg1": L_.LambdaNDCGLoss1, "ndcg2": L_.LambdaNDCGLoss2}
rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
t_end = time.time() + (float(sys.argv[2]) if len(sys.argv) > 2 else 120)
cases = bad = 0
while time.time() < t_end:
    kind = rnd.choice(KINDS)
    B = rnd.choice([1, 2, 7, 64, 255, 256, 300, 777, 1023, 1024, 1025, 1500, 2500, 4096, 5000])
    L = rnd.choice([1, 2, 17, 40, 64, 65, 100, 128, 135, 160, 200, 256])
    F = rnd.choice([4, 8, 24, 64, 100, 136, 220, 300, 5, 45, 46])
    if B * L * F > 6e7:
        continue
    g = torch.Generator().manual_seed(rnd.randrange(1 << 30))
    bat = []
    for i in range(2):
        X = torch.randn(B, L, F, generator=g)
        y = torch.randint(0, 5, (B, L), generator=g)
        pat = rnd.choice(["uniform", "full", "short", "zero_some"])
        n = {"uniform": torch.randint(1, L + 1, (B,), generator=g), "full": torch.full((B,), L),
             "short": torch.randint(1, min(L, 8) + 1, (B,), generator=g), "zero_some": torch.randint(0, L + 1, (B,), generator=g)}[pat]
        bat.append([t.to(dev) for t in (X, y, n)])
    W0 = (torch.rand(F, generator=g) - 0.5) / F ** 0.5
    b0 = torch.zeros(1)
    kid = getattr(_C, kind.upper())
    lr = 0.01
    st = _C.stream_of(bat[0][0])
    nws = lib.ltr_linear_workspace_bytes(B, L, F)

    def run(lazy):
        Wd, bd = W0.clone().to(dev), b0.clone().to(dev)
        ws = torch.full((nws // 4 + 64,), float("nan"), device=dev)
        loss = torch.empty(B, device=dev); bucket = torch.zeros(F + 2, device=dev)
        out = []; pending = 0
        for k in range(4):
            Xd, yd, nd = bat[k % 2]
            if lazy:
                _C.check(lib.ltr_linear_sgd_lazy_step_f32(kid, 1.0, Xd.data_ptr(), Wd.data_ptr(), bd.data_ptr(), yd.data_ptr(), _C.LABEL_I64, nd.data_ptr(), B, L, F, lr, loss.data_ptr(), bucket.data_ptr(), ws.data_ptr(), ws.numel() * 4, pending, st))
                pending = B
                out.append(loss.clone())
                if k in (1, 3):
                    _C.check(lib.ltr_linear_sgd_flush_f32(kid, Wd.data_ptr(), bd.data_ptr(), pending, L, F, lr, loss.data_ptr(), bucket.data_ptr(), ws.data_ptr(), st)); pending = 0
                    out.append(Wd.clone()); out.append(bucket.clone())
            else:
                _C.check(lib.ltr_linear_sgd_step_f32(kid, 1.0, Xd.data_ptr(), Wd.data_ptr(), bd.data_ptr(), yd.data_ptr(), _C.LABEL_I64, nd.data_ptr(), None, B, L, F, lr, loss.data_ptr(), bucket.data_ptr(), ws.data_ptr(), ws.numel() * 4, None, st))
                out.append(loss.clone())
                if k in (1, 3):
                    out.append(Wd.clone()); out.append(bucket.clone())
        torch.cuda.synchronize()
        if lib.ltr_device_status(1) != 0:
            return None
        return [t.cpu().numpy() for t in out]
    e, l1, l2 = run(False), run(True), run(True)
    ok = e is not None and l1 is not None and l2 is not None
    if ok:
        for a, b, c in zip(e, l1, l2):
            fin = np.isfinite(a).all()
            same = np.array_equal(a, b) if B <= 1024 else np.allclose(a, b, rtol=2e-4, atol=2e-5 * max(1.0, float(np.abs(a).max())))
            ok = ok and fin and same and np.array_equal(b, c)
    # the optimizer path
    if ok and L <= 200:
        torch.manual_seed(1)
        lin = torch.nn.Linear(F, 1).to(dev)
        ma, mb = use_linear_scorer(copy.deepcopy(lin)), use_linear_scorer(copy.deepcopy(lin))
        oa, ob = torch.optim.SGD(ma.parameters(), lr=lr), SGD(mb.parameters(), lr=lr)
        fn = CLS[kind]()
        for m, o in ((ma, oa), (mb, ob)):
            for k in range(3):
                Xd, yd, nd = bat[k % 2]
                ls = fn(m(Xd), yd, nd).mean(); o.zero_grad(); ls.backward(); o.step()
        wa, wb = ma.weight.detach().cpu().numpy(), mb.weight.detach().cpu().numpy()
        ok = np.isfinite(wb).all() and np.allclose(wa, wb, rtol=2e-4, atol=2e-5 * max(1.0, float(np.abs(wa).max())))
    cases += 1
    if not ok:
        bad += 1
        print("MISMATCH", kind, B, L, F, pat, flush=True)
print("cases", cases, "bad", bad)
