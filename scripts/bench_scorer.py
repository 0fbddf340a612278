#!/usr/bin/env python
"""LinearScorer vs torch.nn.Linear(F, 1) in the reference's user code
`loss_fn(model(xs), ys, n).mean().backward()` (hipGraph replay, per step), plus the two scorer
kernels alone against their HBM bytes."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import _benchutil as _bu  # noqa: E402
from pytorchltr_amd import _C  # noqa: E402
from pytorchltr_amd.fused import FusedLinearLoss, LinearScorer  # noqa: E402
from pytorchltr_amd.loss import PairwiseHingeLoss  # noqa: E402

dev = torch.device("cuda", 0)
lib = _C.lib()
for B, L, F in ((1024, 128, 136), (512, 512, 700), (256, 1000, 220), (32, 1000, 220)):
    scores, y, n, X = bench.synth(B, L, F, 0, dev)
    loss_fn = PairwiseHingeLoss()
    res = []
    for name, model, call in (("nn.Linear", torch.nn.Linear(F, 1).to(dev), lambda m: m(X)),
                              ("LinearScorer", LinearScorer(F).to(dev), lambda m: m(X, n))):
        ps = list(model.parameters())

        def step():
            for p_ in ps:
                p_.grad = None
            loss_fn(call(model), y, n).mean().backward()
        for _ in range(5):
            step()
        rp = bench.try_graph(step)
        t = _bu.time_wall(rp or step, 100, lambda: None) / 100 * 1e6
        res.append("%s + loss: %.1f us" % (name, t))
    fl = FusedLinearLoss(F, "hinge").to(dev)
    pf = list(fl.parameters())

    def fstep():
        for p_ in pf:
            p_.grad = None
        fl(X, y, n).mean().backward()
    for _ in range(5):
        fstep()
    rp = bench.try_graph(fstep)
    res.append("FusedLinearLoss: %.1f us" % (_bu.time_wall(rp or fstep, 100, lambda: None) / 100 * 1e6))
    W = torch.randn(F, device=dev)
    bias = torch.zeros(1, device=dev)
    sc = torch.empty(B, L, device=dev)
    g = torch.randn(B, L, device=dev) * (torch.arange(L, device=dev)[None, :] < n[:, None])
    out = torch.empty(F + 1, device=dev)
    wsb = lib.ltr_linear_grad_workspace_bytes(B, L, F)
    ws = torch.empty(wsb // 4, device=dev)
    k1 = lambda: _C.check(lib.ltr_linear_scores_f32(X.data_ptr(), W.data_ptr(), bias.data_ptr(), n.data_ptr(), B, L, F,
                                                    sc.data_ptr(), torch.cuda.current_stream().cuda_stream))
    k3 = lambda: _C.check(lib.ltr_linear_grad_f32(X.data_ptr(), g.data_ptr(), n.data_ptr(), B, L, F, out.data_ptr(),
                                                  ws.data_ptr(), wsb, torch.cuda.current_stream().cuda_stream))
    real = int(n.clamp(max=L).sum()) * F * 4
    for nm, fn in (("scores", k1), ("grad", k3)):
        for _ in range(5):
            fn()
        t, _ = _bu.time_launches(fn, per_graph=10, replays=10)
        res.append("%s kernel %.1f us (%.2f TB/s of real rows)" % (nm, t, real / (t * 1e-6) / 1e12))
    print("B=%d L=%d F=%d | " % (B, L, F) + " | ".join(res), flush=True)
