#!/usr/bin/env python
"""Dev check of the parts kernel: parity vs the oracle on every row + kernel time vs the other plans.
python scripts/dev/parts_check.py [--shapes B,L,F,kind ...] [--time]"""
import argparse, os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import numpy as np
import torch
from oracle import ltr_oracle as O
from tests.conftest import synth
from pytorchltr_amd import _C
from _benchutil import time_launches

ap = argparse.ArgumentParser()
ap.add_argument("--shapes", nargs="*", default=["6,512,700,hinge", "6,1000,220,dcg_hinge", "40,300,64,logistic",
                                                "33,600,136,arp1", "20,700,220,arp2", "512,512,700,hinge",
                                                "256,1000,220,dcg_hinge"])
ap.add_argument("--time", action="store_true")
ap.add_argument("--nocheck", action="store_true")
args = ap.parse_args()
dev = torch.device("cuda:0")
lib = _C.lib()
for spec in args.shapes:
    B, L, F, kind = spec.split(",")
    B, L, F = int(B), int(L), int(F)
    s, y, n, X, W, b = synth(B, L, 5, F=F)
    Xd, Wd, bd, yd, nd = X.to(dev), W.to(dev), b.to(dev), y.to(dev), n.to(dev)
    k = O.KINDS[kind]
    plan = lib.ltr_linear_fused_plan(k, B, L, F)
    ws = torch.empty(lib.ltr_linear_workspace_bytes(B, L, F) // 4 + 64, device=dev)
    loss = torch.empty(B, device=dev)
    dW = torch.empty(F, device=dev); db = torch.empty(1, device=dev)
    def step():
        st = torch.cuda.current_stream().cuda_stream
        rc = lib.ltr_linear_partials_f32(k, 1.0, Xd.data_ptr(), Wd.data_ptr(), bd.data_ptr(), yd.data_ptr(), 0,
                                         nd.data_ptr(), B, L, F, loss.data_ptr(), None, ws.data_ptr(), st)
        assert rc == 0, rc
    out = {"shape": spec, "plan": plan}
    if not args.nocheck:
        for rep in range(3):
            step()
            rc = lib.ltr_linear_reduce_f32(ws.data_ptr(), None, B, F, dW.data_ptr(), db.data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            want_l, _, want_dW, want_db = O.linear_pairwise(kind, X.numpy(), W.numpy(), float(b[0]), y.numpy(), n.numpy(),
                                                            np.full(B, 1.0 / B))
            got_l = loss.cpu().numpy()
            rtol = 5e-4 if L > 256 else 2e-5
            okl = np.allclose(got_l, want_l, rtol=rtol, atol=1e-5)
            tol = 2e-5 * max(1.0, float(np.max(np.abs(want_dW)))) * 10
            errW = float(np.max(np.abs(dW.cpu().numpy() - want_dW)))
            errb = abs(float(db.cpu()[0]) - want_db)
            out["rep%d" % rep] = {"loss_ok": bool(okl), "dW_err": errW, "tol": tol, "db_err": errb,
                                  "bad_rows": int(np.sum(~np.isclose(got_l, want_l, rtol=rtol, atol=1e-5)))}
            if rep == 0:
                first = (got_l.copy(), dW.cpu().numpy().copy())
            else:
                out["rep%d" % rep]["bit_identical"] = bool(np.array_equal(got_l, first[0]) and np.array_equal(dW.cpu().numpy(), first[1]))
        out["status"] = lib.ltr_device_status(0)
    if args.time:
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        tt, _ = time_launches(step, per_graph=10, replays=5)
        out["kernel_us"] = round(tt, 2)
    print(json.dumps(out), flush=True)
