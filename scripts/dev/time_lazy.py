"""Steady-state time per synchronous-SGD step: ltr_linear_sgd_step_f32 (two launches) against ltr_linear_sgd_lazy_step_f32
(the update of step k inside step k + 1's launch) + one flush per timed region.  python scripts/dev/time_lazy.py [workload ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from pytorchltr_amd import _C
dev = torch.device("cuda:0")
lib = _C.lib()
for case in (sys.argv[1:] or ["hinge:1024x128x136", "ndcg2:1024x128x136", "hinge:2048x128x136", "logistic:1024x100x220", "hinge:512x128x136"]):
    kind, shp = case.split(":")
    B, L, F = (int(v) for v in shp.split("x"))
    nbuf = bench.nbuf_for(B, L, F)
    bat = bench.make_batches(B, L, F, nbuf, 0, dev)
    fs = bench.FusedStep(kind, B, L, F, dev)
    st = torch.cuda.current_stream().cuda_stream
    state = {"pending": 0}
    def eager(i):
        fs.sgd_step(bat[i % nbuf])
    def lazy(i):
        b = bat[i % nbuf]
        _C.check(lib.ltr_linear_sgd_lazy_step_f32(fs.kind_id, 1.0, b["X"].data_ptr(), fs.W.data_ptr(), fs.bias.data_ptr(), b["rel"].data_ptr(),
                                                  _C.LABEL_I64, b["n"].data_ptr(), B, L, F, bench.SGD_LR, fs.lossv.data_ptr(), fs.flat.data_ptr(),
                                                  fs.part.data_ptr(), fs.ws_bytes, state["pending"], st))
        state["pending"] = int(os.environ.get("LTR_TIME_LAZY_PEND", B))      # (timing experiment: a pending batch of fewer rows)
    def flush():
        _C.check(lib.ltr_linear_sgd_flush_f32(fs.kind_id, fs.W.data_ptr(), fs.bias.data_ptr(), state["pending"], L, F, bench.SGD_LR, fs.lossv.data_ptr(),
                                              fs.flat.data_ptr(), fs.part.data_ptr(), st))
        state["pending"] = 0
    out = [case]
    for name, fn, fin in (("eager", eager, None), ("lazy", lazy, flush), ("eager", eager, None), ("lazy", lazy, flush)):
        for i in range(50): fn(i)
        if fin: fin()
        torch.cuda.synchronize()
        ts = []
        K = 4000
        for rep in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(K): fn(i)
            if fin: fin()
            th = time.perf_counter()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / K * 1e6)
            hs = (th - t0) / K * 1e6
        out.append("%s %.2f/%.2f (host %.2f)" % (name, min(ts), sorted(ts)[2], hs))
    _C.device_status()
    print(" | ".join(out), flush=True)
