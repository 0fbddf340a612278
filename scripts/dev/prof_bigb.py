#!/usr/bin/env python
"""The synchronous-SGD step at a batch of many rounds, launch by launch (graph replays, device clock): which launch holds the
time the B-sweep shows between the fused kernel and the step (profiles/r06_sweep_b.jsonl).  EAGER launches between two events
(under graph capture the lazy flush runs one reducer per column group: its tags are a host-side counter)."""
import os, sys, json, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from _benchutil import time_launches
from pytorchltr_amd import _C
dev = torch.device("cuda:0")
lib = _C.lib()
L, F = 128, 136
kind = _C.HINGE
for B in [int(a) for a in sys.argv[1:]] or [4096, 65536]:
    g = torch.Generator().manual_seed(0)
    rel = torch.randint(0, 5, (B, L), generator=g).to(dev)
    n = torch.randint(1, L + 1, (B,), generator=g).to(dev)
    X = torch.randn(B, L, F, device=dev)
    W = torch.randn(F, device=dev) * 0.1
    bias = torch.zeros(1, device=dev)
    loss = torch.empty(B, device=dev)
    part = torch.empty(lib.ltr_linear_workspace_bytes(B, L, F) // 4, device=dev)
    bucket = torch.zeros(F + 2, device=dev)
    cs = lambda: torch.cuda.current_stream().cuda_stream

    def fused():
        _C.check(lib.ltr_linear_partials_f32(kind, 1.0, X.data_ptr(), W.data_ptr(), bias.data_ptr(), rel.data_ptr(), 0, n.data_ptr(),
                                             B, L, F, loss.data_ptr(), None, part.data_ptr(), cs()))

    def two():
        _C.check(lib.ltr_linear_sgd_step_f32(kind, 1.0, X.data_ptr(), W.data_ptr(), bias.data_ptr(), rel.data_ptr(), 0, n.data_ptr(), None,
                                             B, L, F, 1e-6, loss.data_ptr(), bucket.data_ptr(), part.data_ptr(), part.numel() * 4, None, cs()))

    def lazy0():       # the lazy step's launch with nothing pending: the plain kernel writing rows in the lazy layout
        _C.check(lib.ltr_linear_sgd_lazy_step_f32(kind, 1.0, X.data_ptr(), W.data_ptr(), bias.data_ptr(), rel.data_ptr(), 0, n.data_ptr(),
                                                  B, L, F, 1e-6, loss.data_ptr(), bucket.data_ptr(), part.data_ptr(), part.numel() * 4, 0, cs()))

    def lazyB():
        _C.check(lib.ltr_linear_sgd_lazy_step_f32(kind, 1.0, X.data_ptr(), W.data_ptr(), bias.data_ptr(), rel.data_ptr(), 0, n.data_ptr(),
                                                  B, L, F, 1e-6, loss.data_ptr(), bucket.data_ptr(), part.data_ptr(), part.numel() * 4, B, cs()))

    def flush():
        _C.check(lib.ltr_linear_sgd_flush_f32(kind, W.data_ptr(), bias.data_ptr(), B, L, F, 1e-6, loss.data_ptr(), bucket.data_ptr(),
                                              part.data_ptr(), cs()))
    row = dict(B=B)
    for name, fn in ((("lazy_launch_nothing_pending", lazy0), ("lazy_step", lazyB)) if os.environ.get("LAZY_ONLY") else (("fused", fused), ("two_launch_step", two), ("lazy_launch_nothing_pending", lazy0), ("flush", flush), ("lazy_step", lazyB))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        best = 1e30
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _i in range(20):
                fn()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / 20)
        row[name + "_us"] = round(best, 2)
    print(json.dumps(row), flush=True)
    del X
    torch.cuda.empty_cache()
