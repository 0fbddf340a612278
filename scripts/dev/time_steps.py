"""Cold timing of the persistent multi-batch launch against the per-step path (C2: 1024 x 128 x 136 hinge, 5 rotating
batches = 357 MB > the 256 MiB Infinity Cache).  python scripts/dev/time_steps.py [B L F kind [K]]"""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pytorchltr_amd import _C  # noqa: E402
from tests.conftest import synth  # noqa: E402


def main():
    B, L, F = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (1024, 128, 136)
    kind = sys.argv[4] if len(sys.argv) > 4 else "hinge"
    K = int(sys.argv[5]) if len(sys.argv) > 5 else 640
    full = os.environ.get("FULL") == "1"
    dev = torch.device("cuda:0")
    lib = _C.lib()
    kid = getattr(_C, kind.upper())
    nrot = max(2, min(8, int(400e6 // (B * L * F * 4)) + 1))
    bat = []
    for i in range(nrot):
        s, y, n, X, W, b = synth(B, L, i, F=F)
        if full:
            n = torch.full_like(n, L)
        bat.append((X.to(dev), y.to(dev), n.to(dev)))
    W0, b0 = W.to(dev), b.to(dev)
    print("plan", lib.ltr_linear_sgd_steps_plan(kid, B, L, F), "rotation", nrot, "batches")
    ws = torch.empty(lib.ltr_linear_workspace_bytes(B, L, F) // 4 + 64, device=dev)
    loss = torch.empty(K, B, device=dev)
    bucket = torch.empty(K, F + 2, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    lr = 1e-4
    Parr = ctypes.c_void_p * K
    xp = Parr(*[bat[k % nrot][0].data_ptr() for k in range(K)])
    rp = Parr(*[bat[k % nrot][1].data_ptr() for k in range(K)])
    np_ = Parr(*[bat[k % nrot][2].data_ptr() for k in range(K)])

    def persistent(Wd, bd):
        _C.check(lib.ltr_linear_sgd_steps_f32(kid, 1.0, K, xp, rp, _C.LABEL_I64, np_, B, L, F, lr, Wd.data_ptr(), bd.data_ptr(),
                                              loss.data_ptr(), bucket.data_ptr(), ws.data_ptr(), ws.numel() * 4, st))

    def per_step(Wd, bd):
        for k in range(K):
            X, y, n = bat[k % nrot]
            rc = lib.ltr_linear_sgd_step_f32(kid, 1.0, X.data_ptr(), Wd.data_ptr(), bd.data_ptr(), y.data_ptr(), _C.LABEL_I64,
                                             n.data_ptr(), None, B, L, F, lr, loss.data_ptr(), bucket.data_ptr(), ws.data_ptr(),
                                             ws.numel() * 4, None, st)
            if rc:
                _C.check(rc)

    res = {}
    for name, fn in (("per_step", per_step), ("persistent", persistent), ("per_step", per_step), ("persistent", persistent)):
        Wd, bd = W0.clone(), b0.clone()
        fn(Wd, bd)
        torch.cuda.synchronize()
        ts = []
        for rep in range(5):
            Wd, bd = W0.clone(), b0.clone()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn(Wd, bd)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / K * 1e6)
        ts.sort()
        res.setdefault(name, []).append(ts[2])
        print("%-10s %.2f us / step (median of 5 x %d steps; min %.2f)" % (name, ts[2], K, ts[0]), "W[0..2]", Wd[:3].tolist(), flush=True)
    _C.device_status()
    print("RESULT per_step %.2f persistent %.2f" % (min(res["per_step"]), min(res["persistent"])))


if __name__ == "__main__":
    main()
