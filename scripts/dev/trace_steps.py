"""Phase trace of the persistent multi-batch launch (ltr_debug_steps_trace): wall-clock stamps per workgroup and step.
stamps: 0 step start (weights in LDS)  2 row published  3 next tile requested (non-reducer CUs)
        4 reducer: all rows in  5 reducer: weights unit published  6 weights of the step received
python scripts/dev/trace_steps.py [B L F kind]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pytorchltr_amd import _C  # noqa: E402
from tests.conftest import synth  # noqa: E402


def main():
    B, L, F = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (1024, 128, 136)
    kind = sys.argv[4] if len(sys.argv) > 4 else "hinge"
    K = 16
    dev = torch.device("cuda:0")
    lib = _C.lib()
    kid = getattr(_C, kind.upper())
    nrot = 6
    bat = []
    for i in range(nrot):
        s, y, n, X, W, b = synth(B, L, i, F=F)
        bat.append((X.to(dev), y.to(dev), n.to(dev)))
    Wd, bd = W.to(dev), b.to(dev)
    ws = torch.empty(lib.ltr_linear_workspace_bytes(B, L, F) // 4 + 64, device=dev)
    loss = torch.empty(K, B, device=dev)
    bucket = torch.empty(K, F + 2, device=dev)
    G = B
    trace = torch.zeros(K * G * 8, dtype=torch.int64, device=dev)
    Parr = ctypes.c_void_p * K
    xp = Parr(*[bat[k % nrot][0].data_ptr() for k in range(K)])
    rp = Parr(*[bat[k % nrot][1].data_ptr() for k in range(K)])
    np_ = Parr(*[bat[k % nrot][2].data_ptr() for k in range(K)])
    st = torch.cuda.current_stream().cuda_stream
    for rep in range(2):
        if rep == 1:
            lib.ltr_debug_steps_trace(trace.data_ptr())
        _C.check(lib.ltr_linear_sgd_steps_f32(kid, 1.0, K, xp, rp, _C.LABEL_I64, np_, B, L, F, 1e-4, Wd.data_ptr(), bd.data_ptr(),
                                              loss.data_ptr(), bucket.data_ptr(), ws.data_ptr(), ws.numel() * 4, st))
        torch.cuda.synchronize()
    lib.ltr_debug_steps_trace(None)
    t = trace.cpu().numpy().reshape(K, G, 8).astype(np.float64) / 100.0       # us
    starts = np.array([t[k, :, 0].min() for k in range(K)])
    print("step starts (first workgroup), deltas us:", np.round(np.diff(starts), 2).tolist())
    names = ["start", "-", "row out", "tile requested", "rows in (reducer)", "weights out (reducer)", "weights received"]
    n_all = [bat[k % nrot][2].cpu().numpy() for k in range(K)]
    for k in (6, 7, 8):
        t0 = starts[k]
        print("step %d (us after the first workgroup's start):" % k)
        for i, nm in enumerate(names):
            v = t[k, :, i]
            v = v[v > 0] - t0
            if v.size:
                print("   %-24s n=%4d  min %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f" % (nm, v.size, v.min(), np.median(v), np.percentile(v, 90), v.max()))
        red = t[k, :, 4] > 0
        if red.any():
            print("   reducers: own row out at", np.round(np.sort(t[k, red, 2] - t0)[[0, -1]], 2).tolist(), " positions", np.nonzero(red)[0][:8].tolist(), "...")
        late = np.argsort(-t[k, :, 2])[:6]
        print("   last rows out: positions", late.tolist(), "at", np.round(t[k, late, 2] - t0, 2).tolist())


if __name__ == "__main__":
    main()
