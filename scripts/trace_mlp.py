#!/usr/bin/env python
"""Per-phase cycle breakdown of the fused MLP kernel (needs a -DLTR_MLP_TRACE build):
    hipcc ... -DLTR_MLP_TRACE -o build/variants/libltr_mlptrace.so pytorchltr_amd/csrc/ltr_kernels.hip
    python scripts/trace_mlp.py [--full-lists] [--kind hinge]"""
import argparse
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from pytorchltr_amd import _C  # noqa: E402

PHASES = ["weights->LDS", "labels", "layer1 MFMA", "layers 2,3 + barrier", "pair pass", "dH2/dW2/dH1",
          "dH1->LDS", "dW1 MFMA", "fold", "write partial", "(sched) n loaded", "(sched) vote+hist",
          "(sched) prefix", "(sched) rank"]


PHASES_TILE = ["rest of prologue", "(fwd) next fill issue + barrier", "layer-1 tile + L2 partial", "owner: layers 2,3",
               "pair pass", "bwd: refill + dH2 rows", "bwd GEMMs (dW2, dH1, dW1)", "partial vector",
               "(prologue) fragment loads", "(prologue) scheduling", "(fwd) owner + parking of fill before",
               "(fwd) wait for P + image write", "(tuning) scheduling, second run", "(fwd) owner rows", "(fwd) parking"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layout", default="tile", choices=["tile", "wide"],
                    help="tile = 4-wave kernel of ltr_mlp2.inc (default dispatch), wide = LTR_MLP_LAYOUT=1")
    ap.add_argument("--lib", default=os.path.join(ROOT, "build", "variants", "libltr_mlptrace.so"))
    ap.add_argument("--kind", default="hinge")
    ap.add_argument("--full-lists", action="store_true")
    ap.add_argument("--B", type=int, default=1024)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    if args.layout == "wide":
        os.environ["LTR_MLP_LAYOUT"] = "1"
    lib = ctypes.CDLL(args.lib)
    for name, (res, argt) in _C.SIGNATURES.items():
        getattr(lib, name).restype = res
        getattr(lib, name).argtypes = argt
    B, L, F, H1, H2 = args.B, 128, 136, 50, 10
    _, rel, n, X = bench.synth(B, L, F, 0, dev)
    if args.full_lists:
        n = torch.full_like(n, L)
    g = torch.Generator().manual_seed(0)
    params = [(torch.rand(*s, generator=g) - 0.5).to(dev) for s in ((H1, F), (H1,), (H2, H1), (H2,), (1, H2), (1,))]
    P = lib.ltr_mlp_param_count(F, H1, H2)
    wsb = lib.ltr_mlp_workspace_bytes(B, F, H1, H2)
    ws = torch.empty(wsb // 4, device=dev)
    grads = torch.empty(P, device=dev)
    loss = torch.empty(B, device=dev)
    grid = wsb // (4 * P)
    for it in range(3):
        trace = torch.zeros(grid * 16, dtype=torch.int64, device=dev)
        rc = lib.ltr_mlp_pairwise_f32(getattr(_C, args.kind.upper()), 1.0, X.data_ptr(), *[p.data_ptr() for p in params],
                                      rel.data_ptr(), 0, n.data_ptr(), None, B, L, F, H1, H2, loss.data_ptr(),
                                      trace.data_ptr(), grads.data_ptr(), None, ws.data_ptr(), wsb,
                                      torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc
        torch.cuda.synchronize()
    t = trace.view(grid, 16).double().cpu()
    t = t[t.sum(1) > 0]                      # (the workspace is sized for the larger grid)
    grid = t.shape[0]
    tot = t[:, :12].sum(1) + t[:, 13:15].sum(1)
    print("workgroups %d, queries/workgroup %.1f; total cycles/workgroup mean %.0f max %.0f" % (
        grid, B / grid, tot.mean(), tot.max()))
    for i, name in enumerate(PHASES_TILE if args.layout == "tile" else PHASES):
        print("  %-24s mean %9.0f  (%5.1f%%)   max %9.0f" % (name, t[:, i].mean(), 100 * t[:, i].mean() / tot.mean(), t[:, i].max()))


if __name__ == "__main__":
    main()
