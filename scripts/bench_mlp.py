#!/usr/bin/env python
"""Times the fused MLP scorer + loss step (ltr_mlp_pairwise_f32) next to the unfused composition
(torch.nn.Linear layers on rocBLAS + the HIP loss kernel + autograd) on one GPU.

    python scripts/bench_mlp.py [--B 1024 --L 128 --F 136 --kind hinge --full-lists]

Prints one JSON line: per-launch time (hipGraph-batched, as bench.py's time_launches), queries/s,
useful TFLOP/s against the 157.3 TF f32-MFMA peak, and HBM GB/s of the feature stream."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import _benchutil as _bu  # noqa: E402

F32_MFMA_PEAK_TF = 157.3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=1024)
    ap.add_argument("--L", type=int, default=128)
    ap.add_argument("--F", type=int, default=136)
    ap.add_argument("--H1", type=int, default=50)
    ap.add_argument("--H2", type=int, default=10)
    ap.add_argument("--kind", default="hinge")
    ap.add_argument("--full-lists", action="store_true")
    ap.add_argument("--no-unfused", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    from pytorchltr_amd import _C
    from pytorchltr_amd import loss as L_
    from pytorchltr_amd.fused import FusedMLPLoss
    B, L, F, H1, H2 = args.B, args.L, args.F, args.H1, args.H2
    scores, relevance, n, X = bench.synth(B, L, F, 0, dev)
    if args.full_lists:
        n = torch.full_like(n, L)
    torch.manual_seed(0)
    m = FusedMLPLoss(F, args.kind, hidden=(H1, H2)).to(dev)
    lib = _C.lib()
    kind_id = getattr(_C, args.kind.upper())
    P = lib.ltr_mlp_param_count(F, H1, H2)
    ws_bytes = lib.ltr_mlp_workspace_bytes(B, F, H1, H2)
    ws = torch.empty(ws_bytes // 4, device=dev)
    grads = torch.empty(P, device=dev)
    lossv = torch.empty(B, device=dev)
    lsum = torch.zeros(1, device=dev)
    params = [p.detach().contiguous() for p in (m.l1.weight, m.l1.bias, m.l2.weight, m.l2.bias,
                                                m.l3.weight, m.l3.bias)]

    def launch():
        _C.check(lib.ltr_mlp_pairwise_f32(
            kind_id, 1.0, X.data_ptr(), *[p.data_ptr() for p in params], relevance.data_ptr(),
            _C.LABEL_I64, n.data_ptr(), None, B, L, F, H1, H2, lossv.data_ptr(), None,
            grads.data_ptr(), lsum.data_ptr(), ws.data_ptr(), ws_bytes,
            torch.cuda.current_stream().cuda_stream))

    for _ in range(5):
        launch()
    torch.cuda.synchronize()
    us, graphed = _bu.time_launches(launch, per_graph=10, replays=10)
    docs = int(n.clamp(max=L).sum())
    flops_per_doc = 2 * (F * H1 + H1 * H2 + H2) + 2 * (F * H1 + 2 * H1 * H2 + H2)   # fwd + bwd (no dX)
    out = {
        "workload": "MLP %d-%d-%d-1 + %s, B=%d, L=%d%s" % (F, H1, H2, args.kind, B, L,
                                                           " (full lists)" if args.full_lists else ""),
        "fused_step_us": us, "graphed": graphed, "queries_per_s": B / (us * 1e-6),
        "useful_tflops": docs * flops_per_doc / (us * 1e-6) / 1e12,
        "frac_of_f32_mfma_peak": docs * flops_per_doc / (us * 1e-6) / 1e12 / F32_MFMA_PEAK_TF,
        "padded_tile_tflops": B * L * flops_per_doc / (us * 1e-6) / 1e12,
        "feature_stream_GBs": docs * F * 4 / (us * 1e-6) / 1e9,
    }
    # forward only (evaluation): fused scores kernel vs the three torch layers
    sc = torch.empty(B, L, device=dev)

    def launch_scores():
        _C.check(lib.ltr_mlp_scores_f32(X.data_ptr(), *[p.data_ptr() for p in params], n.data_ptr(),
                                        B, L, F, H1, H2, sc.data_ptr(), torch.cuda.current_stream().cuda_stream))
    for _ in range(5):
        launch_scores()
    us_s, _ = _bu.time_launches(launch_scores, per_graph=10, replays=10)
    out["scores_only_us"] = us_s

    def torch_scores():
        with torch.no_grad():
            h = torch.relu(torch.nn.functional.linear(X, m.l1.weight, m.l1.bias))
            h = torch.relu(torch.nn.functional.linear(h, m.l2.weight, m.l2.bias))
            return torch.nn.functional.linear(h, m.l3.weight, m.l3.bias)
    for _ in range(5):
        torch_scores()
    us_t, _ = _bu.time_launches(torch_scores, per_graph=5, replays=10)
    out["scores_only_torch_layers_us"] = us_t
    if not args.no_unfused:
        loss_fn = {"hinge": L_.PairwiseHingeLoss, "dcg_hinge": L_.PairwiseDCGHingeLoss,
                   "logistic": L_.PairwiseLogisticLoss, "arp1": L_.LambdaARPLoss1,
                   "arp2": L_.LambdaARPLoss2, "ndcg1": L_.LambdaNDCGLoss1,
                   "ndcg2": L_.LambdaNDCGLoss2}[args.kind]()
        ps = list(m.parameters())

        def unfused():
            for p in ps:
                p.grad = None
            loss_fn(m.score(X), relevance, n).mean().backward()

        def module():
            for p in ps:
                p.grad = None
            m(X, relevance, n).backward()
        for name, fn in (("unfused_torch_layers_plus_loss", unfused), ("fused_module_autograd", module)):
            for _ in range(5):
                fn()
            res = {"eager_us": _bu.time_wall(fn, 50, lambda: None) / 50 * 1e6}
            rp = bench.try_graph(fn)
            if rp is not None:
                res["hipgraph_us"] = _bu.time_wall(rp, 100, lambda: None) / 100 * 1e6
            out[name] = res
    print(json.dumps(out))


if __name__ == "__main__":
    main()
