#!/usr/bin/env python
"""Per-CU byte histogram of the register-tile launch from a -DLTR_TRACE build (the stamps carry HW_ID / XCC_ID):
python scripts/cu_bytes.py build/variants/libltr_trace.so [--workload c2]
For each of several cold launches: rows (= sum of n[b]) of the queries every CU hosted -- min / mean / max over the CUs, and the
same split by the half of the dispatch round the CU's first workgroup fell in (straight dealing gives the
first half of every round the longer half of each quartile)."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import WORKLOADS, synth  # noqa: E402
from pytorchltr_amd import _C  # noqa: E402
from scripts.trace_regtile import load  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("lib")
    ap.add_argument("--workload", default="c2")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    B, L, F, kind = WORKLOADS[args.workload]
    lib = load(args.lib)
    k = getattr(_C, kind.upper())
    W = (torch.rand(F, device=dev) * 2 - 1) / F ** 0.5
    bias = torch.zeros(1, device=dev)
    loss = torch.empty(B, device=dev)
    part = torch.empty(lib.ltr_linear_workspace_bytes(B, L, F) // 4 + 64, device=dev)
    tr = torch.zeros(B * 16, dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for rep in range(6):
        _, rel, n, X = synth(B, L, F, 100 + rep, dev)
        rc = lib.ltr_linear_partials_f32(k, 1.0, X.data_ptr(), W.data_ptr(), bias.data_ptr(), rel.data_ptr(), 0,
                                         n.data_ptr(), B, L, F, loss.data_ptr(), tr.data_ptr(), part.data_ptr(), st)
        assert rc == 0, rc
        torch.cuda.synchronize()
        t = tr.cpu().view(B, 16)
        meta = t[:, 15]
        blk = meta & 0xffffffff
        hw = (meta >> 32) & 0xffff
        xcc = (meta >> 48) & 0xf
        cu = ((hw >> 8) & 0xf) | (((hw >> 13) & 0x7) << 4) | (xcc << 8)
        rows = n.cpu().double().clamp(0, L)              # (trace row b belongs to QUERY b)
        uniq, inv = torch.unique(cu, return_inverse=True)
        per_cu = torch.zeros(len(uniq), dtype=torch.float64).index_add_(0, inv, rows)
        cnt = torch.zeros(len(uniq), dtype=torch.float64).index_add_(0, inv, torch.ones(B, dtype=torch.float64))
        first = torch.full((len(uniq),), 1 << 40, dtype=torch.int64).scatter_reduce(0, inv, blk, reduce="amin")
        half = ((first % 256) >= 128)
        span = (t[:, 6].max() - t[:, 14].min()).item() * 10.0
        print("launch %d: %d CUs, workgroups per CU %d..%d; rows per CU min %.0f mean %.1f max %.0f (max/mean %.3f); CUs whose first "
              "block id is in the first / second half of a round: mean rows %.1f / %.1f; launch span %.0f ns"
              % (rep, len(uniq), cnt.min(), cnt.max(), per_cu.min(), per_cu.mean(), per_cu.max(), per_cu.max() / per_cu.mean(),
                 per_cu[~half].mean(), per_cu[half].mean() if half.any() else float("nan"), span))


if __name__ == "__main__":
    main()
