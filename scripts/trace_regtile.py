#!/usr/bin/env python
"""Timeline of the register-tile kernel from a -DLTR_TRACE -DLTR_TRACE_WALL build:
python scripts/trace_regtile.py build/variants/libltr_trace.so [--full] [--kind hinge] [--nbuf 5]

Stamps (100 MHz wall clock, per workgroup, relative to the first workgroup's entry):
entry | 0 after scheduling | 1 loads landed + dots + barrier | 2 scores folded | 3 rankings (NDCG kinds) |
4 pair pass | 5 gradients final | 6 partials written."""
import argparse
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import WORKLOADS, synth  # noqa: E402
from pytorchltr_amd import _C  # noqa: E402


def load(path):
    h = ctypes.CDLL(path)
    for name, (restype, argtypes) in _C.SIGNATURES.items():
        fn = getattr(h, name)
        fn.restype = restype
        fn.argtypes = argtypes
    return h


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("lib")
    ap.add_argument("--workload", default="c2")
    ap.add_argument("--kind", default="")
    ap.add_argument("--full", action="store_true")
    ap.add_argument("--nbuf", type=int, default=5)
    ap.add_argument("--B", type=int, default=0)
    ap.add_argument("--n", type=int, default=0, help="every list this long")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    B, L, F, kind = WORKLOADS[args.workload]
    if args.B:
        B = args.B
    kind = args.kind or kind
    lib = load(args.lib)
    k = getattr(_C, kind.upper())
    bufs = []
    for i in range(args.nbuf):
        _, rel, n, X = synth(B, L, F, 100 + i, dev)
        if args.full:
            n = torch.full_like(n, L)
        if args.n:
            n = torch.full_like(n, args.n)
        bufs.append((X, rel, n))
    W = (torch.rand(F, device=dev) * 2 - 1) / F ** 0.5
    bias = torch.zeros(1, device=dev)
    loss = torch.empty(B, device=dev)
    part = torch.empty(lib.ltr_linear_workspace_bytes(B, L, F) // 4 + 64, device=dev)
    tr = torch.zeros(B * 16, dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    rows = []
    for rep in range(3 * args.nbuf):
        X, rel, n = bufs[rep % args.nbuf]
        rc = lib.ltr_linear_partials_f32(k, 1.0, X.data_ptr(), W.data_ptr(), bias.data_ptr(), rel.data_ptr(), 0,
                                         n.data_ptr(), B, L, F, loss.data_ptr(), tr.data_ptr(), part.data_ptr(), st)
        assert rc == 0, rc
        torch.cuda.synchronize()
        if rep >= args.nbuf:
            rows.append((tr.cpu().view(B, 16).clone(), n.cpu()))
    names = ["entry", "sched", "loaded", "scores", "ranks", "pair", "gfin", "end"]
    acc = None
    for t, n in rows:
        t = t.clone()
        meta = t[:, 15]
        blk = (meta & 0xffffffff)
        order = torch.argsort(blk)
        ts = torch.stack([t[:, 14]] + [t[:, i] for i in range(7)], dim=1).double()
        t0 = ts[:, 0].min()
        ts = (ts - t0) * 10.0                       # ns
        ts = ts[order]
        nn = n[order].double()
        q = max(1, B // 4)
        out = []
        for qi in range(4):
            sl = slice(qi * q, (qi + 1) * q)
            out.append(torch.cat([ts[sl].mean(0), ts[sl, 7:8].max(0).values, nn[sl].mean(0, keepdim=True)]))
        out = torch.stack(out)
        acc = out if acc is None else acc + out
        span = ts[:, 7].max()
    acc /= len(rows)
    print("%s %s %s: mean ns since first entry, by quartile of block id (dispatch order); last launch span %.0f ns"
          % (args.workload, kind, "full lists" if args.full else "ragged", span))
    print("quartile  " + "  ".join("%8s" % nm for nm in names) + "   max_end   mean_n")
    for qi in range(4):
        print("   %d     " % qi + "  ".join("%8.0f" % v for v in acc[qi, :8]) + "  %8.0f  %6.1f" % (acc[qi, 8], acc[qi, 9]))
    # phase durations averaged over all workgroups of the last launch
    d = ts[:, 1:] - ts[:, :-1]
    print("phase mean ns: " + "  ".join("%s %.0f" % (nm, v) for nm, v in zip(names[1:], d.mean(0))))
    # inside the pair pass (wave 0): setup | steps + flush | barrier | rest, from the extra stamps 8..10
    tt = t.double()
    inner = torch.stack([tt[:, 8] - tt[:, 3], tt[:, 9] - tt[:, 8], tt[:, 10] - tt[:, 9], tt[:, 4] - tt[:, 10]], dim=1) * 10.0
    print("pair pass inner mean ns (wave 0): setup %.0f  steps+flush %.0f  sum+barrier %.0f  tail %.0f" % tuple(inner.mean(0).tolist()))
    # per-CU occupancy picture of the last launch: workgroups per (xcc, se, cu)
    hw = (meta >> 32) & 0xffff
    xcc = (meta >> 48) & 0xf
    cu = ((hw >> 8) & 0xf) | (((hw >> 13) & 0x7) << 4) | (xcc << 8)
    uniq, cnt = torch.unique(cu, return_counts=True)
    print("distinct (xcc,se,cu) ids %d; workgroups per id min %d max %d" % (len(uniq), cnt.min(), cnt.max()))


if __name__ == "__main__":
    main()
