#!/usr/bin/env python
"""Host-side timing of the SVMrank parser (libltr_io) next to the reference's own C parser
(oracle/_ref/libsvmrank_ref.so, when present) on a synthetic MSLR-shaped file.

    python tests/bench_parser.py [--rows 120000 --features 136]"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pytorchltr_amd.datasets.svmrank import parse_svmrank_file  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=120000)
    ap.add_argument("--features", type=int, default=136)
    args = ap.parse_args()
    rng = np.random.default_rng(0)
    vals = rng.random((args.rows, args.features))
    q = np.repeat(np.arange(args.rows // 120 + 1), 120)[:args.rows]
    y = rng.integers(0, 5, args.rows)
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "big.txt")
        with open(path, "w") as f:
            for i in range(args.rows):
                f.write("%d qid:%d " % (y[i], q[i]) +
                        " ".join("%d:%.6f" % (j + 1, vals[i, j]) for j in range(args.features)) + "\n")
        size = os.path.getsize(path)
        out = {"file_MB": size / 1e6, "rows": args.rows, "features": args.features,
               "host_cpus": os.cpu_count(), "threads": {}}
        ref = None
        try:
            from oracle import build_ref
            if build_ref.build() is not None:
                t0 = time.perf_counter()
                rc, X, Y, Q = build_ref.parse_svmrank_file(path)
                out["reference_parser_s"] = time.perf_counter() - t0
                ref = (X, Y, Q)
        except Exception as exc:                                    # noqa: BLE001
            out["reference_parser_error"] = str(exc)
        for nt in (1, 2, 4, 8, 16, 32, 64):
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                xs, ys, qs = parse_svmrank_file(path, n_threads=nt)
                best = min(best, time.perf_counter() - t0)
            out["threads"][str(nt)] = {"seconds": best, "MB_per_s": size / 1e6 / best}
        if ref is not None:
            out["identical_to_reference"] = bool(np.array_equal(xs, ref[0]) and np.array_equal(ys, ref[1])
                                                 and np.array_equal(qs, ref[2]))
        print(json.dumps(out))


if __name__ == "__main__":
    main()
