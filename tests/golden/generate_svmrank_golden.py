#!/usr/bin/env python
"""Golden vectors for the SVMrank ingestion row from the REAL reference.

Two sources, both run in this container only:
  * the reference's C parser, compiled from where it lies by oracle/build_ref.py
    (oracle/_ref/libsvmrank_ref.so), run on tests/golden/svmrank_dataset.txt (the data file
    the reference's own dataset tests use, tests/datasets/resources/dataset.txt) and on the
    synthetic texts below (valid edge cases and malformed inputs with their status codes);
  * the reference's Python SVMRankDataset (built scratch copy, see generate_collate_golden.py)
    for the normalize / filter_queries constructor paths on the same data file.

    PYTORCHLTR_REFERENCE=/tmp/ref_build python tests/golden/generate_svmrank_golden.py

Writes tests/golden/svmrank_vectors.npz.  Data only.
"""
import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
from oracle import build_ref  # noqa: E402

REFERENCE = os.environ.get("PYTORCHLTR_REFERENCE", "/tmp/ref_build")
DATA_SRC = os.path.join(build_ref.REF_ROOT, "tests", "datasets", "resources", "dataset.txt")
DATA = os.path.join(HERE, "svmrank_dataset.txt")

# name -> text.  Everything the reference's automaton distinguishes (svmrank_parser.h:80-132).
TEXTS = {
    "plain": "3 qid:1 1:0.5 2:1 3:0.25\n0 qid:1 1:1.5 2:0 3:7\n1 qid:2 1:2 2:3 3:4\n",
    "no_trailing_newline": "3 qid:1 1:0.5 2:1\n0 qid:1 1:1.5 2:8.25",
    "eof_drops_sign": "1 qid:4 1:2 2:-3.5",
    "eof_in_exponent": "1 qid:4 1:2 2:3.5e-2",
    "one_based_shift": "1 qid:7 5:1 9:2\n2 qid:7 6:3\n",
    "zero_based": "1 qid:7 0:1 3:2\n2 qid:7 2:3\n",
    "sparse_rows": "0 qid:1 10:1\n1 qid:1 \n2 qid:2 3:4 # only three\n",
    "negatives": "1 qid:1 1:-2 2:--3 3:-0.125 4:-1.5e2 5:-0\n",
    "exponents": "1 qid:1 1:1.5e3 2:1.5E3 3:2.5e+2 4:2.5e-2 5:1.0e0 6:7.25e- 7:7.25e+ \n",
    "exponent_chain": "1 qid:1 1:1.5e10 2:2.5e-10 3:1.0e22 4:1.0e-22 5:123456.789012e-3\n",
    "long_fraction": "1 qid:1 1:0.123456789012345678 2:3.141592653589793 3:0.000001 4:1000000.000001\n",
    "leading_blanks": "   2   qid:5   1:1    2:2   \n 0 qid:5 1:3 2:4\n",
    "comments": "# header line\n1 qid:1 1:1 # trailing 2:9\n   # indented comment\n2 qid:1 1:2#tight\n",
    "crlf": "1 qid:1 1:1 2:2\r\n0 qid:1 1:3 2:4\r\n",
    "cr_mid_line": "1 qid:1 1:1\r 2:2\n0 qid:1 1:3\n",
    "repeated_column": "1 qid:1 1:1 1:5 2:2\n",
    "unordered_columns": "1 qid:1 3:3 1:1 2:2\n",
    "big_numbers": "4 qid:123456789012 100000:1 1:9007199254740993\n",
    "multi_digit_label": "12 qid:3 1:1\n007 qid:3 1:2\n",
    "no_features_at_all": "1 qid:1 \n2 qid:2 \n",
    "trailing_blank_lines_of_spaces": "1 qid:1 1:1\n   ",
    "empty_file": "",
    # malformed (status 2)
    "bad_empty_line": "1 qid:1 1:1\n\n2 qid:1 1:2\n",
    "bad_negative_label": "-1 qid:1 1:1\n",
    "bad_float_label": "1.5 qid:1 1:1\n",
    "bad_missing_qid": "1 1:1 2:2\n",
    "bad_qid_word": "1 qud:1 1:1\n",
    "bad_qid_empty": "1 qid: 1:1\n",
    "bad_plus_sign": "1 qid:1 1:+5\n",
    "bad_exponent_without_fraction": "1 qid:1 1:5e3\n",
    "bad_dot_without_digits": "1 qid:1 1:5. 2:1\n",
    "bad_leading_dot": "1 qid:1 1:.5\n",
    "bad_missing_value": "1 qid:1 1: 2:1\n",
    "bad_missing_colon": "1 qid:1 1 2:1\n",
    "bad_tab": "1 qid:1\t1:1\n",
    "bad_nan": "1 qid:1 1:nan\n",
    "bad_double_exponent_sign": "1 qid:1 1:1.5e--2\n",
    "bad_bare_exponent": "1 qid:1 1:7.25e 2:1\n",
    "bad_second_line": "1 qid:1 1:1\nx qid:1 1:1\n",
}


def main():
    if not os.path.exists(DATA):
        shutil.copyfile(DATA_SRC, DATA)          # a data file of the reference's tests
    arrays = {}
    rc, xs, ys, qids = build_ref.parse_svmrank_file(DATA)
    assert rc == 0
    arrays["dataset/xs"], arrays["dataset/ys"], arrays["dataset/qids"] = xs, ys, qids
    names = sorted(TEXTS)
    arrays["texts/names"] = np.array(names)
    with tempfile.TemporaryDirectory() as tmp:
        for name in names:
            path = os.path.join(tmp, name + ".txt")
            with open(path, "w", newline="") as f:
                f.write(TEXTS[name])
            rc, xs, ys, qids = build_ref.parse_svmrank_file(path)
            arrays["texts/%s/text" % name] = np.frombuffer(TEXTS[name].encode(), dtype=np.uint8)
            arrays["texts/%s/status" % name] = np.int64(rc)
            if rc == 0:
                arrays["texts/%s/xs" % name] = xs
                arrays["texts/%s/ys" % name] = ys
                arrays["texts/%s/qids" % name] = qids

    sys.path.insert(0, REFERENCE)
    from pytorchltr.datasets.svmrank.svmrank import SVMRankDataset
    for tag, kw in (("plain", {}), ("normalize", {"normalize": True}),
                    ("filter", {"filter_queries": True}),
                    ("normalize_filter", {"normalize": True, "filter_queries": True})):
        ds = SVMRankDataset(DATA, **kw)
        feats, rels, ns, qs = [], [], [], []
        for i in range(len(ds)):
            item = ds[i]
            feats.append(item.features.numpy())
            rels.append(item.relevance.numpy())
            ns.append(int(item.n))
            qs.append(int(item.qid))
        arrays["ds/%s/features" % tag] = np.concatenate(feats, axis=0)
        arrays["ds/%s/relevance" % tag] = np.concatenate(rels, axis=0)
        arrays["ds/%s/n" % tag] = np.array(ns, dtype=np.int64)
        arrays["ds/%s/qid" % tag] = np.array(qs, dtype=np.int64)
        arrays["ds/%s/xs64" % tag] = np.asarray(ds._xs)
    np.savez_compressed(os.path.join(HERE, "svmrank_vectors.npz"), **arrays)
    print("wrote", len(arrays), "arrays;",
          sum(int(arrays["texts/%s/status" % n]) == 0 for n in names), "valid texts,",
          sum(int(arrays["texts/%s/status" % n]) != 0 for n in names), "malformed")


if __name__ == "__main__":
    main()
