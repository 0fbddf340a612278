"""the SVMrank parser (include/ltr_io.h, csrc/svmrank_parser.cpp) and
the dataset constructor built on it.

Three checkers, strongest first:
  * committed outputs of the REAL reference (tests/golden/svmrank_vectors.npz, made by
    tests/golden/generate_svmrank_golden.py): its C parser on the reference's own test data
    file and on edge-case / malformed texts, and its SVMRankDataset(normalize, filter_queries);
  * the reference's C parser itself, compiled from where it lies into oracle/_ref/ by
    oracle/build_ref.py, on seeded random and mutated files (skipped where neither the
    reference nor a prebuilt oracle/_ref is available);
  * properties: thread-count independence, float32 = one rounding of float64.
Everything is compared bit-exactly.
"""
import ctypes
import os
import random
import re

import numpy as np
import pytest

from pytorchltr_amd import _io
from pytorchltr_amd.datasets import svmrank as S

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden", "svmrank_vectors.npz")
DATA = os.path.join(HERE, "golden", "svmrank_dataset.txt")


@pytest.fixture(scope="module")
def golden():
    return np.load(GOLDEN)


def _write(tmp_path, name, text):
    path = os.path.join(str(tmp_path), name)
    with open(path, "wb") as f:
        f.write(text if isinstance(text, bytes) else text.encode())
    return path


def _same(got, want):
    xs, ys, qids = got
    X, Y, Q = want
    assert xs.shape == X.shape
    assert xs.dtype == np.float64 and ys.dtype == np.int32 and qids.dtype == np.int64
    assert np.array_equal(xs.view(np.uint64), np.ascontiguousarray(X).view(np.uint64))   # bit-exact
    assert np.array_equal(ys, Y)
    assert np.array_equal(qids, Q)


# ---- the C ABI ------------------------------------------------------------------------------

def test_abi_exports_match_header():
    header = open(os.path.join(HERE, "..", "include", "ltr_io.h")).read()
    declared = set(re.findall(r"\b(ltr_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_io.SIGNATURES)
    lib = ctypes.CDLL(_io.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name


def test_status_codes_are_the_reference_ones():
    # PARSE_OK / FILE / FORMAT / MEMORY = 0..3 (svmrank_parser.h:21-24)
    assert (_io.OK, _io.FILE_ERROR, _io.FORMAT_ERROR, _io.MEMORY_ERROR) == (0, 1, 2, 3)
    assert _io.lib().ltr_io_error_string(2) == b"not in SVMrank format"


def test_null_arguments_and_missing_file(tmp_path):
    lib = _io.lib()
    rows, cols, h = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_void_p()
    assert lib.ltr_svmrank_open(None, 1, ctypes.byref(h), ctypes.byref(rows), ctypes.byref(cols)) == _io.ARG_ERROR
    rc = lib.ltr_svmrank_open(os.fsencode(os.path.join(str(tmp_path), "nope.txt")), 1, ctypes.byref(h),
                              ctypes.byref(rows), ctypes.byref(cols))
    assert rc == _io.FILE_ERROR and not h.value
    assert lib.ltr_svmrank_read(None, None, None, None, None) == _io.ARG_ERROR
    lib.ltr_svmrank_close(None)
    with pytest.raises(OSError):
        S.parse_svmrank_file(os.path.join(str(tmp_path), "nope.txt"))
    with pytest.raises(OSError):
        S.parse_svmrank_file(str(tmp_path))                 # a directory


# ---- golden: reference parser outputs ---------------------------------------------------------

def test_reference_dataset_file(golden):
    for threads in (1, 2, 7):
        _same(S.parse_svmrank_file(DATA, n_threads=threads),
              (golden["dataset/xs"], golden["dataset/ys"], golden["dataset/qids"]))


def test_reference_dataset_known_answers():
    # the literal expectations of the reference's own test (tests/datasets/svmrank/test_svmrank.py:29-71)
    xs, ys, qids = S.parse_svmrank_file(DATA)
    off = S.query_offsets(qids)
    assert len(off) - 1 == 4
    assert [int(c) for c in np.diff(off)] == [6, 9, 14, 10]
    assert [int(q) for q in qids[off[:-1]]] == [1, 16, 60, 63]
    assert xs.shape[1] == 45
    assert xs[off[0] + 1, 2] == 1.0 and ys[off[0] + 1] == 2
    assert float(np.float32(xs[off[1] + 5, 3])) == pytest.approx(0.422507) and ys[off[1] + 5] == 1
    assert float(np.float32(xs[off[2] + 12, 2])) == pytest.approx(0.461538) and ys[off[2] + 12] == 0
    assert float(np.float32(xs[off[3] + 8, 2])) == pytest.approx(0.25) and ys[off[3] + 8] == 0


def test_golden_texts(golden, tmp_path):
    names = [str(n) for n in golden["texts/names"]]
    assert len(names) >= 39
    for name in names:
        text = golden["texts/%s/text" % name].tobytes()
        status = int(golden["texts/%s/status" % name])
        path = _write(tmp_path, name + ".txt", text)
        for threads in (1, 3):
            if status == 0:
                _same(S.parse_svmrank_file(path, n_threads=threads),
                      (golden["texts/%s/xs" % name], golden["texts/%s/ys" % name],
                       golden["texts/%s/qids" % name]))
            else:
                assert status == _io.FORMAT_ERROR
                with pytest.raises(ValueError, match="not in SVMrank format"):
                    S.parse_svmrank_file(path, n_threads=threads)


@pytest.mark.parametrize("tag,kw", [("plain", {}), ("normalize", {"normalize": True}),
                                     ("filter", {"filter_queries": True}),
                                     ("normalize_filter", {"normalize": True, "filter_queries": True})])
def test_dataset_constructor_host_side(golden, tag, kw):
    """parse -> offsets -> normalise -> filter, against the reference's SVMRankDataset items."""
    xs, ys, qids = S.parse_svmrank_file(DATA)
    off = S.query_offsets(qids)
    uq = qids[off[:-1]]
    if kw.get("normalize"):
        S.normalize_queries(xs, off)
    assert np.array_equal(xs.view(np.uint64), golden["ds/%s/xs64" % tag].view(np.uint64))
    keep = np.ones(len(uq), bool)
    if kw.get("filter_queries"):
        keep = np.add.reduceat(ys.astype(np.int64), off[:-1]) > 0
    rows = np.repeat(keep, np.diff(off))
    assert np.array_equal(xs[rows].astype(np.float32), golden["ds/%s/features" % tag])
    assert np.array_equal(ys[rows].astype(np.int64), golden["ds/%s/relevance" % tag])
    assert np.array_equal(np.diff(off)[keep], golden["ds/%s/n" % tag])
    assert np.array_equal(uq[keep], golden["ds/%s/qid" % tag])


def test_normalize_known_answers():
    # tests/datasets/svmrank/test_svmrank.py:97-124
    xs, ys, qids = S.parse_svmrank_file(DATA)
    off = S.query_offsets(qids)
    S.normalize_queries(xs, off)
    x = xs[off[0]:off[1]].astype(np.float32)
    assert [float(v) for v in x[:, 1]] == pytest.approx([1.0, 0.5, 0.25, 0.0, 0.125, 0.5])
    assert [float(v) for v in x[:, 0]] == pytest.approx(
        [0.24242424242424246, 0.12121212121212122, 0.060606060606060615, 0.0, 1.0, 0.12121212121212122])


# ---- documented divergences ---------------------------------------------

def test_record_without_blank_after_qid_keeps_its_qid(tmp_path):
    # the reference leaves qids[row] uninitialised here (STORE_QID only fires on ' ')
    path = _write(tmp_path, "a.txt", "1 qid:5 1:1\n2 qid:6\n3 qid:7#c\n4 qid:8\r\n0 qid:9")
    xs, ys, qids = S.parse_svmrank_file(path)
    assert ys.tolist() == [1, 2, 3, 4, 0] and qids.tolist() == [5, 6, 7, 8, 9]
    assert xs.shape == (5, 1) and xs[:, 0].tolist() == [1.0, 0, 0, 0, 0]


@pytest.mark.parametrize("text", ["7", "1 qid:1 1:1\n3", "1 qid", "1 qid:1 2:", "1 qid:1 2:1.", "1 qid:1 2:1.5e", "1 qid:1 2"])
def test_truncated_record_is_rejected(tmp_path, text):
    # the reference returns OK with arrays shorter than `rows` for these
    with pytest.raises(ValueError):
        S.parse_svmrank_file(_write(tmp_path, "t.txt", text))


# ---- against the compiled reference (oracle/_ref) ---------------------------------------------

def _ref():
    from oracle import build_ref
    if build_ref.build() is None:
        pytest.skip("reference parser sources not present and oracle/_ref not prebuilt")
    return build_ref


def _value(r):
    s = "-" * r.choice([0, 0, 0, 1, 1, 2]) + str(r.randint(0, 10 ** r.choice([1, 3, 6, 9])))
    if r.random() < 0.6:
        s += "." + "".join(r.choice("0123456789") for _ in range(r.randint(1, 8)))
        if r.random() < 0.3:
            s += r.choice("eE") + r.choice(["", "+", "-"]) + str(r.randint(0, 12))
    return s


def _file(r, lines):
    out, qid, base, width = [], r.randint(0, 5), r.choice([0, 1, 1, 3]), r.randint(1, 40)
    for _ in range(lines):
        if r.random() < 0.2:
            qid += r.randint(1, 3)
        if r.random() < 0.05:
            out.append(" " * r.randint(0, 2) + "# full comment line")
        cols = sorted(r.sample(range(base, base + width), r.randint(0, min(width, 12))))
        if r.random() < 0.1:
            r.shuffle(cols)
        line = " " * r.choice([0, 0, 2]) + "%d %sqid:%d " % (r.randint(0, 4), " " * r.choice([0, 1]), qid)
        line += " ".join(" " * r.choice([0, 0, 1]) + "%d:%s" % (c, _value(r)) for c in cols)
        line += r.choice(["", "", "", " ", " # note 1:2", "#x", "\r"])
        out.append(line)
    return "\n".join(out) + ("\n" if r.random() < 0.8 else "")


def _mutate(r, s):
    b = list(s)
    for _ in range(r.randint(1, 3)):
        if not b:
            break
        i, k = r.randrange(len(b)), r.random()
        if k < 0.4:
            b[i] = r.choice(" 0123456789:.-+eEqid#\n\r\tx")
        elif k < 0.7:
            del b[i]
        else:
            b.insert(i, r.choice(" 0123456789:.-+eEqid#\n\r"))
    return "".join(b)


# inputs on which the reference is inconsistent with itself (see the divergence tests above)
_QUIRK = re.compile(r"qid:\d+([\n#\r]|$)|^ *\d+$|qid:?$|:-*$|\.$|[eE]$|^ *\d+ +q?i?d?$| \d+$", re.M)


def test_random_and_mutated_files_match_compiled_reference(tmp_path):
    ref = _ref()
    r = random.Random(20240928)
    compared = rejected = 0
    for it in range(1500):
        text = _file(r, r.randint(0, 25))
        if r.random() < 0.6:
            text = _mutate(r, text)
        path = _write(tmp_path, "f.txt", text)
        rc, X, Y, Q = ref.parse_svmrank_file(path)
        if rc == 0:
            if _QUIRK.search(text):
                continue
            _same(S.parse_svmrank_file(path, n_threads=r.choice([1, 2, 3])), (X, Y, Q))
            compared += 1
        else:
            assert rc == _io.FORMAT_ERROR
            with pytest.raises(ValueError):
                S.parse_svmrank_file(path, n_threads=r.choice([1, 2, 3]))
            rejected += 1
    assert compared > 400 and rejected > 300


def test_large_file_threads_and_float32(tmp_path):
    """~6 MB file (so several parser threads really run): thread-count independence, the
    compiled reference when available, and float32 = one rounding of the float64 result."""
    rng = np.random.default_rng(5)
    rows, F = 4000, 136
    vals = rng.random((rows, F))
    q = np.repeat(np.arange(rows // 40), 40)
    y = rng.integers(0, 5, rows)
    path = os.path.join(str(tmp_path), "big.txt")
    with open(path, "w") as f:
        for i in range(rows):
            f.write("%d qid:%d " % (y[i], q[i]) + " ".join("%d:%.6f" % (j + 1, vals[i, j]) for j in range(F)) + "\n")
    assert os.path.getsize(path) > 4 << 20
    one = S.parse_svmrank_file(path, n_threads=1)
    for threads in (2, 5, 0):
        _same(S.parse_svmrank_file(path, n_threads=threads), one)
    assert one[0].shape == (rows, F) and np.array_equal(one[1], y) and np.array_equal(one[2], q)
    x32 = S.parse_svmrank_file(path, n_threads=3, dtype=np.float32)[0]
    assert x32.dtype == np.float32 and np.array_equal(x32, one[0].astype(np.float32))
    from oracle import build_ref
    if build_ref.build() is not None:
        rc, X, Y, Q = build_ref.parse_svmrank_file(path)
        assert rc == 0
        _same(one, (X, Y, Q))


# ---- device: file -> RaggedQueries -> padded batch ---------------------------------------------

@pytest.mark.gpu
@pytest.mark.parametrize("tag,kw", [("plain", {}), ("normalize_filter", {"normalize": True, "filter_queries": True})])
def test_load_svmrank_to_device(golden, tag, kw):
    from pytorchltr_amd.datasets import UniformSampler
    ds = S.load_svmrank(DATA, device="cuda", **kw)
    n = golden["ds/%s/n" % tag]
    assert len(ds) == len(n)
    assert np.array_equal(ds.features.cpu().numpy(), golden["ds/%s/features" % tag])
    assert np.array_equal(ds.relevance.cpu().numpy(), golden["ds/%s/relevance" % tag])
    batch = ds.collate(list(range(len(ds))))
    assert batch.features.shape == (len(n), int(n.max()), 45)        # test_collate_dense_all
    assert np.array_equal(batch.n.cpu().numpy(), n)
    assert np.array_equal(batch.qid.cpu().numpy(), golden["ds/%s/qid" % tag])
    off = np.concatenate([[0], np.cumsum(n)])
    for b in range(len(n)):
        want = golden["ds/%s/features" % tag][off[b]:off[b + 1]]
        assert np.array_equal(batch.features[b, :n[b]].cpu().numpy(), want)
        assert not batch.features[b, n[b]:].any()
    small = ds.collate([0, 1, 2], UniformSampler(max_list_size=3))
    assert small.features.shape == (3, 3, 45)                        # test_collate_dense_3


def test_example3_files_match_pinned_digests_and_known_features():
    """tests/golden/example3_*.dat are the Example3 bytes the reference pins by sha256
    (example3.py:29-30); query 1 normalises to the matrix shown in docs/source/datasets.rst."""
    import hashlib
    want = {"train": "503aa66c6a1b1bb8a86b14e52163dcdb5bcffc017981afdff4cf026eacc592cf",
            "test": "81aaac13dfc5180edce38a588cec80ee00b5d85662e00d1b7ac1d3f98242698e"}
    for split, digest in want.items():
        path = os.path.join(HERE, "golden", "example3_%s.dat" % split)
        assert hashlib.sha256(open(path, "rb").read()).hexdigest() == digest
    xs, ys, qids = S.parse_svmrank_file(os.path.join(HERE, "golden", "example3_train.dat"))
    assert xs.shape == (12, 5) and qids.tolist() == [1] * 4 + [2] * 4 + [3] * 4
    assert ys.tolist() == [3, 2, 1, 1, 1, 2, 1, 1, 2, 3, 4, 1]
    off = S.query_offsets(qids)
    S.normalize_queries(xs, off)
    assert np.allclose(xs[:4], [[1, 1, 0, 1 / 3, 0], [0, 0, 1, 0, 1], [0, 1, 0, 1, 0], [0, 0, 1, 2 / 3, 0]])


@pytest.mark.gpu
def test_basic_usage_example_end_to_end():
    """examples/01_basic_usage.py = the reference's examples/01-basic-usage.py: file -> parser ->
    device split -> device collate -> HIP loss -> SGD -> HIP ndcg.  Reference trace on this data:
    0.8617 at start, 1.0000 after training."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "basic_usage", os.path.join(HERE, "..", "examples", "01_basic_usage.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    trace = mod.run(log=lambda msg: None)
    assert len(trace) == 4
    assert trace[0] == pytest.approx(0.8617, abs=1e-4)
    assert trace[-1] == pytest.approx(1.0, abs=1e-6)
    assert all(b >= a - 1e-6 for a, b in zip(trace, trace[1:]))
