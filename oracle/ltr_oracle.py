"""ctypes/numpy front-end of ``oracle/ltr_oracle.c``.

TEST INFRASTRUCTURE ONLY -- not part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module; ``pytorchltr_amd`` never does (tests/test_boundary.py greps for it).

The C file is the CPU restatement of the reference hot path (double precision,
scalar loops); this module only marshals numpy arrays in and out.  Parity
status: PINNED -- see the header of ``ltr_oracle.c`` and
``tests/test_oracle_golden.py``.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "ltr_oracle.c")
_BUILD_DIR = os.path.join(_HERE, "_build")
_LIB_PATH = os.path.join(_BUILD_DIR, "libltr_oracle.so")

KINDS = {
    "hinge": 0,
    "dcg_hinge": 1,
    "logistic": 2,
    "arp1": 3,
    "arp2": 4,
    "ndcg1": 5,
    "ndcg2": 6,
}

_lib = None


def build(force=False):
    """Compile the C restatement with gcc (seconds).  Returns the .so path."""
    os.makedirs(_BUILD_DIR, exist_ok=True)
    stale = (not os.path.exists(_LIB_PATH)
             or os.path.getmtime(_LIB_PATH) < os.path.getmtime(_SRC))
    if force or stale:
        subprocess.check_call(
            ["gcc", "-O2", "-fPIC", "-shared", "-std=c99", "-o", _LIB_PATH, _SRC, "-lm"])
    return _LIB_PATH


def _load():
    global _lib
    if _lib is None:
        # LTR_ORACLE_LIB: another build of ltr_oracle.c (the sanitizer build of scripts/sanitize_host.sh)
        _lib = ctypes.CDLL(os.environ.get("LTR_ORACLE_LIB") or build())
    return _lib


def _d(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def _n(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.int64))


def _p(a, ct=ctypes.c_double):
    return a.ctypes.data_as(ctypes.POINTER(ct))


def _bl(scores):
    s = _d(scores)
    if s.ndim == 3:
        s = s.reshape(s.shape[0], s.shape[1])
    return s


def pairwise_loss(kind, scores, relevance, n, sigma=1.0, need_grad=True):
    """(loss[B], dscores[B,L] or None) for kind in KINDS."""
    s = _bl(scores)
    y = _bl(relevance)
    nn = _n(n)
    B, L = s.shape
    loss = np.zeros(B, dtype=np.float64)
    ds = np.zeros((B, L), dtype=np.float64) if need_grad else None
    rc = _load().oracle_pairwise_loss(
        ctypes.c_int(KINDS[kind]), ctypes.c_double(sigma), _p(s), _p(y),
        _p(nn, ctypes.c_int64), B, L, _p(loss),
        _p(ds) if need_grad else None)
    if rc != 0:
        raise RuntimeError("oracle_pairwise_loss failed: %d" % rc)
    return loss, ds


class tie_priorities:
    """Context: rank with the reference's random tie-break given ONE drawn permutation --
    `tie[j]` is document j's priority among equal scores (smaller ranks first).  Applies to
    rank_by_score, dcg/ndcg, arp and the rankings inside the Lambda losses of this oracle."""

    def __init__(self, tie):
        self.tie = None if tie is None else np.ascontiguousarray(np.asarray(tie, dtype=np.int32))

    def __enter__(self):
        _load().oracle_set_tie(None if self.tie is None else _p(self.tie, ctypes.c_int32))
        return self

    def __exit__(self, *exc):
        _load().oracle_set_tie(None)
        return False


def listwise_softmax(scores, relevance, n, need_grad=True):
    """(loss[B], dscores[B,L] or None) of the listwise softmax cross-entropy -- parity unpinned,
    the reference has no such loss (see ltr_oracle.c)."""
    s = _bl(scores)
    y = _bl(relevance)
    nn = _n(n)
    B, L = s.shape
    loss = np.zeros(B, dtype=np.float64)
    ds = np.zeros((B, L), dtype=np.float64) if need_grad else None
    rc = _load().oracle_listwise_softmax(_p(s), _p(y), _p(nn, ctypes.c_int64), B, L, _p(loss),
                                         _p(ds) if need_grad else None)
    if rc != 0:
        raise RuntimeError("oracle_listwise_softmax failed: %d" % rc)
    return loss, ds


def rank_by_score(scores, n):
    s = _bl(scores)
    nn = _n(n)
    B, L = s.shape
    out = np.zeros((B, L), dtype=np.int64)
    _load().oracle_rank_by_score(_p(s), _p(nn, ctypes.c_int64), B, L, _p(out, ctypes.c_int64))
    return out


def dcg(scores, relevance, n, k=None, exp=True, normalize=False):
    s = _bl(scores)
    y = _bl(relevance)
    nn = _n(n)
    B, L = s.shape
    kk = 0 if k is None else int(k)
    out = np.zeros((B,) if kk > 0 else (B, L), dtype=np.float64)
    rc = _load().oracle_dcg(_p(s), _p(y), _p(nn, ctypes.c_int64), B, L, kk,
                            int(bool(exp)), int(bool(normalize)), _p(out))
    if rc != 0:
        raise RuntimeError("oracle_dcg failed: %d" % rc)
    return out


def ndcg(scores, relevance, n, k=None, exp=True):
    return dcg(scores, relevance, n, k=k, exp=exp, normalize=True)


def arp(scores, relevance, n):
    s = _bl(scores)
    y = _bl(relevance)
    nn = _n(n)
    B, L = s.shape
    out = np.zeros(B, dtype=np.float64)
    rc = _load().oracle_arp(_p(s), _p(y), _p(nn, ctypes.c_int64), B, L, _p(out))
    if rc != 0:
        raise RuntimeError("oracle_arp failed: %d" % rc)
    return out


def mask_padded_values(xs, n, mask_value=-np.inf):
    s = _bl(xs)
    nn = _n(n)
    B, L = s.shape
    out = np.zeros((B, L), dtype=np.float64)
    _load().oracle_mask_padded_values(_p(s), _p(nn, ctypes.c_int64), B, L,
                                      ctypes.c_double(mask_value), _p(out))
    return out


def batch_pairs(x):
    s = _bl(x)
    B, L = s.shape
    out = np.zeros((B, L, L, 2), dtype=np.float64)
    _load().oracle_batch_pairs(_p(s), B, L, _p(out))
    return out


def mlp_pairwise(kind, X, params, relevance, n, grad_out, sigma=1.0):
    """ReLU MLP (F-H1-H2-1) scorer + loss + backward.  params = (W1, b1, W2, b2, W3, b3) in
    torch.nn.Linear layout.  Returns (loss[B], scores[B,L], grads dict with the same keys)."""
    Xd = _d(X)
    B, L, F = Xd.shape
    W1, b1, W2, b2, W3, b3 = [_d(a) for a in params]
    H1, H2 = W1.shape[0], W2.shape[0]
    assert W1.shape == (H1, F) and W2.shape == (H2, H1) and W3.size == H2 and b3.size == 1
    y = _bl(relevance)
    nn = _n(n)
    go = _d(grad_out).reshape(B)
    loss = np.zeros(B, dtype=np.float64)
    scores = np.zeros((B, L), dtype=np.float64)
    P = H1 * F + H1 + H2 * H1 + 2 * H2 + 1
    grads = np.zeros(P, dtype=np.float64)
    fn = _load().oracle_mlp_pairwise
    fn.restype = ctypes.c_int
    rc = fn(ctypes.c_int(KINDS[kind]), ctypes.c_double(sigma), _p(Xd), _p(W1), _p(b1.reshape(-1)),
            _p(W2), _p(b2.reshape(-1)), _p(W3.reshape(-1)), ctypes.c_double(float(b3.reshape(-1)[0])),
            _p(y), _p(nn, ctypes.c_int64), _p(go), B, L, F, H1, H2, _p(loss), _p(scores), _p(grads))
    if rc != 0:
        raise RuntimeError("oracle_mlp_pairwise failed: %d" % rc)
    o = 0
    out = {}
    for key, shape in (("W1", (H1, F)), ("b1", (H1,)), ("W2", (H2, H1)), ("b2", (H2,)),
                       ("W3", (1, H2)), ("b3", (1,))):
        size = int(np.prod(shape))
        out[key] = grads[o:o + size].reshape(shape).copy()
        o += size
    out["flat"] = grads
    return loss, scores, out


def linear_pairwise(kind, X, W, bias, relevance, n, grad_out, sigma=1.0):
    """Linear(F,1) scorer + loss: returns (loss[B], scores[B,L], dW[F], db)."""
    Xd = _d(X)
    B, L, F = Xd.shape
    Wd = _d(W).reshape(F)
    y = _bl(relevance)
    nn = _n(n)
    go = _d(grad_out).reshape(B)
    loss = np.zeros(B, dtype=np.float64)
    scores = np.zeros((B, L), dtype=np.float64)
    dW = np.zeros(F, dtype=np.float64)
    db = ctypes.c_double(0.0)
    rc = _load().oracle_linear_pairwise(
        ctypes.c_int(KINDS[kind]), ctypes.c_double(sigma), _p(Xd), _p(Wd),
        ctypes.c_double(float(bias)), _p(y), _p(nn, ctypes.c_int64), _p(go),
        B, L, F, _p(loss), _p(scores), _p(dW), ctypes.byref(db))
    if rc != 0:
        raise RuntimeError("oracle_linear_pairwise failed: %d" % rc)
    return loss, scores, dW, db.value
