"""DCG / NDCG on MI355X (reference: evaluation/dcg.py:8-99).

One kernel per call ranks each query by score (and, for NDCG, by label) with an in-LDS
counting rank, applies gains and log2 discounts and either reduces to metric@k or prefix-sums
the whole curve.  As in the reference, labels of padded documents are not masked, and equal
scores are ordered by a random permutation drawn per call (``pytorchltr_amd.utils.tie_breaking``
selects the deterministic index order instead).
"""
from typing import Optional

import torch as _torch

from pytorchltr_amd import _C
from pytorchltr_amd import _ties
from pytorchltr_amd._prepare import prepare as _prepare


def _cutoff(k, L):
    """Column the reference's `dcg[:, :k][:, -1]` (dcg.py:97-98) selects, as a 1-based k."""
    if k is None:
        return 0
    k = int(k)
    kk = k if k >= 0 else L + k         # python slice semantics: `:k` with negative k
    if kk <= 0:
        raise IndexError("index -1 is out of bounds for dimension 1 with size 0")
    return min(kk, L)


def _run(scores, relevance, n, k, exp, normalize):
    s, r, nn = _prepare(scores, relevance, n)
    B, L = s.shape
    kk = _cutoff(k, L)
    out = _torch.empty((B,) if kk > 0 else (B, L), dtype=_torch.float32, device=s.device)
    if B > 0:
        # the reference ranks through rank_by_score with its global-RNG tie-break (dcg.py:85)
        # (round 3: a seed drawn on the host, hashed into tie words inside the kernel -- no randperm launches)
        sd = _ties.draw_seed(L, s.device)
        with _C.device_ctx(s):
            if sd is None:
                _C.check(_C.lib().ltr_dcg_f32(_C.ptr(s), _C.ptr(r), _C.label_dtype(r), _C.ptr(nn), B, L, kk,
                                              int(bool(exp)), int(normalize), _C.ptr(out), _C.stream_of(s)))
            else:
                _C.check(_C.lib().ltr_dcg_seed_f32(_C.ptr(s), _C.ptr(r), _C.label_dtype(r), _C.ptr(nn), sd[0],
                                                   _C.ptr(sd[1]), B, L, kk, int(bool(exp)), int(normalize),
                                                   _C.ptr(out), _C.stream_of(s)))
    return out


def ndcg(scores: _torch.FloatTensor, relevance: _torch.LongTensor,
         n: _torch.LongTensor, k: Optional[int] = None,
         exp: Optional[bool] = True) -> _torch.FloatTensor:
    r"""Normalized DCG: dcg(scores, y) / dcg(y, y), with 0/0 := 0 (reference :8-38).

    Returns (batch, list_size) NDCG at every rank, or (batch,) NDCG@k when k is given.
    """
    return _run(scores, relevance, n, k, exp, True)


def dcg(scores: _torch.FloatTensor, relevance: _torch.LongTensor,
        n: _torch.LongTensor, k: Optional[int] = None,
        exp: Optional[bool] = True) -> _torch.FloatTensor:
    r"""DCG: :math:`\sum_i \mathrm{gain}(y_{\pi_i}) / \log_2(1 + i)` with gain
    :math:`2^y - 1` (exp=True) or :math:`y` (reference :41-99).

    Returns (batch, list_size) DCG at every rank, or (batch,) DCG@k when k is given
    (k larger than list_size means DCG@list_size, as in the reference).
    """
    return _run(scores, relevance, n, k, exp, False)
