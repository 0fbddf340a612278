"""L1 helpers of the hot path on MI355X (reference: utils/tensor_operations.py).

The losses and metrics never call these (their kernels fuse masking, ranking and the pair
expansion); they are provided because the reference exports them.
"""
from typing import Optional

import torch as _torch

from pytorchltr_amd import _C
from pytorchltr_amd import _ties
from pytorchltr_amd._prepare import as_2d as _as_2d
from pytorchltr_amd._prepare import prepare_n as _prepare_n
from pytorchltr_amd._prepare import prepare_scores_f32 as _prepare_scores


def mask_padded_values(xs: _torch.FloatTensor, n: _torch.LongTensor,
                       mask_value: float = -float('inf'),
                       mutate: bool = False):
    """Sets entries j >= n[b] of every row to `mask_value` (reference :6-26).

    With mutate=True, `xs` is overwritten in place and returned (any dtype / layout; contiguous
    fp32 takes the HIP kernel)."""
    _C.require_device(xs, "xs")
    x2 = _as_2d(xs, "xs")
    nn = _prepare_n(n, x2.shape[0])
    B, L = x2.shape
    if mutate and (xs.dtype != _torch.float32 or not xs.is_contiguous()):
        # any other dtype / layout: the same in-place masking with device-side torch ops
        cols = _torch.arange(L, device=xs.device).reshape(1, L)
        x2.masked_fill_(cols >= nn.reshape(B, 1), mask_value)
        return xs
    if mutate:
        src, out = x2, x2
    else:
        src = _prepare_scores(xs)
        out = _torch.empty_like(src)
    if B > 0 and L > 0:
        with _C.device_ctx(src):
            _C.check(_C.lib().ltr_mask_padded_values_f32(
                _C.ptr(src), _C.ptr(nn), B, L, float(mask_value), _C.ptr(out),
                _C.stream_of(src)))
    if mutate:
        return xs
    out = out.reshape(xs.shape)
    return out if xs.dtype == _torch.float32 else out.to(xs.dtype)


def _rank(scores2d, nn, seed=None):
    """seed: None (index order) or (seed, seed_tensor) from _ties.draw_seed."""
    B, L = scores2d.shape
    ranking = _torch.empty(B, L, dtype=_torch.int64, device=scores2d.device)
    if B > 0:
        with _C.device_ctx(scores2d):
            if seed is None:
                _C.check(_C.lib().ltr_rank_by_score_f32(
                    _C.ptr(scores2d), _C.ptr(nn), B, L, _C.ptr(ranking), _C.stream_of(scores2d)))
            else:
                _C.check(_C.lib().ltr_rank_by_score_seed_f32(
                    _C.ptr(scores2d), _C.ptr(nn), seed[0], _C.ptr(seed[1]), B, L, _C.ptr(ranking),
                    _C.stream_of(scores2d)))
    return ranking


def tiebreak_argsort(
        x: _torch.FloatTensor,
        descending: bool = True,
        generator: Optional[_torch.Generator] = None) -> _torch.LongTensor:
    """Per-row argsort (reference :29-45).

    Ties are broken like the reference does: by one random permutation per call, shared by all
    rows, drawn from `generator` (or the default generator of the tensor's device).  Under
    ``pytorchltr_amd.utils.tie_breaking("index")`` and without a generator, ties fall back to
    column order (deterministic)."""
    s = _prepare_scores(x)
    if not descending:
        s = -s
    nn = _torch.full((s.shape[0],), s.shape[1], dtype=_torch.int64, device=s.device)
    return _rank(s, nn, _ties.draw_seed(s.shape[1], s.device, generator))


def rank_by_score(
        scores: _torch.FloatTensor,
        n: _torch.LongTensor,
        generator: Optional[_torch.Generator] = None) -> _torch.LongTensor:
    """Indices that sort each row by decreasing score, padded documents last (reference
    :48-64).  Ties among real documents: random permutation per call (see tiebreak_argsort);
    the padded tail comes out in index order."""
    s = _prepare_scores(scores)
    nn = _prepare_n(n, s.shape[0])
    max_l = _C.max_list_len()
    if s.shape[1] > max_l:
        raise ValueError("list_size %d exceeds the supported maximum %d" % (s.shape[1], max_l))
    return _rank(s, nn, _ties.draw_seed(s.shape[1], s.device, generator))


def _plackettluce_from_uniform(scores, n, uniform):
    """Deterministic core of the sampler: the ranking implied by one uniform(0,1) draw per
    document (keys on the device, then the rank kernel)."""
    s = _prepare_scores(scores)
    nn = _prepare_n(n, s.shape[0])
    B, L = s.shape
    u = uniform.reshape(B, L).to(device=s.device, dtype=_torch.float32).contiguous()
    keys = _torch.empty_like(s)
    if B > 0:
        with _C.device_ctx(s):
            _C.check(_C.lib().ltr_plackettluce_keys_f32(_C.ptr(s), _C.ptr(nn), _C.ptr(u), B, L,
                                                        _C.ptr(keys), _C.stream_of(s)))
    return _rank(keys, nn)          # continuous keys: ties have probability ~0, index order


def rank_by_plackettluce(
        scores: _torch.FloatTensor, n: _torch.LongTensor,
        generator: Optional[_torch.Generator] = None) -> _torch.LongTensor:
    """Samples a ranking from a Plackett-Luce distribution, padded documents last
    (reference :67-91): sort ascending by log(-log u) - log_softmax(scores).

    `generator`, if given, must be a generator of the scores' device."""
    _C.require_device(scores, "scores")
    shape = (scores.shape[0], scores.shape[1])
    kw = {} if generator is None else {"generator": generator}
    u = _torch.rand(shape, device=scores.device, dtype=_torch.float32, **kw)
    return _plackettluce_from_uniform(scores, n, u)


def batch_pairs(x: _torch.Tensor) -> _torch.Tensor:
    """Materialises all pairs: p[b,i,j,0] = x[b,i], p[b,i,j,1] = x[b,j] (reference :94-119).

    Returns a (batch, list_size, list_size, 2) tensor of x's dtype (4- or 8-byte dtypes)."""
    _C.require_device(x, "x")
    x2 = _as_2d(x, "x").contiguous()
    B, L = x2.shape
    esize = x2.element_size()
    if esize not in (4, 8):
        raise TypeError("batch_pairs supports 4- and 8-byte dtypes, got %s" % x2.dtype)
    out = _torch.empty(B, L, L, 2, dtype=x2.dtype, device=x2.device)
    if B > 0 and L > 0:
        with _C.device_ctx(x2):
            _C.check(_C.lib().ltr_batch_pairs(_C.ptr(x2), esize, B, L, _C.ptr(out),
                                              _C.stream_of(x2)))
    return out
