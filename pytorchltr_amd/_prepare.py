"""Argument normalisation shared by losses and metrics (host logic, no kernels).

Follows the reference's conventions: scores arrive as (B, L) or (B, L, 1)
(loss/pairwise_additive.py:61-65, utils/tensor_operations.py:62-63), relevance as (B, L) or
(B, L, 1) integer or float, n as (B) int64 (int32 accepted).  Inputs are never mutated.
"""
import torch

from . import _C


def as_2d(t, what):
    if t.dim() == 3 and t.shape[2] == 1:
        return t.reshape(t.shape[0], t.shape[1])
    if t.dim() == 2:
        return t
    raise ValueError("`%s` must have shape (batch, list_size) or (batch, list_size, 1), got %s"
                     % (what, tuple(t.shape)))


def prepare_scores(scores):
    _C.require_device(scores, "scores")
    s = as_2d(scores, "scores")
    if s.dtype not in (torch.float32, torch.float64):
        if not s.dtype.is_floating_point:
            raise TypeError("scores must be floating point, got %s" % s.dtype)
        s = s.float()          # half / bfloat16: arithmetic is fp32, result is cast back
    return s.contiguous()


def prepare_scores_f32(scores):
    """fp32-only consumers (metrics, helpers, fused scorer)."""
    s = prepare_scores(scores)
    return s if s.dtype == torch.float32 else s.float()


def prepare_relevance(relevance, like):
    _C.require_device(relevance, "relevance")
    r = as_2d(relevance, "relevance")
    if r.shape != like.shape:
        raise ValueError("relevance shape %s does not match scores shape %s"
                         % (tuple(r.shape), tuple(like.shape)))
    if r.dtype not in (torch.int64, torch.float32, torch.int32):
        r = r.float()
    return r.contiguous()


def prepare_n(n, batch):
    _C.require_device(n, "n")
    if n.dim() != 1 or n.shape[0] != batch:
        raise ValueError("`n` must have shape (batch,) = (%d,), got %s" % (batch, tuple(n.shape)))
    if n.dtype != torch.int64:
        if n.dtype.is_floating_point:
            raise TypeError("`n` must be an integer tensor, got %s" % n.dtype)
        n = n.long()
    return n.contiguous()


def prepare(scores, relevance, n, allow_f64=False, limit_len=True):
    s = prepare_scores(scores) if allow_f64 else prepare_scores_f32(scores)
    r = prepare_relevance(relevance, s)
    nn = prepare_n(n, s.shape[0])
    if not (s.device == r.device == nn.device):
        raise RuntimeError("scores, relevance and n must be on the same device")
    max_l = _C.max_list_len()
    if limit_len and s.shape[1] > max_l:
        raise ValueError("list_size %d exceeds the supported maximum %d" % (s.shape[1], max_l))
    if s.shape[1] == 0:
        raise ValueError("list_size must be positive")
    return s, r, nn
