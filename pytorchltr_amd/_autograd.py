"""autograd.Function glue: one fused HIP kernel computes loss AND d loss/d scores in forward;
backward is a row scale by the incoming per-query gradient (ltr_scale_rows_f32)."""
import torch
from torch.autograd.function import once_differentiable

from . import _C
from ._prepare import prepare


class PairwiseLossFunction(torch.autograd.Function):
    """loss[b] = L_kind(scores[b,:], relevance[b,:], n[b]); gradient only w.r.t. scores."""

    @staticmethod
    def forward(ctx, scores, relevance, n, kind, sigma):
        s, r, nn = prepare(scores, relevance, n, allow_f64=True)
        B, L = s.shape
        need_grad = ctx.needs_input_grad[0]
        f64 = s.dtype == torch.float64          # fp64 in -> fp64 arithmetic, like the reference
        loss = torch.empty(B, dtype=s.dtype, device=s.device)
        ds = torch.empty(B, L, dtype=s.dtype, device=s.device) if need_grad else None
        if B > 0:
            entry = _C.lib().ltr_pairwise_loss_f64 if f64 else _C.lib().ltr_pairwise_loss_f32
            with _C.device_ctx(s):
                _C.check(entry(kind, float(sigma), _C.ptr(s), _C.ptr(r), _C.label_dtype(r),
                               _C.ptr(nn), B, L, _C.ptr(loss), _C.ptr(ds), _C.stream_of(s)))
        if need_grad:
            ctx.save_for_backward(ds)
        ctx.in_shape = scores.shape
        ctx.in_dtype = scores.dtype
        return loss if scores.dtype == loss.dtype else loss.to(scores.dtype)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        (ds,) = ctx.saved_tensors
        B, L = ds.shape
        go = grad_out.reshape(B).to(ds.dtype).contiguous()
        out = torch.empty_like(ds)
        if B > 0:
            entry = (_C.lib().ltr_scale_rows_f64 if ds.dtype == torch.float64
                     else _C.lib().ltr_scale_rows_f32)
            with _C.device_ctx(ds):
                _C.check(entry(_C.ptr(ds), _C.ptr(go), B, L, _C.ptr(out), _C.stream_of(ds)))
        out = out.reshape(ctx.in_shape)
        if ctx.in_dtype != out.dtype:
            out = out.to(ctx.in_dtype)
        return out, None, None, None, None


def pairwise_loss(scores, relevance, n, kind, sigma=1.0):
    return PairwiseLossFunction.apply(scores, relevance, n, kind, sigma)


def pairwise_loss_and_grad(scores, relevance, n, kind, sigma=1.0, cfg=None):
    """Direct (no autograd) call: returns (loss[B], dscores[B,L]).  `cfg` = (owners, dpt,
    msplit) forces a launch shape (tests / tuning)."""
    s, r, nn = prepare(scores, relevance, n)
    B, L = s.shape
    loss = torch.empty(B, dtype=torch.float32, device=s.device)
    ds = torch.empty(B, L, dtype=torch.float32, device=s.device)
    if B > 0:
        with _C.device_ctx(s):
            if cfg is None:
                rc = _C.lib().ltr_pairwise_loss_f32(
                    kind, float(sigma), _C.ptr(s), _C.ptr(r), _C.label_dtype(r), _C.ptr(nn),
                    B, L, _C.ptr(loss), _C.ptr(ds), _C.stream_of(s))
            else:
                rc = _C.lib().ltr_pairwise_loss_f32_cfg(
                    kind, float(sigma), _C.ptr(s), _C.ptr(r), _C.label_dtype(r), _C.ptr(nn),
                    B, L, _C.ptr(loss), _C.ptr(ds), int(cfg[0]), int(cfg[1]), int(cfg[2]),
                    _C.stream_of(s))
            _C.check(rc)
    return loss, ds
