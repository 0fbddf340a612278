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
            lib = _C.lib()
            with _C.device_ctx(s):
                # long lists on a small batch: several workgroups share a query (needs scratch)
                ws_bytes = 0 if f64 else lib.ltr_pairwise_loss_workspace_bytes(kind, B, L)
                if ws_bytes > 0:
                    ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=s.device)
                    _C.check(lib.ltr_pairwise_loss_ws_f32(
                        kind, float(sigma), _C.ptr(s), _C.ptr(r), _C.label_dtype(r), _C.ptr(nn), B, L,
                        _C.ptr(loss), _C.ptr(ds), _C.ptr(ws), ws_bytes, _C.stream_of(s)))
                else:
                    entry = lib.ltr_pairwise_loss_f64 if f64 else lib.ltr_pairwise_loss_f32
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
    msplit) forces a launch shape (tests / tuning); cfg = "split" takes the workspace entry point
    (several workgroups per query when the library decides that pays)."""
    s, r, nn = prepare(scores, relevance, n)
    B, L = s.shape
    loss = torch.empty(B, dtype=torch.float32, device=s.device)
    ds = torch.empty(B, L, dtype=torch.float32, device=s.device)
    if B > 0:
        with _C.device_ctx(s):
            if cfg == "split":
                ws_bytes = _C.lib().ltr_pairwise_loss_workspace_bytes(kind, B, L)
                ws = torch.empty(max(ws_bytes, 4) // 4, dtype=torch.float32, device=s.device)
                rc = _C.lib().ltr_pairwise_loss_ws_f32(
                    kind, float(sigma), _C.ptr(s), _C.ptr(r), _C.label_dtype(r), _C.ptr(nn),
                    B, L, _C.ptr(loss), _C.ptr(ds), _C.ptr(ws), ws_bytes, _C.stream_of(s))
            elif cfg is None:
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
