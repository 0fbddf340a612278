"""autograd.Function glue: one fused HIP kernel computes loss AND d loss/d scores in forward;
backward is a row scale by the incoming per-query gradient (ltr_scale_rows_f32).

Host cost matters here: the kernels take 6-10 us at the MSLR shape, so the eager drop-in call
`loss_fn(scores, relevance, n).mean().backward()` is bound by what runs on the host around them.
The common case -- contiguous fp32 device scores, int64/int32/fp32 labels, int64 n -- therefore
skips the generic argument normalisation (`_prepare.prepare`), and the per-shape questions to the
library (does this shape take the split-query workspace path?) are asked once and cached."""
import torch
from torch.autograd.function import once_differentiable

from . import _C
from ._prepare import prepare

_LABEL_CODE = {torch.int64: _C.LABEL_I64, torch.float32: _C.LABEL_F32, torch.int32: _C.LABEL_I32}
_ws_cache = {}          # (kind, B, L) -> workspace bytes of ltr_pairwise_loss_ws_f32 (0 = plain path)
LISTWISE_SOFTMAX = 100  # pseudo-kind of this module: ltr_listwise_softmax_f32 (not a pairwise loss)


def _fast_args(scores, relevance, n):
    """(B, L) when the tensors can go to the C ABI as they are, else None."""
    if scores.dtype is not torch.float32 or not scores.is_cuda or not scores.is_contiguous():
        return None
    d = scores.dim()
    if d == 3:
        if scores.shape[2] != 1:
            return None
    elif d != 2:
        return None
    B, L = scores.shape[0], scores.shape[1]
    if (relevance.dtype not in _LABEL_CODE or not relevance.is_contiguous()
            or relevance.numel() != B * L or relevance.shape[0] != B or relevance.dim() not in (2, 3)
            or relevance.shape[1] != L):
        return None
    if n.dtype is not torch.int64 or n.dim() != 1 or n.shape[0] != B or not n.is_contiguous():
        return None
    dev = scores.device
    if relevance.device != dev or n.device != dev:
        return None
    if L == 0 or L > _C.max_list_len():
        return None
    return B, L


def _workspace_bytes(kind, B, L):
    key = (kind, B, L)
    ws = _ws_cache.get(key)
    if ws is None:
        ws = _ws_cache[key] = int(_C.lib().ltr_pairwise_loss_workspace_bytes(kind, B, L))
    return ws


class PairwiseLossFunction(torch.autograd.Function):
    """loss[b] = L_kind(scores[b,:], relevance[b,:], n[b]); gradient only w.r.t. scores."""

    @staticmethod
    def forward(ctx, scores, relevance, n, kind, sigma):
        fast = _fast_args(scores, relevance, n)
        if fast is not None:
            s, r, nn = scores, relevance, n
            B, L = fast
        else:
            s, r, nn = prepare(scores, relevance, n, allow_f64=True)
            B, L = s.shape
        need_grad = ctx.needs_input_grad[0]
        f64 = s.dtype is torch.float64          # fp64 in -> fp64 arithmetic, like the reference
        if f64 and L > _C.max_list_len_f64():
            raise ValueError("list_size %d exceeds the fp64 maximum %d (fp32 scores: %d)"
                             % (L, _C.max_list_len_f64(), _C.max_list_len()))
        dev = s.device
        loss = torch.empty(B, dtype=s.dtype, device=dev)
        ds = torch.empty((B, L), dtype=s.dtype, device=dev) if need_grad else None
        if B > 0:
            lib = _C.lib()
            with _C.device_ctx(s):
                st = _C.stream_of(s)
                dsp = ds.data_ptr() if need_grad else None
                if kind == LISTWISE_SOFTMAX:
                    if f64:
                        raise TypeError("the listwise softmax loss computes in fp32; cast the scores")
                    rc = lib.ltr_listwise_softmax_f32(s.data_ptr(), r.data_ptr(), _LABEL_CODE[r.dtype],
                                                      nn.data_ptr(), B, L, loss.data_ptr(), dsp, st)
                elif f64:
                    rc = lib.ltr_pairwise_loss_f64(kind, float(sigma), s.data_ptr(), r.data_ptr(),
                                                   _LABEL_CODE[r.dtype], nn.data_ptr(), B, L,
                                                   loss.data_ptr(), dsp, st)
                else:
                    # long lists on a small batch: several workgroups share a query (needs scratch)
                    ws_bytes = _workspace_bytes(kind, B, L)
                    if ws_bytes > 0:
                        ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=dev)
                        rc = lib.ltr_pairwise_loss_ws_f32(
                            kind, float(sigma), s.data_ptr(), r.data_ptr(), _LABEL_CODE[r.dtype],
                            nn.data_ptr(), B, L, loss.data_ptr(), dsp, ws.data_ptr(), ws_bytes, st)
                    else:
                        rc = lib.ltr_pairwise_loss_f32(kind, float(sigma), s.data_ptr(), r.data_ptr(),
                                                       _LABEL_CODE[r.dtype], nn.data_ptr(), B, L,
                                                       loss.data_ptr(), dsp, st)
                if rc != 0:
                    _C.check(rc)
        if need_grad:
            ctx.save_for_backward(ds)
        ctx.in_shape = scores.shape
        ctx.in_dtype = scores.dtype
        return loss if scores.dtype is loss.dtype else loss.to(scores.dtype)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        (ds,) = ctx.saved_tensors
        B, L = ds.shape
        out = torch.empty_like(ds)
        if B > 0:
            lib = _C.lib()
            f64 = ds.dtype is torch.float64
            with _C.device_ctx(ds):
                st = _C.stream_of(ds)
                if (not f64 and grad_out.dtype is torch.float32 and grad_out.dim() == 1
                        and grad_out.stride(0) == 0):
                    # `.sum().backward()`: autograd hands over an expanded scalar
                    rc = lib.ltr_scale_rows_uniform_f32(ds.data_ptr(), grad_out.data_ptr(), B, L,
                                                        out.data_ptr(), st)
                else:
                    go = grad_out
                    if go.dtype is not ds.dtype or go.dim() != 1 or not go.is_contiguous():
                        go = go.reshape(B).to(ds.dtype).contiguous()
                    entry = lib.ltr_scale_rows_f64 if f64 else lib.ltr_scale_rows_f32
                    rc = entry(ds.data_ptr(), go.data_ptr(), B, L, out.data_ptr(), st)
                if rc != 0:
                    _C.check(rc)
        if out.shape != ctx.in_shape:
            out = out.reshape(ctx.in_shape)
        if ctx.in_dtype is not out.dtype:
            out = out.to(ctx.in_dtype)
        return out, None, None, None, None


def pairwise_loss(scores, relevance, n, kind, sigma=1.0):
    # scores that have not been computed yet (fused.LazyScores, what LinearScorer returns in a training step): scores,
    # loss and weight-gradient rows in ONE pass over the features instead of scorer kernel + loss kernel + gradient kernel
    fused = getattr(scores, "fused_loss", None)
    if fused is not None:
        out = fused(relevance, n, kind, sigma)
        if out is not None:
            return out
        scores = scores.materialize()
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
            if kind == LISTWISE_SOFTMAX:
                rc = _C.lib().ltr_listwise_softmax_f32(
                    _C.ptr(s), _C.ptr(r), _C.label_dtype(r), _C.ptr(nn), B, L, _C.ptr(loss),
                    _C.ptr(ds), _C.stream_of(s))
            elif cfg == "split":
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
