"""Listwise softmax cross-entropy (ListNet top-one) on MI355X.

NOT part of the reference: ``pytorchltr/loss/__init__.py:1-7`` exports the seven pairwise classes
only and nothing in the reference's code, tests or docs defines a listwise loss.  The project
brief names "listwise softmax cross-entropy" on the hot path, so this module provides it with the
same call signature as the reference's losses -- **parity unpinned: there is no reference
implementation to pin it to**; the specification is ``include/ltr_hip.h``
(``ltr_listwise_softmax_f32``), restated in fp64 by ``oracle/ltr_oracle.c`` and checked against
finite differences.
"""
import torch as _torch
from torch.autograd.function import once_differentiable as _once

from pytorchltr_amd import _C
from pytorchltr_amd._prepare import prepare as _prepare


class _ListwiseSoftmaxFunction(_torch.autograd.Function):
    """Its own small Function: the pairwise plumbing does not apply -- the kernel has no
    LDS-bound list limit (one wave walks the list), and it computes in fp32 whatever the score dtype
    (fp64 / half scores are cast in and the result cast back, gradients included)."""

    @staticmethod
    def forward(ctx, scores, relevance, n):
        s, r, nn = _prepare(scores, relevance, n, allow_f64=False, limit_len=False)
        B, L = s.shape
        need_grad = ctx.needs_input_grad[0]
        loss = _torch.empty(B, dtype=_torch.float32, device=s.device)
        ds = _torch.empty(B, L, dtype=_torch.float32, device=s.device) if need_grad else None
        if B > 0:
            with _C.device_ctx(s):
                _C.check(_C.lib().ltr_listwise_softmax_f32(_C.ptr(s), _C.ptr(r), _C.label_dtype(r), _C.ptr(nn), B, L,
                                                           _C.ptr(loss), _C.ptr(ds), _C.stream_of(s)))
        if need_grad:
            ctx.save_for_backward(ds)
        ctx.in_shape, ctx.in_dtype = scores.shape, scores.dtype
        return loss if scores.dtype is _torch.float32 else loss.to(scores.dtype)

    @staticmethod
    @_once
    def backward(ctx, grad_out):
        (ds,) = ctx.saved_tensors
        out = ds * grad_out.reshape(-1, 1).to(ds.dtype)
        return out.reshape(ctx.in_shape).to(ctx.in_dtype), None, None


class ListwiseSoftmaxLoss(_torch.nn.Module):
    r"""ListNet top-one cross-entropy between the label and the score distributions of a query:

    .. math::
        l(\mathbf{s}, \mathbf{y}) = -\sum_{j < n} \mathrm{softmax}(\mathbf{y})_j
        \ln \mathrm{softmax}(\mathbf{s})_j

    over the ``n`` real documents (padded documents take no part); ``n = 0`` gives 0.

    Shape:
        - scores: :math:`(N, \texttt{list\_size})` or :math:`(N, \texttt{list\_size}, 1)`
        - relevance: :math:`(N, \texttt{list\_size})`
        - n: :math:`(N)`
        - output: :math:`(N)`
    """

    def __init__(self):
        super().__init__()

    def forward(self, scores: _torch.FloatTensor, relevance: _torch.LongTensor,
                n: _torch.LongTensor) -> _torch.FloatTensor:
        return _ListwiseSoftmaxFunction.apply(scores, relevance, n)


ListNetLoss = ListwiseSoftmaxLoss
