"""Linear scorer fused with a ranking loss (one HIP launch for scores, loss and weight grads).

The reference has no such module: its users write
``loss_fn(torch.nn.Linear(F, 1)(xs), ys, n)`` (examples/01-basic-usage.py:66-75,
tests/test_integration.py:42).  ``FusedLinearLoss`` is that exact composition -- same
parameters (``weight`` (1, F), ``bias`` (1,), state_dict-compatible with ``nn.Linear(F, 1)``),
same per-query output -- computed by ``ltr_linear_partials_f32`` so the (B, L, F) feature
tensor crosses HBM once instead of twice plus the score round trip.
"""
import math

import torch
from torch.autograd.function import once_differentiable

from . import _C
from ._prepare import prepare_n, prepare_relevance

_KIND_BY_NAME = {
    "hinge": _C.HINGE, "dcg_hinge": _C.DCG_HINGE, "logistic": _C.LOGISTIC,
    "arp1": _C.ARP1, "arp2": _C.ARP2, "ndcg1": _C.NDCG1, "ndcg2": _C.NDCG2,
}


def _resolve_loss(loss):
    """Accepts a kind name or an instance of a pytorchltr_amd loss module."""
    if isinstance(loss, str):
        return _KIND_BY_NAME[loss], 1.0
    kind = getattr(loss, "_kind", None)
    if kind is None:
        raise TypeError("loss must be a kind name or a pytorchltr_amd.loss module")
    return kind, float(getattr(loss, "sigma", 1.0))


def _prepare_features(xs):
    _C.require_device(xs, "xs")
    if xs.dim() != 3:
        raise ValueError("features must have shape (batch, list_size, features)")
    if xs.dtype != torch.float32:
        xs = xs.float()
    return xs.contiguous()


class _LinearLossFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xs, weight, bias, relevance, n, kind, sigma, want_scores):
        X = _prepare_features(xs)
        B, L, F = X.shape
        if weight.numel() != F:
            raise ValueError("weight has %d elements, features have %d" % (weight.numel(), F))
        W = weight.detach().reshape(F).float().contiguous()
        bvec = None if bias is None else bias.detach().reshape(1).float().contiguous()
        r = prepare_relevance(relevance, X[:, :, 0])
        nn = prepare_n(n, B)
        loss = torch.empty(B, dtype=torch.float32, device=X.device)
        ws_bytes = _C.lib().ltr_linear_workspace_bytes(B, L, F)
        ws = torch.empty(max(ws_bytes, 4) // 4, dtype=torch.float32, device=X.device)
        part = ws[:(F + 1) * B].view(F + 1, B)        # (F+1, B) partials; tail = kernel scratch
        scores = torch.empty(B, L, dtype=torch.float32, device=X.device) if want_scores else None
        if B > 0:
            with _C.device_ctx(X):
                _C.check(_C.lib().ltr_linear_partials_f32(
                    kind, float(sigma), _C.ptr(X), _C.ptr(W), _C.ptr(bvec), _C.ptr(r),
                    _C.label_dtype(r), _C.ptr(nn), B, L, F, _C.ptr(loss), _C.ptr(scores),
                    _C.ptr(part), _C.stream_of(X)))
        ctx.save_for_backward(part)
        ctx.w_shape = weight.shape
        ctx.has_bias = bias is not None
        if want_scores:
            ctx.mark_non_differentiable(scores)
            return loss, scores
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_loss, *unused):
        (part,) = ctx.saved_tensors
        F1, B = part.shape
        F = F1 - 1
        go = grad_loss.reshape(B).float().contiguous()
        dW = torch.empty(F, dtype=torch.float32, device=part.device)
        db = torch.empty(1, dtype=torch.float32, device=part.device)
        with _C.device_ctx(part):
            _C.check(_C.lib().ltr_linear_reduce_f32(_C.ptr(part), _C.ptr(go), B, F, _C.ptr(dW),
                                                    _C.ptr(db), _C.stream_of(part)))
        return (None, dW.reshape(ctx.w_shape), db if ctx.has_bias else None,
                None, None, None, None, None)


class FusedLinearLoss(torch.nn.Module):
    """``loss_fn(Linear(in_features, 1)(xs), relevance, n)`` as one fused op.

    Gradients flow to ``weight`` and ``bias`` (not to ``xs``: features are data).
    """

    def __init__(self, in_features, loss="hinge", bias=True):
        super().__init__()
        self.in_features = in_features
        self.kind, self.sigma = _resolve_loss(loss)
        self.weight = torch.nn.Parameter(torch.empty(1, in_features))
        self.bias = torch.nn.Parameter(torch.empty(1)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        # same init as torch.nn.Linear(in_features, 1)
        torch.nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            bound = 1.0 / math.sqrt(self.in_features)
            torch.nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, xs, relevance, n, return_scores=False):
        return _LinearLossFunction.apply(xs, self.weight, self.bias, relevance, n, self.kind,
                                         self.sigma, bool(return_scores))


def linear_loss_step(xs, weight, bias, relevance, n, loss="hinge", grad_out=None,
                     return_scores=False, return_loss_sum=False):
    """One fused fwd+bwd step without autograd: returns (loss[B], dW[F], db[1][, scores][, loss_sum]).

    dW/db are the gradients of ``sum_b grad_out[b] * loss[b]``; grad_out=None means the
    ``.mean()`` of the reference's training loop (1/B each).  Two launches: the fused
    scorer+loss kernel and the cross-query reduction (which also totals the loss)."""
    kind, sigma = _resolve_loss(loss)
    X = _prepare_features(xs)
    B, L, F = X.shape
    W = weight.detach().reshape(F).float().contiguous()
    bvec = None if bias is None else bias.detach().reshape(1).float().contiguous()
    r = prepare_relevance(relevance, X[:, :, 0])
    nn = prepare_n(n, B)
    lossv = torch.empty(B, dtype=torch.float32, device=X.device)
    dW = torch.empty(F, dtype=torch.float32, device=X.device)
    db = torch.empty(1, dtype=torch.float32, device=X.device)
    lsum = torch.zeros(1, dtype=torch.float32, device=X.device) if return_loss_sum else None
    scores = torch.empty(B, L, dtype=torch.float32, device=X.device) if return_scores else None
    ws_bytes = _C.lib().ltr_linear_workspace_bytes(B, L, F)
    ws = torch.empty(max(ws_bytes, 4) // 4, dtype=torch.float32, device=X.device)
    go = None if grad_out is None else grad_out.reshape(B).float().contiguous()
    with _C.device_ctx(X):
        st = _C.stream_of(X)
        _C.check(_C.lib().ltr_linear_partials_f32(
            kind, float(sigma), _C.ptr(X), _C.ptr(W), _C.ptr(bvec), _C.ptr(r), _C.label_dtype(r),
            _C.ptr(nn), B, L, F, _C.ptr(lossv), _C.ptr(scores), _C.ptr(ws), st))
        _C.check(_C.lib().ltr_linear_reduce_loss_f32(
            _C.ptr(ws), _C.ptr(go), _C.ptr(lossv), B, F, _C.ptr(dW), _C.ptr(db), _C.ptr(lsum), st))
    out = (lossv, dW, db)
    if return_scores:
        out = out + (scores,)
    if return_loss_sum:
        out = out + (lsum,)
    return out
