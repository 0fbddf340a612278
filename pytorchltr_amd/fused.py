"""Linear scorer fused with a ranking loss (one HIP launch for scores, loss and weight grads).

The reference has no such module: its users write
``loss_fn(torch.nn.Linear(F, 1)(xs), ys, n)`` (examples/01-basic-usage.py:66-75,
tests/test_integration.py:42).  ``FusedLinearLoss`` is that exact composition -- same
parameters (``weight`` (1, F), ``bias`` (1,), state_dict-compatible with ``nn.Linear(F, 1)``),
same per-query output -- computed by ``ltr_linear_partials_f32`` so the (B, L, F) feature
tensor crosses HBM once instead of twice plus the score round trip.
"""
import ctypes
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


def _padded_rows(xs):
    """A (B, L, F) view whose rows sit a multiple of four floats apart -- F4 = (F + 3) & ~3 > F, what
    ``RaggedQueries(pad_features_to=4)`` collates (feature counts like Example3's 5, LETOR's 45 / 46: rows that are not
    whole float4 otherwise run the scalar kernels) -- as the (B, L, F4) tensor it lives in, or None.  The hidden
    columns meet zero weights, so any FINITE content is harmless (the collate writes zeros)."""
    if xs.dim() != 3 or xs.dtype is not torch.float32 or not xs.is_cuda:
        return None
    # only batches the collate of this package made: there the hidden columns are zeros.  A user's own view with the same
    # strides (big[..., :5] of an 8-column tensor) may hide NaN / Inf / metadata columns, which nn.Linear ignores and a
    # zero weight does not (NaN * 0): those are copied to a contiguous batch like any other view
    if not getattr(xs, "_ltr_zero_padded_rows", False):
        return None
    B, L, F = xs.shape
    F4 = (F + 3) & ~3
    if F4 == F or B == 0 or L == 0 or xs.stride(2) != 1 or xs.stride(1) != F4 or (B > 1 and xs.stride(0) != L * F4):
        return None
    off = xs.storage_offset()
    if off % 4 != 0 or xs.untyped_storage().nbytes() < 4 * (off + B * L * F4):
        return None
    return xs.as_strided((B, L, F4), (L * F4, F4, 1), off)


def _prepare_features(xs):
    """The feature batch as the kernels take it: contiguous fp32 (B, L, F') on the device -- F' = F, or the padded
    row width of a `_padded_rows` view (the callers zero-pad the weights to match and crop the gradient)."""
    if xs.dtype is torch.float32 and xs.is_cuda and xs.dim() == 3 and xs.is_contiguous():
        return xs                                   # the common case: nothing to do
    _C.require_device(xs, "xs")
    if xs.dim() != 3:
        raise ValueError("features must have shape (batch, list_size, features)")
    if xs.dtype != torch.float32:
        xs = xs.float()
    padded = _padded_rows(xs)
    return padded if padded is not None else xs.contiguous()


def _pad_weight(W, Fp):
    """W (F elements) zero-padded to the feature tensor's row width Fp."""
    if W.numel() == Fp:
        return W
    out = torch.zeros(Fp, dtype=torch.float32, device=W.device)
    out[:W.numel()] = W
    return out


def _flat_f32(t, count):
    """A parameter as a contiguous fp32 vector of `count` elements (no copy in the common case)."""
    if t.dtype is torch.float32 and t.is_contiguous() and t.numel() == count:
        return t
    return t.detach().reshape(count).float().contiguous()


class _ShapeOnly:
    def __init__(self, *shape):
        self.shape = torch.Size(shape)


def _labels_and_n(relevance, n, B, L, dev):
    """relevance / n as the C ABI takes them; the generic normalisation only when needed."""
    if (relevance.dtype in _LABEL_CODE and relevance.is_contiguous() and relevance.device == dev
            and relevance.dim() in (2, 3) and relevance.shape[0] == B and relevance.shape[1] == L
            and relevance.numel() == B * L):
        r = relevance
    else:
        r = prepare_relevance(relevance, _ShapeOnly(B, L))
        if r.device != dev:
            raise RuntimeError("features and relevance must be on the same device")
    if (n.dtype is torch.int64 and n.dim() == 1 and n.shape[0] == B and n.is_contiguous()
            and n.device == dev):
        nn = n
    else:
        nn = prepare_n(n, B)
        if nn.device != dev:
            raise RuntimeError("features and n must be on the same device")
    return r, nn


_LABEL_CODE = {torch.int64: _C.LABEL_I64, torch.float32: _C.LABEL_F32, torch.int32: _C.LABEL_I32}
_linear_ws_cache = {}       # (B, L, F) -> ltr_linear_workspace_bytes


def _linear_ws_bytes(B, L, F):
    key = (B, L, F)
    v = _linear_ws_cache.get(key)
    if v is None:
        v = _linear_ws_cache[key] = max(int(_C.lib().ltr_linear_workspace_bytes(B, L, F)), 4)
    return v


class _LinearLossFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xs, weight, bias, relevance, n, kind, sigma, want_scores):
        X = _prepare_features(xs)
        B, L, F = X.shape                           # (F: the row width in memory, >= the logical feature count)
        Fw = xs.shape[2]
        if weight.numel() != Fw:
            raise ValueError("weight has %d elements, features have %d" % (weight.numel(), Fw))
        dev = X.device
        r, nn = _labels_and_n(relevance, n, B, L, dev)
        # parameters that pytorchltr_amd.optim.SGD updates lazily: the lazy launch (this batch's rows + the previous
        # step's update in ONE launch); its backward and the optimizer step then launch nothing
        st = getattr(weight, "_ltr_lazy", None)
        if st is not None:
            if (not want_scores and bias is not None and Fw == F and ctx.needs_input_grad[1] and ctx.needs_input_grad[2]
                    and getattr(bias, "_ltr_lazy", None) is st):
                loss = st.forward(ctx, X, r, _LABEL_CODE[r.dtype], nn, kind, sigma)
                if loss is not None:
                    ctx.lazy_loss = loss.detach()            # (an alias without the autograd node: no reference cycle)
                    ctx.save_for_backward(X, r, nn)
                    ctx.lazy_args = (kind, sigma)
                    ctx.dims = (B, F)
                    ctx.Fw = Fw
                    ctx.w_shape = weight.shape
                    ctx.has_bias = True
                    return loss
            st.settle()                                      # the ordinary path reads the weights: no update may be pending
        W = _pad_weight(_flat_f32(weight, Fw), F)
        bvec = None if bias is None else _flat_f32(bias, 1)
        loss = torch.empty(B, dtype=torch.float32, device=dev)
        # (B, PF) partials, then the kernel's scratch
        ws = torch.empty(_linear_ws_bytes(B, L, F) // 4, dtype=torch.float32, device=dev)
        scores = torch.empty((B, L), dtype=torch.float32, device=dev) if want_scores else None
        if B > 0:
            with _C.device_ctx(X):
                rc = _C.lib().ltr_linear_partials_f32(
                    kind, float(sigma), X.data_ptr(), W.data_ptr(),
                    None if bvec is None else bvec.data_ptr(), r.data_ptr(), _LABEL_CODE[r.dtype],
                    nn.data_ptr(), B, L, F, loss.data_ptr(),
                    None if scores is None else scores.data_ptr(), ws.data_ptr(), _C.stream_of(X))
                if rc != 0:
                    _C.check(rc)
        ctx.save_for_backward(ws)
        ctx.dims = (B, F)
        ctx.Fw = Fw
        ctx.w_shape = weight.shape
        ctx.has_bias = bias is not None
        if want_scores:
            ctx.mark_non_differentiable(scores)
            return loss, scores
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_loss, *unused):
        lz = getattr(ctx, "lazy", None)
        if lz is not None:
            st = lz[0]
            out = st.backward(ctx, grad_loss)
            if out is not None:
                return (None, out[0], out[1], None, None, None, None, None)
            # an upstream gradient that is not one broadcast scalar (per-query weights): the rows again, the ordinary way
            # (the weights have not moved since the forward pass: no step can have been taken on this gradient)
            if lz[1] != st.gen:
                raise RuntimeError("the gradient rows of this forward pass are gone (a second backward pass through a "
                                   "lazily updated scorer after later training steps)")
            X, r, nn = ctx.saved_tensors
            B, L, F = X.shape
            kind, sigma = ctx.lazy_args
            ws = torch.empty(_linear_ws_bytes(B, L, F) // 4, dtype=torch.float32, device=X.device)
            scratch = torch.empty(B, dtype=torch.float32, device=X.device)
            with _C.device_ctx(X):
                _C.check(_C.lib().ltr_linear_partials_f32(
                    kind, float(sigma), X.data_ptr(), st.w_raw.data_ptr(), st.b_raw.data_ptr(), r.data_ptr(),
                    _LABEL_CODE[r.dtype], nn.data_ptr(), B, L, F, scratch.data_ptr(), None, ws.data_ptr(), _C.stream_of(X)))
        else:
            (ws,) = ctx.saved_tensors
        B, F = ctx.dims
        go = grad_loss
        dW = torch.empty(F, dtype=torch.float32, device=ws.device)
        db = torch.empty(1, dtype=torch.float32, device=ws.device)
        # `.mean().backward()` / `.sum().backward()`: autograd hands over an expanded scalar (stride 0) -- the reduction
        # reads it where it is instead of a (B,) copy being made first
        bcast = go.dtype is torch.float32 and go.dim() == 1 and go.stride(0) == 0
        if not bcast and (go.dtype is not torch.float32 or go.dim() != 1 or not go.is_contiguous()):
            go = go.reshape(B).float().contiguous()
        with _C.device_ctx(ws):
            entry = _C.lib().ltr_linear_reduce_bcast_f32 if bcast else _C.lib().ltr_linear_reduce_f32
            rc = entry(ws.data_ptr(), go.data_ptr(), B, F, dW.data_ptr(), db.data_ptr(), _C.stream_of(ws))
            if rc != 0:
                _C.check(rc)
        return (None, dW[:ctx.Fw].reshape(ctx.w_shape), db if ctx.has_bias else None, None, None, None, None, None)


_pieces_cache = {}


def _prefer_pieces(kind, B, L, F):
    """True where the three-kernel pieces (streaming scorer, split-query loss, streaming weight gradient) beat the
    one-kernel fused path: a few long lists that neither the cluster nor the parts kernel takes (cached per shape)."""
    key = (kind, B, L, F)
    v = _pieces_cache.get(key)
    if v is None:
        v = False
        if B > 0 and _C.lib().ltr_pairwise_loss_workspace_bytes(kind, B, L) != 0 and \
                _C.lib().ltr_linear_fused_plan(kind, B, L, F) not in (_C.PLAN_CLUSTER, _C.PLAN_PARTS):
            cus = torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count
            heavy_pairs = kind not in (_C.HINGE, _C.DCG_HINGE)
            v = (2 * B <= cus) or (heavy_pairs and B <= cus and L > 768)
        _pieces_cache[key] = v
    return v


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
        if xs.dim() == 3 and xs.is_cuda and self._pieces_cached(xs.shape[0], xs.shape[1]):
            # a few long lists: one workgroup per query cannot fill the GPU; the balanced pieces
            # (streaming scorer, split-query loss, streaming weight gradient) are faster
            from ._autograd import PairwiseLossFunction
            scores = _LinearScoreFunction.apply(xs, self.weight, self.bias, n)
            loss = PairwiseLossFunction.apply(scores, relevance, n, self.kind, self.sigma)
            return (loss, scores.detach().squeeze(-1)) if return_scores else loss
        return _LinearLossFunction.apply(xs, self.weight, self.bias, relevance, n, self.kind,
                                         self.sigma, bool(return_scores))

    def _pieces_cached(self, B, L):
        # measured on MI355X, fused kernel -> pieces, us (B x L x F; hinge / logistic / LambdaNDCG2):
        #   32 x 1000 x 220:  78/136/173 -> 46/52/77      128 x 1000 x 220: 84/145/181 -> 55/66/93
        #   256 x 1000 x 220: 86/145/183 -> 87/113/147    384 x 1000 x 220: 90/149/186 -> 114/149/204
        #   128 x 600 x 136:  41/64/90 -> 39/47/72        256 x 600 x 136:  44/66/93 -> 48/60/89
        # the cluster kernel (features once, a query over several workgroups) beats the pieces wherever it applies:
        # 32 x 1000 x 220 hinge 32 vs 46 us, logistic 43 vs 52 -- and so does the parts kernel (round 4, module forward
        # + backward replayed, pieces vs fused: 64 x 512 x 700 LambdaNDCG2 91 vs 48 us, 100 x 1000 x 700 hinge 120 vs 78)
        return _prefer_pieces(self.kind, B, L, self.in_features)

    def _prefer_pieces(self, B, L):
        return _prefer_pieces(self.kind, B, L, self.in_features)


def linear_loss_step(xs, weight, bias, relevance, n, loss="hinge", grad_out=None,
                     return_scores=False, return_loss_sum=False):
    """One fused fwd+bwd step without autograd: returns (loss[B], dW[F], db[1][, scores][, loss_sum]).

    dW/db are the gradients of ``sum_b grad_out[b] * loss[b]``; grad_out=None means the
    ``.mean()`` of the reference's training loop (1/B each).  Two launches: the fused
    scorer+loss kernel and the cross-query reduction (which also totals the loss)."""
    kind, sigma = _resolve_loss(loss)
    X = _prepare_features(xs)
    B, L, F = X.shape                               # (F: the row width in memory -- a padded view's F4)
    Fw = xs.shape[2]
    W = _pad_weight(_flat_f32(weight, Fw), F)
    bvec = None if bias is None else _flat_f32(bias, 1)
    r, nn = _labels_and_n(relevance, n, B, L, X.device)
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
    out = (lossv, dW[:Fw], db)
    if return_scores:
        out = out + (scores,)
    if return_loss_sum:
        out = out + (lsum,)
    return out


class LazySGD:
    """The reference's training loop body, one batch per call as the DataLoader hands them over, with the optimiser
    step applied LAZILY (include/ltr_hip.h: ltr_linear_sgd_lazy_step_f32; examples/01-basic-usage.py:66-75):

        opt = LazySGD(weight, bias, lr, loss="hinge")
        for xs, ys, n in loader:
            opt.step(xs, ys, n)         # loss_fn(Linear(F, 1)(xs), ys, n).mean().backward(); SGD.step()
        opt.flush()                     # the last batch's update; also before the weights are read in between

    `step` launches ONE kernel: it computes this batch's per-query gradient rows and applies the PREVIOUS batch's
    update inside the same launch (its first workgroups sum that batch's rows and hand the new weights to every
    workgroup before the dot products), so the reduction launch and a kernel boundary per step are gone.  Weights,
    gradients and losses are bit-identical to the eager step's.  `weight` (F elements) / `bias` (one element) are
    contiguous fp32 device tensors, updated in place; every batch must have the same (B, L, F).

    Data parallel (one process per GPU, every rank steps through its shard of each global batch):
    ``LazySGD(weight, bias, lr, loss, mailbox=MailboxOverlap(F, count=B_shard, device=dev))`` -- the step stays ONE launch
    per rank: the reducer workgroups all-reduce their column sums through the peers' mailboxes inside the launch
    (include/ltr_hip.h: ltr_linear_sgd_lazy_step_dp_f32); all ranks hold bit-identical weights after every step and
    the update is the gradient of the mean over the GLOBAL batch (`mailbox.global_count` queries)."""

    def __init__(self, weight, bias, lr, loss="hinge", mailbox=None):
        self.kind, self.sigma = _resolve_loss(loss)
        if weight.dtype is not torch.float32 or not weight.is_contiguous() or not weight.is_cuda:
            raise ValueError("weight must be a contiguous fp32 device tensor (it is updated in place)")
        if bias is None or bias.dtype is not torch.float32 or bias.numel() != 1 or not bias.is_cuda:
            raise ValueError("bias must be an fp32 device tensor of one element (it is updated in place)")
        self.weight, self.bias, self.lr = weight, bias, float(lr)
        if mailbox is not None and not getattr(mailbox, "ok", False):
            raise ValueError("the mailbox is not usable (%s): take RcclOverlap.sgd_step instead" % getattr(mailbox, "why", "?"))
        self.mailbox = mailbox
        self._mb = None if mailbox is None else mailbox.mbox
        self._scale = 0.0 if mailbox is None else 1.0 / float(mailbox.global_count)
        self.pending = 0
        self.shape = None
        self.loss = self.bucket = self.ws = None

    def step(self, xs, relevance, n):
        X = _prepare_features(xs)
        B, L, F = X.shape
        if self.shape is None:
            if self.weight.numel() != F:
                raise ValueError("weight has %d elements, the rows %d features" % (self.weight.numel(), F))
            self.shape = (B, L, F)
            dev = X.device
            self.loss = torch.empty(B, dtype=torch.float32, device=dev)
            self.bucket = torch.zeros(F + 2, dtype=torch.float32, device=dev)
            nb = _C.lib().ltr_linear_workspace_bytes(B, L, F)
            self.ws = torch.empty(max(nb, 4) // 4, dtype=torch.float32, device=dev)
        elif tuple(X.shape) != self.shape:
            raise ValueError("every batch must have the shape of the first one %r (flush() and make a new LazySGD otherwise)" % (self.shape,))
        r, nn = _labels_and_n(relevance, n, B, L, X.device)
        with _C.device_ctx(X):
            _C.check(_C.lib().ltr_linear_sgd_lazy_step_dp_f32(
                self.kind, float(self.sigma), _C.ptr(X), self.weight.data_ptr(), self.bias.data_ptr(), _C.ptr(r),
                _C.label_dtype(r), _C.ptr(nn), B, L, F, self.lr, self.loss.data_ptr(), self.bucket.data_ptr(),
                self.ws.data_ptr(), self.ws.numel() * 4, self.pending, self._scale, None, 0, self._mb, _C.stream_of(X)))
        for t in (X, r, nn):
            t.record_stream(torch.cuda.current_stream(X.device))
        self.pending = B

    def flush(self):
        """Applies the pending update; returns (mean loss, dW | db) of the last batch."""
        if self.pending:
            B, L, F = self.shape
            with _C.device_ctx(self.ws):
                _C.check(_C.lib().ltr_linear_sgd_flush_dp_f32(
                    self.kind, self.weight.data_ptr(), self.bias.data_ptr(), self.pending, L, F, self.lr, self._scale, None, 0,
                    self.loss.data_ptr(), self.bucket.data_ptr(), self.ws.data_ptr(), self._mb, _C.stream_of(self.ws)))
            self.pending = 0
        if self.bucket is None:
            return None
        F = self.shape[2]
        count = float(self.shape[0]) if self.mailbox is None else float(self.mailbox.global_count)
        return self.bucket[F + 1] / count, self.bucket[:F + 1]


# ---------------------------------------------------------------------------------------------
# The Linear(F, 1) scorer on its own (unfused drop-in composition)
# ---------------------------------------------------------------------------------------------
class _LinearScoreFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xs, weight, bias, n):
        X = _prepare_features(xs)
        B, L, F = X.shape                           # (F: the row width in memory -- a padded view's F4)
        Fw = xs.shape[2]
        W = _pad_weight(weight.detach().reshape(Fw).float().contiguous(), F)
        ctx.Fw = Fw
        bvec = None if bias is None else bias.detach().reshape(1).float().contiguous()
        nn = None if n is None else prepare_n(n, B)
        scores = torch.empty(B, L, dtype=torch.float32, device=X.device)
        if B > 0:
            with _C.device_ctx(X):
                _C.check(_C.lib().ltr_linear_scores_f32(_C.ptr(X), _C.ptr(W), _C.ptr(bvec), _C.ptr(nn),
                                                        B, L, F, _C.ptr(scores), _C.stream_of(X)))
        # (the layer's input needs a gradient only when it is not the feature batch itself -- a hidden
        # layer's output: the weight is kept for grad_xs = grad_scores (x) weight in that case)
        ctx.xs_grad = bool(ctx.needs_input_grad[0])
        ctx.save_for_backward(X, nn if nn is not None else torch.empty(0, device=X.device),
                              W if ctx.xs_grad else torch.empty(0, device=X.device))
        ctx.xs_shape = xs.shape
        ctx.has_n = nn is not None
        ctx.w_shape = weight.shape
        ctx.has_bias = bias is not None
        return scores.unsqueeze(-1)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_scores):
        X, nn, Wsaved = ctx.saved_tensors
        B, L, F = X.shape
        g = grad_scores.reshape(B, L).float().contiguous()
        lib = _C.lib()
        out = torch.empty(F + 1, dtype=torch.float32, device=X.device)
        ws_bytes = lib.ltr_linear_grad_workspace_bytes(B, L, F)
        ws = torch.empty(max(ws_bytes, 4) // 4, dtype=torch.float32, device=X.device)
        with _C.device_ctx(X):
            _C.check(lib.ltr_linear_grad_f32(_C.ptr(X), _C.ptr(g), _C.ptr(nn) if ctx.has_n else None,
                                             B, L, F, _C.ptr(out), _C.ptr(ws), ws_bytes, _C.stream_of(X)))
        gx = None
        if ctx.xs_grad:
            gm = g
            if ctx.has_n:                                       # padded documents were not scored
                gm = g * (torch.arange(L, device=g.device)[None, :] < nn[:, None])
            gx = (gm.unsqueeze(-1) * Wsaved[:ctx.Fw].reshape(1, 1, ctx.Fw)).reshape(ctx.xs_shape)
        return (gx, out[:ctx.Fw].reshape(ctx.w_shape), out[F:] if ctx.has_bias else None, None)


class LazyScores(torch.Tensor):
    """``Linear(F, 1)(xs)`` that has not been computed yet.

    What :class:`LinearScorer` returns in a training step: a (B, L, 1) fp32 tensor -- shape, dtype and device answer
    without any work -- that remembers ``(xs, weight, bias)``.  The loss modules of this package recognise it and run
    the features-once fused kernel (scores, loss, gradient and the per-query weight-gradient rows in ONE pass over the
    feature tile; ``.backward()`` is then the cross-query reduction straight into ``weight.grad`` / ``bias.grad``): the
    unchanged user script ``loss_fn(model(xs), ys, n).mean().backward()`` (examples/01-basic-usage.py:66-75,
    docs/source/getting-started.rst:86-101) reaches the kernel that ``FusedLinearLoss`` reaches.  Anybody else who
    touches the tensor -- a metric, ``.detach()``, arithmetic, ``print`` -- gets the real scores: they are computed by
    the streaming scorer kernel on first use (autograd-connected to the layer's parameters) and kept."""

    @staticmethod
    def __new__(cls, xs, weight, bias, n):
        r = torch.Tensor._make_wrapper_subclass(cls, (xs.shape[0], xs.shape[1], 1), dtype=torch.float32, device=xs.device)
        r._xs, r._weight, r._bias, r._n = xs, weight, bias, n
        r._real = None
        r._versions = (weight._version, -1 if bias is None else bias._version)
        return r

    def materialize(self):
        """The real (B, L, 1) scores (computed once, autograd-connected to the layer's parameters)."""
        if self._real is None:
            self._check_versions()
            self._real = _LinearScoreFunction.apply(self._xs, self._weight, self._bias, self._n)
        return self._real

    def _check_versions(self):
        if self._versions != (self._weight._version, -1 if self._bias is None else self._bias._version):
            raise RuntimeError("the scorer's parameters were modified in place between model(xs) and the first use of its "
                               "scores; use the scores (or call .materialize()) before the optimizer step")

    def fused_loss(self, relevance, n, kind, sigma):
        """loss[b] of the pairwise loss `kind` on these scores, or None when the scores have to be computed anyway."""
        if self._real is not None or kind == _LISTWISE_SOFTMAX:
            return None
        xs = self._xs
        B, L, F = xs.shape
        if _prefer_pieces(kind, B, L, (F + 3) & ~3 if not xs.is_contiguous() else F):
            return None
        self._check_versions()
        return _LinearLossFunction.apply(xs, self._weight, self._bias, relevance, n, kind, sigma, False)

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func in _LAZY_METADATA:
            if func == _REQUIRES_GRAD_GET:
                a = args[0]
                return bool(a._weight.requires_grad or (a._bias is not None and a._bias.requires_grad))
            with torch._C.DisableTorchFunctionSubclass():
                return func(*args, **kwargs)

        def real(a):
            if isinstance(a, LazyScores):
                return a.materialize()
            if isinstance(a, (list, tuple)):
                return type(a)(real(v) for v in a)
            return a
        with torch._C.DisableTorchFunctionSubclass():
            return func(*[real(a) for a in args], **{k: real(v) for k, v in kwargs.items()})


    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        # (only reached by calls that bypass the Python API -- every torch function and Tensor method goes through
        # __torch_function__ above, where autograd still sees the real scores)
        def real(a):
            if isinstance(a, LazyScores):
                return a.materialize()
            if isinstance(a, (list, tuple)):
                return type(a)(real(v) for v in a)
            return a
        return func(*[real(a) for a in args], **{k: real(v) for k, v in (kwargs or {}).items()})


_LISTWISE_SOFTMAX = 100                  # (_autograd.LISTWISE_SOFTMAX: not a pairwise kind, no fused kernel)
_REQUIRES_GRAD_GET = torch.Tensor.requires_grad.__get__
_LAZY_METADATA = {
    torch.Tensor.shape.__get__, torch.Tensor.dtype.__get__, torch.Tensor.device.__get__, torch.Tensor.is_cuda.__get__,
    torch.Tensor.ndim.__get__, torch.Tensor.layout.__get__, torch.Tensor.size, torch.Tensor.dim, torch.Tensor.numel,
    torch.Tensor.nelement, torch.Tensor.ndimension, torch.Tensor.is_floating_point, torch.Tensor.is_complex,
    torch.Tensor.get_device, _REQUIRES_GRAD_GET,
}


class LinearScorer(torch.nn.Module):
    """``torch.nn.Linear(in_features, 1)`` for (B, L, F) feature batches, state_dict-compatible
    with it, computed by HBM-streaming kernels instead of rocBLAS' one-column GEMM:
    ``loss_fn(scorer(xs), ys, n)`` is the reference's user code unchanged.  ``scorer(xs, n)`` also
    skips the padded documents (score 0).  While gradients are enabled the layer returns :class:`LazyScores`: a loss
    module of this package then runs scores + loss + weight gradient as ONE kernel over the features (what
    ``FusedLinearLoss`` does), anything else that touches the scores computes them on the spot.  The feature batch gets
    no gradient (it is data); an input that requires one (the layer sits behind others) gets
    ``grad_scores (x) weight``.

    ONE difference from ``nn.Linear`` follows from the laziness: scores that are first USED after ``optimizer.step()`` --
    ``scores = model(xs); loss_fn(scores, ys, n).mean().backward(); optimizer.step(); ndcg(scores, ys, n)`` -- would have to
    be computed with weights that no longer exist; that first use raises a RuntimeError instead of returning post-step
    scores.  Use the scores before the step (any use computes and keeps them), call ``scores.materialize()``, or build the
    layer with ``lazy=False`` (``use_linear_scorer`` keeps the default) -- then every call computes its scores at once and
    only the fused ``FusedLinearLoss`` / ``linear_loss_step`` entry points take the one-pass kernel."""

    def __init__(self, in_features, bias=True, lazy=True):
        super().__init__()
        self.in_features = in_features
        self.lazy = lazy
        self.weight = torch.nn.Parameter(torch.empty(1, in_features))
        self.bias = torch.nn.Parameter(torch.empty(1)) if bias else None
        torch.nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            bound = 1.0 / math.sqrt(in_features)
            torch.nn.init.uniform_(self.bias, -bound, bound)
            self.weight._ltr_scorer_bias = self.bias          # (pytorchltr_amd.optim.SGD pairs them up by this)

    def forward(self, xs, n=None):
        if not (torch.is_tensor(xs) and xs.dim() == 3 and xs.is_cuda and xs.dtype is torch.float32
                and xs.shape[-1] == self.in_features and not torch.is_autocast_enabled()):
            # not an fp32 (B, L, F) feature batch on the device (a value head, a gate, a 2-D input, half precision under
            # autocast ...): the plain layer, with nn.Linear's dtype rules and errors
            return torch.nn.functional.linear(xs, self.weight, self.bias)
        if (self.lazy and torch.is_grad_enabled() and not xs.requires_grad and (xs.is_contiguous() or _padded_rows(xs) is not None)
                and (self.weight.requires_grad or (self.bias is not None and self.bias.requires_grad))):
            return LazyScores(xs, self.weight, self.bias, n)
        return _LinearScoreFunction.apply(xs, self.weight, self.bias, n)


def use_linear_scorer(model, predicate=None):
    """The one-liner for an existing training script: every ``torch.nn.Linear(F, 1)`` inside `model`
    (or `model` itself) is replaced by a :class:`LinearScorer` that SHARES its parameters -- same
    ``state_dict`` keys, same Parameter objects, so optimisers, checkpoints and ``model.parameters()``
    are unaffected -- and `loss_fn(model(xs), ys, n).mean().backward()` stops paying rocBLAS for a
    one-column GEMM (304 us -> 46 us per step at the C2 shape; reference user code:
    examples/01-basic-usage.py:36,66-75).  Returns the model (a new module when `model` itself was the
    Linear layer).  Only applied where the layer's input is the (B, L, F) feature batch, i.e. where the
    user wrote ``Linear(F, 1)`` as the scorer: `predicate(name, module) -> bool` restricts the replacement (default: every
    ``Linear(*, 1)``); a converted layer that is fed anything but a 3-D device tensor falls back to ``F.linear``."""
    def convert(lin):
        sc = LinearScorer(lin.in_features, bias=lin.bias is not None)
        sc.weight = lin.weight                      # the same Parameter objects
        if lin.bias is not None:
            sc.bias = lin.bias
            sc.weight._ltr_scorer_bias = sc.bias
        return sc

    def is_scorer(m):
        return isinstance(m, torch.nn.Linear) and m.out_features == 1

    if is_scorer(model):
        return convert(model) if (predicate is None or predicate("", model)) else model
    for name, child in list(model.named_children()):
        if is_scorer(child):
            if predicate is None or predicate(name, child):
                setattr(model, name, convert(child))
        else:
            use_linear_scorer(child, predicate)
    return model


# ---------------------------------------------------------------------------------------------
# ReLU MLP scorer (the network of the reference's guide) fused with the loss -- SURVEY.md 8 f-2
# ---------------------------------------------------------------------------------------------
MLP_MAX_LIST_LEN = 128           # feature rows wider than 144 floats
MLP_MAX_LIST_LEN_NARROW = 256    # F <= 144
MLP_NARROW_FEATURES = 144
MLP_MAX_FEATURES = 224
MLP_MAX_HIDDEN = (64, 16)


def mlp_max_list_len(F):
    """Longest list the fused MLP kernels take at F features: 256 for F <= 144 (the tile kernel),
    128 for wider rows (ltr_mlp_max_list_len)."""
    return MLP_MAX_LIST_LEN_NARROW if F <= MLP_NARROW_FEATURES else MLP_MAX_LIST_LEN


def mlp_supported(L, F, H1, H2):
    """Shapes the fused MLP kernel takes (see include/ltr_hip.h: ltr_mlp_pairwise_f32)."""
    return (0 < F <= MLP_MAX_FEATURES and F % 4 == 0 and 0 < L <= mlp_max_list_len(F)
            and 0 < H1 <= MLP_MAX_HIDDEN[0] and 0 < H2 <= MLP_MAX_HIDDEN[1])


def _pad_features(xs, w1):
    """Feature counts that are not a multiple of 4 (MQ2007: 46, Yahoo: 699): zero columns are
    appended to the features and to W1 -- same scores, and the extra dW1 columns are dropped.  Costs
    one copy of the batch; pad the stored split once to avoid it."""
    extra = (-xs.shape[-1]) % 4
    if extra == 0:
        return xs, w1, 0
    return (torch.nn.functional.pad(xs, (0, extra)), torch.nn.functional.pad(w1, (0, extra)), extra)


def _flat_params(params, F):
    # (fp32 contiguous tensors -- nn.Linear parameters -- go to the C ABI as they are)
    W1, b1, W2, b2, W3, b3 = [t if (t.dtype is torch.float32 and t.is_contiguous())
                              else t.detach().float().contiguous() for t in params]
    H1, H2 = W1.shape[0], W2.shape[0]
    if W1.shape != (H1, F) or b1.numel() != H1 or W2.shape != (H2, H1) or b2.numel() != H2 \
            or W3.numel() != H2 or b3.numel() != 1:
        raise ValueError("parameters must be those of Linear(F,H1), Linear(H1,H2), Linear(H2,1)")
    return (W1, b1, W2, b2, W3, b3), H1, H2


def _split_grads(flat, F, H1, H2):
    """The six gradient tensors as views of the flat buffer [dW1 | db1 | dW2 | db2 | dW3 | db3]."""
    w1, b1, w2, b2, w3, b3 = flat.split((H1 * F, H1, H2 * H1, H2, H2, 1))
    return (w1.view(H1, F), b1, w2.view(H2, H1), b2, w3.view(1, H2), b3)


_mlp_sizes = {}          # (B, F, H1, H2) -> (parameter count, workspace bytes)
_mlp_workspaces = {}     # (device index, stream, bytes) -> scratch tensor (consumed in stream order)


def _mlp_sizes_for(B, F, H1, H2):
    key = (B, F, H1, H2)
    v = _mlp_sizes.get(key)
    if v is None:
        lib = _C.lib()
        v = _mlp_sizes[key] = (int(lib.ltr_mlp_param_count(F, H1, H2)),
                               int(lib.ltr_mlp_workspace_bytes(B, F, H1, H2)))
    return v


def _mlp_workspace(dev, stream, ws_bytes):
    """The per-workgroup partial vectors (15 MB at the guide's network) live only between the two
    launches of one call, which run in stream order: one buffer per (device, stream) is reused."""
    if torch.cuda.is_current_stream_capturing():
        # a buffer allocated under capture belongs to the graph's pool: never cache it
        return torch.empty(max(ws_bytes, 4) // 4, dtype=torch.float32, device=dev)
    key = (dev.index, stream, ws_bytes)
    ws = _mlp_workspaces.get(key)
    if ws is None:
        if len(_mlp_workspaces) > 16:
            _mlp_workspaces.clear()
        ws = _mlp_workspaces[key] = torch.empty(max(ws_bytes, 4) // 4, dtype=torch.float32, device=dev)
    return ws


def mlp_loss_step(xs, params, relevance, n, loss="hinge", grad_out=None, return_scores=False,
                  return_loss_sum=False, out=None):
    """One fused forward+backward step of ``loss_fn(mlp(xs), relevance, n)`` without autograd.

    Args:
        xs: (B, L, F) float32 features on the device.
        params: ``(W1, b1, W2, b2, W3, b3)`` of ``Linear(F,H1) / ReLU / Linear(H1,H2) / ReLU /
            Linear(H2,1)`` in torch layout.
        grad_out: (B,) weights of the per-query losses; None = 1/B (the ``.mean()`` of the guide's
            training loop, docs/source/getting-started.rst:95).
        out: optional preallocated flat gradient buffer of ``ltr_mlp_param_count`` floats.

    Returns:
        ``(loss[B], grads)`` with ``grads = (dW1, db1, dW2, db2, dW3, db3)`` views of one flat
        buffer (available as ``grads[0].base`` / the ``out`` argument), then optionally the scores
        (valid for documents < n only) and ``loss_sum`` (1,).
    """
    kind, sigma = _resolve_loss(loss)
    X = _prepare_features(xs)
    B, L, F = X.shape
    flat_params, H1, H2 = _flat_params(params, F)
    if not mlp_supported(L, F, H1, H2):
        raise ValueError("fused MLP kernel takes L <= %d (F <= 144: %d), F <= %d with F %% 4 == 0, "
                         "hidden <= %s; got L=%d F=%d hidden=(%d, %d)"
                         % (MLP_MAX_LIST_LEN, MLP_MAX_LIST_LEN_NARROW, MLP_MAX_FEATURES,
                            MLP_MAX_HIDDEN, L, F, H1, H2))
    dev = X.device
    r, nn = _labels_and_n(relevance, n, B, L, dev)
    lib = _C.lib()
    P, ws_bytes = _mlp_sizes_for(B, F, H1, H2)
    lossv = torch.empty(B, dtype=torch.float32, device=dev)
    flat = out if out is not None else torch.empty(P, dtype=torch.float32, device=dev)
    if flat.numel() != P or flat.dtype != torch.float32 or not flat.is_contiguous():
        raise ValueError("out must be a contiguous float32 tensor of %d elements" % P)
    # (loss_sum is written, not accumulated, by the reduction launch; B == 0 launches nothing)
    lsum = None
    if return_loss_sum:
        lsum = (torch.empty if B > 0 else torch.zeros)(1, dtype=torch.float32, device=dev)
    scores = torch.zeros(B, L, dtype=torch.float32, device=dev) if return_scores else None
    go = None if grad_out is None else grad_out.reshape(B).float().contiguous()
    with _C.device_ctx(X):
        st = _C.stream_of(X)
        ws = _mlp_workspace(dev, st, ws_bytes)
        rc = lib.ltr_mlp_pairwise_f32(
            kind, float(sigma), X.data_ptr(), flat_params[0].data_ptr(), flat_params[1].data_ptr(),
            flat_params[2].data_ptr(), flat_params[3].data_ptr(), flat_params[4].data_ptr(),
            flat_params[5].data_ptr(), r.data_ptr(), _LABEL_CODE[r.dtype], nn.data_ptr(),
            None if go is None else go.data_ptr(), B, L, F, H1, H2, lossv.data_ptr(),
            None if scores is None else scores.data_ptr(), flat.data_ptr(),
            None if lsum is None else lsum.data_ptr(), ws.data_ptr(), ws_bytes, st)
        if rc != 0:
            _C.check(rc)
    res = (lossv, _split_grads(flat, F, H1, H2))
    if return_scores:
        res = res + (scores,)
    if return_loss_sum:
        res = res + (lsum,)
    return res


def mlp_scores(xs, params, n=None):
    """``model(xs)`` of the guide's network as one fused kernel (no autograd): (B, L) float32
    scores for documents < n[b] and 0 for the padded ones; n=None scores every document.
    Shapes outside the kernel's limits raise ValueError (see :func:`mlp_supported`)."""
    X = _prepare_features(xs)
    B, L, F = X.shape
    flat_params, H1, H2 = _flat_params(params, F)
    if not mlp_supported(L, F, H1, H2):
        raise ValueError("fused MLP kernel takes L <= %d (F <= 144: %d), F <= %d with F %% 4 == 0, "
                         "hidden <= %s" % (MLP_MAX_LIST_LEN, MLP_MAX_LIST_LEN_NARROW, MLP_MAX_FEATURES,
                                           MLP_MAX_HIDDEN))
    nn = (torch.full((B,), L, dtype=torch.int64, device=X.device) if n is None else prepare_n(n, B))
    scores = torch.empty(B, L, dtype=torch.float32, device=X.device)
    if B > 0:
        with _C.device_ctx(X):
            _C.check(_C.lib().ltr_mlp_scores_f32(
                _C.ptr(X), *[_C.ptr(t) for t in flat_params], _C.ptr(nn), B, L, F, H1, H2,
                _C.ptr(scores), _C.stream_of(X)))
    return scores


class _MLPLossFunction(torch.autograd.Function):
    """Reduced (mean / sum) loss of the fused MLP step; the kernel already produced the parameter
    gradients of the reduced loss, backward only scales them by the incoming scalar."""

    @staticmethod
    def forward(ctx, xs, relevance, n, kind_sigma, mean, *params):
        B = xs.shape[0]
        go = None if mean else torch.ones(B, dtype=torch.float32, device=xs.device)
        loss = _KindProxy(*kind_sigma)
        xs, w1, extra = _pad_features(xs, params[0].detach())
        lossv, grads, lsum = mlp_loss_step(xs, (w1,) + tuple(params[1:]), relevance, n, loss=loss,
                                           grad_out=go, return_loss_sum=True)
        ctx.save_for_backward(grads[0]._base)       # the flat buffer: one scale in backward
        ctx.dims = (xs.shape[2], params[0].shape[0], params[2].shape[0])
        ctx.extra = extra
        ctx.shapes = [p.shape for p in params]
        ctx.per_query = lossv
        total = lsum.reshape(())
        out = total / B if (mean and B > 0) else total
        ctx.mark_non_differentiable(lossv)
        return out, lossv

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_total, grad_unused):
        (flat,) = ctx.saved_tensors
        parts = list(_split_grads(flat * grad_total, *ctx.dims))
        if ctx.extra:
            parts[0] = parts[0][:, :ctx.dims[0] - ctx.extra]
        scaled = tuple(g if g.shape == s else g.reshape(s) for g, s in zip(parts, ctx.shapes))
        return (None, None, None, None, None) + scaled


class _KindProxy:
    def __init__(self, kind, sigma):
        self._kind = kind
        self.sigma = sigma


class FusedMLPLoss(torch.nn.Module):
    """The guide's scorer and loss as one module: ``l1``/``l2``/``l3`` are ordinary
    ``torch.nn.Linear`` layers (state_dict-compatible with the ``Model`` class of
    docs/source/getting-started.rst:40-50), and ``forward(xs, relevance, n)`` returns the
    *reduced* loss ``loss_fn(model(xs), relevance, n).mean()`` (or ``.sum()``) whose backward
    fills the six parameter gradients -- computed by one fused MFMA kernel.

    Feature counts that are not a multiple of 4 are zero-padded on the fly.  Shapes the kernel
    does not take (lists longer than 256 -- 128 beyond 144 features --, more than 224 features, ...) run as the unfused
    composition: rocBLAS layers + the HIP loss kernel.  ``score(xs)`` evaluates the
    network alone (for the metrics).
    """

    def __init__(self, in_features, loss="hinge", hidden=(50, 10), reduction="mean"):
        super().__init__()
        if reduction not in ("mean", "sum"):
            raise ValueError("reduction must be 'mean' or 'sum'")
        self.in_features = in_features
        self.reduction = reduction
        self.kind, self.sigma = _resolve_loss(loss)
        self.l1 = torch.nn.Linear(in_features, hidden[0])
        self.l2 = torch.nn.Linear(hidden[0], hidden[1])
        self.l3 = torch.nn.Linear(hidden[1], 1)
        self.last_losses = None

    def score(self, xs, n=None):
        """``model(xs)``: (B, L, 1) scores.  Under ``torch.no_grad()`` (evaluation) and for shapes
        the fused kernel takes it is one launch (padded documents, when ``n`` is given, score 0);
        otherwise the three ``nn.Linear`` layers, with autograd."""
        if (not torch.is_grad_enabled() and xs.dim() == 3 and xs.is_cuda
                and mlp_supported(xs.shape[1], (xs.shape[2] + 3) & ~3, self.l1.out_features,
                                  self.l2.out_features)):
            xp, w1, _ = _pad_features(xs, self.l1.weight.detach())
            return mlp_scores(xp, (w1,) + self._params()[1:], n).unsqueeze(-1)
        o1 = torch.nn.functional.relu(self.l1(xs))
        o2 = torch.nn.functional.relu(self.l2(o1))
        return self.l3(o2)

    def _params(self):
        return (self.l1.weight, self.l1.bias, self.l2.weight, self.l2.bias, self.l3.weight,
                self.l3.bias)

    def forward(self, xs, relevance, n):
        _C.require_device(xs, "xs")
        B, L, F = xs.shape
        if mlp_supported(L, (F + 3) & ~3, self.l1.out_features, self.l2.out_features):
            total, per_query = _MLPLossFunction.apply(
                xs, relevance, n, (self.kind, self.sigma), self.reduction == "mean", *self._params())
            self.last_losses = per_query
            return total
        from ._autograd import PairwiseLossFunction
        per_query = PairwiseLossFunction.apply(self.score(xs), relevance, n, self.kind, self.sigma)
        self.last_losses = per_query.detach()
        return per_query.mean() if self.reduction == "mean" else per_query.sum()
