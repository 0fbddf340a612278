"""``pytorchltr_amd.optim.SGD``: ``torch.optim.SGD`` (same constructor, same results) that turns the reference's training
loop body

    loss = loss_fn(model(xs), ys, n).mean()         # model = use_linear_scorer(nn.Linear(F, 1))
    optimizer.zero_grad()
    loss.backward()
    optimizer.step()

(examples/01-basic-usage.py:66-75) into ONE kernel launch per step for the scorer's parameters: only the import line of
the optimizer changes.  How:

* ``loss_fn(model(xs), ys, n)`` already is one fused launch (scores + loss + per-query weight-gradient rows:
  ``fused.LazyScores``).  With this optimizer the launch is the LAZY step of the C ABI
  (``ltr_linear_sgd_lazy_step_dp_f32``): it also applies the PREVIOUS step's update -- its first workgroups sum the previous
  batch's rows, update ``weight`` / ``bias`` in place and hand the new weights to every workgroup before the dot products.
* ``loss.backward()`` launches nothing for the scorer: ``weight.grad`` / ``bias.grad`` become :class:`LazyGrad` tensors --
  shape, dtype and device answer at once, the VALUES are computed (one reduction launch) only if somebody reads them.
* ``optimizer.step()`` launches nothing either: it records "this batch's update is pending" with the group's ``lr`` and
  the upstream gradient autograd handed over (a device scalar: ``.mean()``'s 1 / B is read by the kernel, not by the host).
* FLUSH ON READ: the scorer's parameters become :class:`_LazyParameter` (``nn.Parameter`` subclass; the objects stay the
  same, so ``model.parameters()``, ``state_dict()`` and checkpoints are unaffected).  Anything that reads them through
  torch -- an evaluation pass, ``state_dict()``, ``torch.save``, ``print``, another optimizer -- first applies the pending
  update (one small launch), so nobody ever sees weights ``optimizer.step()`` has not reached yet.

Everything that is not plain SGD on a ``LinearScorer`` (momentum, weight decay, nesterov, maximize, other parameters, a
non-uniform upstream gradient, gradient accumulation, shapes the register-tile kernel does not take) runs as
``torch.optim.SGD`` does, on gradients that are materialised on the spot: same numbers, more launches.
"""
import weakref

import torch

from . import _C

__all__ = ["SGD", "LazyGrad"]


class _PendingGrad:
    """The gradient of ONE backward pass through a lazy fused forward: where its rows are, what it was scaled with, and --
    once somebody asked -- its values."""
    __slots__ = ("st", "gen", "go", "loss", "shape", "real", "consumed", "__weakref__")

    def __init__(self, st, gen, go, loss, shape):
        self.st, self.gen, self.go, self.loss, self.shape = st, gen, go, loss, shape     # shape = (kind, B, L, F)
        self.real = None            # (dW (1, F), db (1,)) once materialised
        self.consumed = False       # optimizer.step() has taken it (the update is pending or applied)

    def materialize(self):
        if self.real is None:
            self.real = self.st.gradient_of(self)
        return self.real


class LazyGrad(torch.Tensor):
    """``weight.grad`` / ``bias.grad`` of a ``LinearScorer`` trained with :class:`SGD` that has not been computed: the
    per-query gradient rows are on the device, their sum over the queries is taken by the launch that applies the update
    (the next step's) -- or, if anybody touches this tensor first, by a reduction launch of its own, after which it
    behaves like the plain fp32 tensor it stands for."""

    @staticmethod
    def __new__(cls, pg, which, shape, device):
        r = torch.Tensor._make_wrapper_subclass(cls, shape, dtype=torch.float32, device=device)
        r._pg, r._which = pg, which
        return r

    def materialize(self):
        return self._pg.materialize()[self._which]

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func in _GRAD_METADATA:
            with torch._C.DisableTorchFunctionSubclass():
                return func(*args, **kwargs)

        def real(a):
            if isinstance(a, LazyGrad):
                return a.materialize()
            if isinstance(a, (list, tuple)):
                return type(a)(real(v) for v in a)
            return a
        with torch._C.DisableTorchFunctionSubclass():
            return func(*[real(a) for a in args], **{k: real(v) for k, v in kwargs.items()})

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        # (autograd's AccumulateGrad node works below the Python API: it detaches / aliases / clones the gradient it is
        # handed before it stores it -- those stay lazy; anything else gets the values)
        if func in _GRAD_ALIASING and isinstance(args[0], LazyGrad) and args[0]._pg.real is None:
            a = args[0]
            return LazyGrad(a._pg, a._which, a.shape, a.device)

        def real(a):
            if isinstance(a, LazyGrad):
                return a.materialize()
            if isinstance(a, (list, tuple)):
                return type(a)(real(v) for v in a)
            return a
        return func(*[real(a) for a in args], **{k: real(v) for k, v in (kwargs or {}).items()})


_GRAD_METADATA = {
    torch.Tensor.shape.__get__, torch.Tensor.dtype.__get__, torch.Tensor.device.__get__, torch.Tensor.is_cuda.__get__,
    torch.Tensor.ndim.__get__, torch.Tensor.layout.__get__, torch.Tensor.size, torch.Tensor.dim, torch.Tensor.numel,
    torch.Tensor.nelement, torch.Tensor.ndimension, torch.Tensor.is_floating_point, torch.Tensor.is_complex,
    torch.Tensor.get_device, torch.Tensor.requires_grad.__get__, torch.Tensor.is_leaf.__get__, torch.Tensor.grad_fn.__get__,
    torch.Tensor.is_sparse.__get__,
}
_GRAD_ALIASING = {torch.ops.aten.detach.default, torch.ops.aten.alias.default, torch.ops.aten.clone.default}


# what may be asked of a lazily updated parameter without applying the pending update first: nothing that shows values
_PARAM_NO_FLUSH = {
    torch.Tensor.shape.__get__, torch.Tensor.dtype.__get__, torch.Tensor.device.__get__, torch.Tensor.is_cuda.__get__,
    torch.Tensor.ndim.__get__, torch.Tensor.layout.__get__, torch.Tensor.size, torch.Tensor.dim, torch.Tensor.numel,
    torch.Tensor.nelement, torch.Tensor.ndimension, torch.Tensor.is_floating_point, torch.Tensor.is_complex,
    torch.Tensor.get_device, torch.Tensor.requires_grad.__get__, torch.Tensor.requires_grad.__set__, torch.Tensor.requires_grad_,
    torch.Tensor.is_leaf.__get__, torch.Tensor.grad_fn.__get__, torch.Tensor.grad.__get__, torch.Tensor.grad.__set__,
    torch.Tensor.grad.__delete__, torch.Tensor.is_contiguous, torch.Tensor.stride, torch.Tensor.storage_offset,
    torch.Tensor.is_sparse.__get__, torch.Tensor.element_size, torch.Tensor.register_hook, torch.Tensor._version.__get__,
    torch.Tensor.output_nr.__get__, torch.Tensor.names.__get__, torch.Tensor.is_meta.__get__, torch.Tensor.is_quantized.__get__,
    torch.Tensor.__hash__,
}


class _LazyParameter(torch.nn.Parameter):
    """An ``nn.Parameter`` whose SGD update may still be pending (see :class:`SGD`): every torch function that could show
    its values applies the update first.  Objects become this class in place (``p.__class__ = _LazyParameter``), so
    modules, state_dict keys and other references are unaffected."""

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        if func not in _PARAM_NO_FLUSH:
            for a in args:
                st = getattr(a, "_ltr_lazy", None) if isinstance(a, _LazyParameter) else None
                if st is not None and st.pending is not None:
                    st.flush()
                elif isinstance(a, (list, tuple)):
                    for v in a:
                        st = getattr(v, "_ltr_lazy", None) if isinstance(v, _LazyParameter) else None
                        if st is not None and st.pending is not None:
                            st.flush()
        return torch._C._disabled_torch_function_impl(func, types, args, kwargs or {})

    def __repr__(self):
        st = getattr(self, "_ltr_lazy", None)
        if st is not None and st.pending is not None:
            st.flush()
        with torch._C.DisableTorchFunctionSubclass():
            return "Parameter containing:\n" + torch.Tensor.__repr__(self.data)


class _Pending:
    __slots__ = ("pg", "lr")

    def __init__(self, pg, lr):
        self.pg, self.lr = pg, lr


class _LazyLinear:
    """The lazily updated (weight, bias) of one ``LinearScorer``: the persistent workspace its fused launches write their
    gradient rows to, the update ``optimizer.step()`` recorded and the next launch applies, and who still needs the rows."""

    def __init__(self, weight, bias, group):
        self.weight, self.bias, self.group = weight, bias, group
        with torch._C.DisableTorchFunctionSubclass():
            self.w_raw, self.b_raw = weight.data, bias.data           # plain aliases: no flush hook, same storage
        self.ws = None
        self.ws_bytes = 0
        self._ws_need = {}
        self.bucket = torch.zeros(weight.numel() + 2, dtype=torch.float32, device=weight.device)
        self.bucket_gen = -1        # whose gradient the bucket holds
        self.gen = 0                # bumped by every lazy forward: the rows in `ws` belong to generation `gen`
        self.node = None            # weakref to the autograd node of a lazy forward whose backward has not run yet
        self.grad = None            # weakref to the _PendingGrad of the rows in `ws`
        self.pending = None         # _Pending: optimizer.step() recorded it, no launch has applied it yet
        self.keep = None            # tensors the last launch still reads (stream order)
        self.enabled = True

    # ---- conditions ----
    def plain_sgd(self):
        g = self.group
        return (g.get("momentum", 0) == 0 and g.get("weight_decay", 0) == 0 and not g.get("nesterov", False)
                and not g.get("maximize", False) and not g.get("differentiable", False) and g.get("dampening", 0) == 0)

    def storage_ok(self):
        w, b = self.weight, self.bias
        with torch._C.DisableTorchFunctionSubclass():
            return (w.data_ptr() == self.w_raw.data_ptr() and b.data_ptr() == self.b_raw.data_ptr()
                    and w.dtype is torch.float32 and b.dtype is torch.float32 and w.is_contiguous())

    def busy(self):
        """The rows of the last lazy forward are still owed to a backward pass that has not run."""
        return self.node is not None and self.node() is not None

    # ---- the launches ----
    def _ensure_ws(self, B, L, F):
        need = self._ws_need.get((B, L, F))
        if need is None:
            need = self._ws_need[(B, L, F)] = max(int(_C.lib().ltr_linear_workspace_bytes(B, L, F)), 4)
        if need > self.ws_bytes:
            self.settle()                                   # nothing may still live in the old buffer
            self.ws = torch.empty(need // 4 + 16, dtype=torch.float32, device=self.w_raw.device)
            self.ws_bytes = need

    def settle(self):
        """Before the rows in `ws` are overwritten or go away: apply a pending update, compute a gradient that is still
        wanted."""
        if self.pending is not None:
            self.flush()
        pg = self.grad() if self.grad is not None else None
        if pg is not None and pg.real is None and not pg.consumed and pg.gen == self.gen:
            pg.materialize()

    def forward(self, ctx, X, r, rcode, nn, kind, sigma):
        """The lazy fused launch for the batch (X, r, nn); returns the per-query losses, or None when this call must take
        the ordinary path."""
        B, L, F = X.shape
        if (not self.enabled or B == 0 or F != self.w_raw.numel() or not self.plain_sgd() or self.busy()
                or not self.storage_ok() or X.data_ptr() % 16 != 0):
            return None
        lib = _C.lib()
        pend = self.pending
        if pend is not None and (pend.pg.shape[0] != kind or pend.pg.shape[2:] != (L, F)):
            self.flush()                                    # (the pending rows' layout follows from ITS kind / L / F)
            pend = None
        self._ensure_ws(max(B, pend.pg.shape[1] if pend is not None else 0), L, F)
        pend = self.pending
        pg_old = self.grad() if self.grad is not None else None
        if pg_old is not None and pg_old.real is None and not pg_old.consumed and pg_old.gen == self.gen:
            pg_old.materialize()                            # somebody still holds the last gradient: its rows go now
        loss = torch.empty(B, dtype=torch.float32, device=X.device)
        with _C.device_ctx(X):
            rc = lib.ltr_linear_sgd_lazy_step_dp_f32(
                kind, float(sigma), X.data_ptr(), self.w_raw.data_ptr(), self.b_raw.data_ptr(), r.data_ptr(), rcode,
                nn.data_ptr(), B, L, F, float(pend.lr) if pend is not None else 0.0, loss.data_ptr(), self.bucket.data_ptr(),
                self.ws.data_ptr(), self.ws_bytes, pend.pg.shape[1] if pend is not None else 0, 0.0,
                pend.pg.go.data_ptr() if pend is not None else None, pend.pg.go.stride(0) if pend is not None else 0, None,
                _C.stream_of(X))
            if rc != 0:
                _C.check(rc)
        if pend is not None:
            self.keep = (pend.pg.go, pend.pg.loss)          # read by the launch just enqueued
            self.bucket_gen = pend.pg.gen
            self.pending = None
        self.gen += 1
        self.grad = None
        self.node = weakref.ref(ctx)
        ctx.lazy = (self, self.gen, (kind, B, L, F))
        return loss

    def flush(self):
        """Applies the pending update now (one small launch): the reducers of the lazy launch on their own."""
        pend = self.pending
        if pend is None:
            return
        self.pending = None
        kind, B, L, F = pend.pg.shape
        with torch.cuda.device(self.w_raw.device):
            rc = _C.lib().ltr_linear_sgd_flush_dp_f32(
                kind, self.w_raw.data_ptr(), self.b_raw.data_ptr(), B, L, F, float(pend.lr), 0.0, pend.pg.go.data_ptr(),
                pend.pg.go.stride(0), pend.pg.loss.data_ptr(), self.bucket.data_ptr(), self.ws.data_ptr(), None,
                torch.cuda.current_stream(self.w_raw.device).cuda_stream)
            if rc != 0:
                _C.check(rc)
        self.keep = (pend.pg.go, pend.pg.loss)
        self.bucket_gen = pend.pg.gen

    def gradient_of(self, pg):
        """(dW (1, F), db (1,)) of the backward pass `pg` -- asked for by somebody who touched a LazyGrad."""
        kind, B, L, F = pg.shape
        if pg.consumed:
            # optimizer.step() has been here: the launch that applies the update also leaves the gradient in the bucket
            if self.pending is not None and self.pending.pg is pg:
                self.flush()
            if self.bucket_gen != pg.gen:
                raise RuntimeError("this gradient is no longer available: later training steps have reused its buffers "
                                   "(read .grad before the next but one step, or clone it)")
            out = self.bucket[:F + 1].clone()
        else:
            if pg.gen != self.gen:
                raise RuntimeError("this gradient is no longer available: a later forward pass has reused its buffers")
            out = torch.empty(F + 2, dtype=torch.float32, device=self.w_raw.device)
            with torch.cuda.device(self.w_raw.device):
                rc = _C.lib().ltr_linear_lazy_rows_reduce_f32(
                    kind, B, L, F, 0.0, pg.go.data_ptr(), pg.go.stride(0), pg.loss.data_ptr(), out.data_ptr(), self.ws.data_ptr(), 1,
                    torch.cuda.current_stream(self.w_raw.device).cuda_stream)
                if rc != 0:
                    _C.check(rc)
        with torch._C.DisableTorchFunctionSubclass():
            return {"w": out[:F].reshape(self.weight.shape), "b": out[F:F + 1].reshape(self.bias.shape)}

    def backward(self, ctx, grad_loss):
        """Called by the fused forward's backward: lazy gradients for (weight, bias), or None when the rows are gone or
        the upstream gradient is not one broadcast scalar (the caller then recomputes the ordinary way)."""
        _, gen, shape = ctx.lazy
        self.node = None
        go = grad_loss
        # (the upstream gradient where autograd left it: `.sum()`'s expanded scalar -- stride 0 --, `.mean()`'s vector of 1 / B or
        # any per-query weights -- stride 1; the launch that applies the update reads it there)
        if gen != self.gen or not (go.dtype is torch.float32 and go.dim() == 1 and go.stride(0) in (0, 1) and go.is_cuda):
            return None
        pg = _PendingGrad(self, gen, go, ctx.lazy_loss, shape)
        self.grad = weakref.ref(pg)
        dev = self.w_raw.device
        with torch._C.DisableTorchFunctionSubclass():
            return (LazyGrad(pg, "w", self.weight.shape, dev), LazyGrad(pg, "b", self.bias.shape, dev))

    def take(self):
        """optimizer.step(): the update of the current rows becomes pending.  True when taken."""
        with torch._C.DisableTorchFunctionSubclass():
            gw, gb = self.weight.grad, self.bias.grad
        if not (isinstance(gw, LazyGrad) and isinstance(gb, LazyGrad) and gw._pg is gb._pg):
            return False
        pg = gw._pg
        if pg.st is not self or pg.real is not None or pg.consumed or pg.gen != self.gen or not self.plain_sgd() or not self.storage_ok():
            return False
        if self.pending is not None:
            self.flush()
        pg.consumed = True
        self.pending = _Pending(pg, float(self.group["lr"]))
        return True


class SGD(torch.optim.SGD):
    """``torch.optim.SGD`` with the same constructor.  Parameters of a ``pytorchltr_amd.fused.LinearScorer`` (what
    ``use_linear_scorer`` puts in place of ``nn.Linear(F, 1)``) in a plain-SGD group (no momentum, weight decay, nesterov,
    maximize) are updated LAZILY -- see the module docstring; everything else exactly as ``torch.optim.SGD`` does."""

    def __init__(self, params, lr=1e-3, *args, **kwargs):
        super().__init__(params, lr, *args, **kwargs)
        self._lazy = []
        self._adopt()

    def _adopt(self):
        known = {id(st.weight) for st in self._lazy}
        for group in self.param_groups:
            by_id = {id(p): p for p in group["params"]}
            for p in group["params"]:
                bias = getattr(p, "_ltr_scorer_bias", None)
                if bias is None or id(p) in known or id(bias) not in by_id:
                    continue
                if not (p.is_cuda and p.dtype is torch.float32 and bias.dtype is torch.float32 and p.requires_grad
                        and bias.requires_grad and bias.numel() == 1 and p.is_contiguous()):
                    continue
                if getattr(p, "_ltr_lazy", None) is not None:
                    raise ValueError("this scorer's parameters already belong to another pytorchltr_amd.optim.SGD")
                st = _LazyLinear(p, bias, group)
                p.__class__ = _LazyParameter
                bias.__class__ = _LazyParameter
                p._ltr_lazy = st
                bias._ltr_lazy = st
                self._lazy.append(st)

    def add_param_group(self, param_group):
        super().add_param_group(param_group)
        if hasattr(self, "_lazy"):
            self._adopt()

    def flush(self):
        """Applies every pending update now (reading the parameters does this by itself)."""
        for st in self._lazy:
            st.flush()

    def _step_impl(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        taken = []
        with torch._C.DisableTorchFunctionSubclass():
            for st in self._lazy:
                if st.take():
                    taken.append((st.weight, st.weight.grad))
                    taken.append((st.bias, st.bias.grad))
                    st.weight.grad = None
                    st.bias.grad = None
            rest = any(p.grad is not None for g in self.param_groups for p in g["params"])
        if rest:
            _torch_sgd_step()(self)            # whatever was not taken lazily: torch's own SGD (hooks and all)
        elif self._has_step_hooks():
            SGD._hooks_only(self)              # (registered step hooks run although there was nothing left to do)
        with torch._C.DisableTorchFunctionSubclass():
            for p, g in taken:
                p.grad = g                     # `.grad` stays readable after the step, as with torch.optim.SGD
        return loss

    def _has_step_hooks(self):
        from torch.optim import optimizer as _o
        return bool(self._optimizer_step_pre_hooks or self._optimizer_step_post_hooks
                    or _o._global_optimizer_pre_hooks or _o._global_optimizer_post_hooks)

    def step(self, closure=None):
        """``torch.optim.SGD.step``.  (The host side of a step that has nothing left to launch is kept short: torch's wrapper
        around every optimizer's ``step`` -- profiler ranges, hook dispatch -- only runs when there is work for torch's own
        SGD or a step hook is registered.)"""
        return self._step_impl(closure)
    step.hooked = True                         # (Optimizer.__init__ must not wrap it: _step_impl goes through torch's wrapper itself)

    def zero_grad(self, set_to_none=True):
        if not set_to_none:
            return super().zero_grad(set_to_none=False)
        with torch._C.DisableTorchFunctionSubclass():
            for g in self.param_groups:
                for p in g["params"]:
                    p.grad = None

    def state_dict(self):
        self.flush()
        return super().state_dict()


def _noop_step(self, closure=None):
    return None


_torch_step_cache = []


def _torch_sgd_step():
    """torch.optim.SGD.step inside torch's own wrapper (profiler range, step hooks) -- whether or not a plain torch.optim.SGD
    has been constructed in this process yet (the wrapper is patched onto a class by its first instance)."""
    if not _torch_step_cache:
        base = torch.optim.SGD.step
        if getattr(base, "hooked", False):
            base = getattr(base, "__wrapped__", base)
        _torch_step_cache.append(torch.optim.Optimizer.profile_hook_step(base))
    return _torch_step_cache[0]


# torch's wrapper (profiler range + pre / post hooks) around a step that does nothing: for registered hooks on fully lazy steps
SGD._hooks_only = staticmethod(torch.optim.Optimizer.profile_hook_step(_noop_step))
