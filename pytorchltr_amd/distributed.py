"""Data-parallel use of the hot path: one process per GPU, queries sharded across ranks.

Every loss/metric on the path is a per-query function of (scores[b,:], relevance[b,:], n[b])
(reference: loss/pairwise_additive.py:45,84; evaluation/dcg.py:94-98) -- there is no
cross-query term, so the data path needs NO collective.  The only exchange in a training step
is the standard gradient all-reduce of the scorer's parameters plus (loss-sum, count) for
logging: F+3 floats, latency-bound.  It is issued as ONE flattened bucket so that it costs a
single RCCL launch (xGMI ring/tree choice is irrelevant below a few KB).

Backend-agnostic: "nccl" (= RCCL on ROCm) on GPUs, "gloo" in the CPU tests.
"""
import ctypes
import os

import torch
import torch.distributed as dist


def shard_bounds(total, rank, world_size):
    """Contiguous, balanced [lo, hi) chunk of `total` queries for `rank`."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size")
    base, extra = divmod(int(total), world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_batch(tensors, rank=None, world_size=None):
    """Slices every tensor of a padded batch (features, scores, relevance, n, ...) along dim 0
    to this rank's contiguous chunk of queries."""
    if rank is None:
        rank = dist.get_rank()
    if world_size is None:
        world_size = dist.get_world_size()
    total = tensors[0].shape[0]
    for t in tensors:
        if t.shape[0] != total:
            raise ValueError("all tensors must share the batch dimension")
    lo, hi = shard_bounds(total, rank, world_size)
    return tuple(t[lo:hi] for t in tensors)


def allreduce_step(grads, loss_sum, count, group=None):
    """Sums parameter gradients and (loss_sum, count) over ranks in one flattened all-reduce.

    `grads`: list of gradient tensors whose local values are d(sum of this rank's per-query
    losses)/d(param).  After the call each holds d(global mean loss)/d(param) -- i.e. what a
    single process computes for `loss_fn(...).mean().backward()` on the unsharded batch.
    Returns (global_mean_loss, global_count) as python floats' tensors on the grads' device.
    """
    dev = grads[0].device if grads else loss_sum.device
    sizes = [g.numel() for g in grads]
    flat = torch.empty(sum(sizes) + 2, dtype=torch.float32, device=dev)
    off = 0
    for g, sz in zip(grads, sizes):
        flat[off:off + sz] = g.reshape(-1).float()
        off += sz
    flat[off] = loss_sum.reshape(()).float() if torch.is_tensor(loss_sum) else float(loss_sum)
    flat[off + 1] = float(count)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    total = flat[off + 1].clamp(min=1.0)
    off = 0
    for g, sz in zip(grads, sizes):
        g.copy_((flat[off:off + sz] / total).reshape(g.shape).to(g.dtype))
        off += sz
    return flat[off] / total, flat[off + 1]


def allreduce_metric(metric_values, group=None):
    """Global mean of a per-query metric (e.g. ndcg@10) over all ranks' queries."""
    v = metric_values.reshape(-1).float()
    acc = torch.stack([v.sum(), torch.tensor(float(v.numel()), device=v.device)])
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=group)
    return acc[0] / acc[1].clamp(min=1.0)


class OverlappedBucketAllReduce:
    """One gradient all-reduce PER STEP that does not stall the step: `depth` copies of the bucket
    [dW (F) | db | loss_sum | count] in rotation.  Step i's kernels write bucket i % depth; its all-reduce
    (the first F + 2 floats: the count of a fixed shard never changes and is summed once, at
    construction) is enqueued with async_op=True straight behind them -- it runs on the communicator's
    own stream UNDER the kernels of step i + 1 -- and the compute stream only waits for it when the
    bucket is taken again (step i + depth) or its result is asked for.  A blocking `dist.all_reduce`
    per step makes the compute stream wait for every collective: 15.7 -> 36.7 us per step at one rank
    (round 2); the all-reduce of a 556-byte bucket is latency, not bandwidth, so hiding it is the whole
    game (reference step being sharded: examples/01-basic-usage.py:66-75).

        red = OverlappedBucketAllReduce(F, count=B, device=dev)
        for i, batch in enumerate(batches):
            flat = red.acquire(i)              # bucket of step i (waits for step i - depth's collective)
            ... launch the step's kernels writing bucket_views(flat) ...
            red.launch(i)                      # all-reduce of step i, asynchronous
            g = red.result(i - 1)              # (optional) step i - 1's summed bucket for the optimiser
        red.flush()
    """

    def __init__(self, F, count, device, group=None, depth=2):
        self.F = int(F)
        self.depth = int(depth)
        self.group = group
        self.buckets = [torch.zeros(self.F + 3, dtype=torch.float32, device=device) for _ in range(self.depth)]
        self.views = [b[:self.F + 2] for b in self.buckets]        # what is all-reduced every step
        self.works = [None] * self.depth
        self.active = dist.is_available() and dist.is_initialized()
        cnt = torch.tensor([float(count)], dtype=torch.float32, device=device)
        if self.active:
            dist.all_reduce(cnt, op=dist.ReduceOp.SUM, group=group)      # once: shards are fixed
        for b in self.buckets:
            b[self.F + 2:self.F + 3].copy_(cnt)
        self.global_count = float(cnt.item())

    def _wait(self, k):
        w = self.works[k]
        if w is not None:
            w.wait()                     # the CURRENT STREAM waits (no host block on GPUs)
            self.works[k] = None

    def acquire(self, i):
        k = i % self.depth
        self._wait(k)
        return self.buckets[k]

    def launch(self, i):
        k = i % self.depth
        if self.active:
            self.works[k] = dist.all_reduce(self.views[k], op=dist.ReduceOp.SUM,
                                            group=self.group, async_op=True)

    def result(self, i):
        """The summed bucket of step i (the current stream is made to wait for its all-reduce)."""
        k = i % self.depth
        self._wait(k)
        return self.buckets[k]

    def flush(self):
        for k in range(self.depth):
            self._wait(k)


# ---------------------------------------------------------------------------------------------
# The same exchange without torch.distributed in the step: c10d's all_reduce costs the host 9 us per
# blocking call and 20 us per async_op=True call on MI355X -- more than the whole 15 us step -- so the
# per-step collective is issued by the C ABI itself (ltr_linear_step_f32 with an overlap handle: a few
# HIP calls and ONE ncclAllReduce, no Python between them).  The communicator is RCCL's own
# (ncclCommInitRank over the unique id rank 0 broadcasts through the existing process group).
# ---------------------------------------------------------------------------------------------
class _NcclUniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_char * 128)]


def _load_rccl():
    cands = [os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), "/opt/rocm/lib/librccl.so", "librccl.so"]
    for c in cands:
        try:
            return ctypes.CDLL(c)
        except OSError:
            continue
    return None


class RcclOverlap:
    """Per-step gradient all-reduce on RCCL, overlapped with the next step, driven from the C ABI.

        ov = RcclOverlap(F, count=B, device=dev)          # needs an initialised process group (any backend)
        for i, batch in enumerate(batches):
            ov.step(i, kind, sigma, X, W, bias, rel, rel_dtype, n, grad_out, B, L, F, loss, workspace)
        bucket = ov.result(i)                              # step i's summed [dW | db | loss_sum | count]
        ov.close()

    depth = 0: in-stream mode -- the all-reduce is enqueued on the step's own stream behind its kernels (no
    side stream, no helper thread: one ncclAllReduce call of host cost per step).

    While a step() is in flight no OTHER collective may be issued on this device (torch.distributed's communicator
    and this one would be used from two threads at once -- with depth > 0 the handle's helper thread enqueues the
    all-reduce -- and RCCL orders collectives per communicator only: different ranks could interleave them
    differently); call flush() first.  The RCCL kernel also takes CU slots: a persistent-workgroup kernel (the parts
    kernel) sized for the whole chip then runs with part of its grid not resident and makes progress through its
    tickets, more slowly.

    `ok` is False when the raw communicator could not be built on EVERY rank (library or symbol missing,
    ncclCommInitRank failing, or the self-check against torch.distributed disagreeing); callers then use
    OverlappedBucketAllReduce instead.  All ranks take the same decision.
    """

    def __init__(self, F, count, device, depth=2, group=None):
        from . import _C
        self.F, self.depth, self.device = int(F), int(depth), torch.device(device)
        self.lib = _C.lib()
        self._C = _C
        self.handle = ctypes.c_void_p(None)
        self.comm = ctypes.c_void_p(None)
        self.ok = False
        self.why = "not built"
        self.buckets = [torch.zeros(self.F + 3, dtype=torch.float32, device=self.device) for _ in range(max(1, self.depth))]
        self.rccl = _load_rccl()
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        good = 1 if (self.rccl is not None and all(hasattr(self.rccl, f) for f in (
            "ncclGetUniqueId", "ncclCommInitRank", "ncclAllReduce", "ncclCommDestroy"))) else 0
        good = self._all_agree(good, group)
        cnt = torch.tensor([float(count)], dtype=torch.float32, device=self.device)
        if dist.is_initialized():
            dist.all_reduce(cnt, op=dist.ReduceOp.SUM, group=group)
        self.global_count = float(cnt.item())
        for b in self.buckets:
            b[self.F + 2:self.F + 3].copy_(cnt)
        if not good:
            self.why = "librccl.so or one of its entry points not found on every rank"
            return
        uid = _NcclUniqueId()
        if rank == 0:
            good = 1 if self.rccl.ncclGetUniqueId(ctypes.byref(uid)) == 0 else 0
        blob = [ctypes.string_at(ctypes.byref(uid), 128) if rank == 0 else None]   # (all 128 bytes: .internal stops at a NUL)
        if dist.is_initialized() and world > 1:
            dist.broadcast_object_list(blob, src=0, group=group)
        if not self._all_agree(good, group):
            self.why = "ncclGetUniqueId failed"
            return
        ctypes.memmove(ctypes.byref(uid), blob[0], 128)
        self.rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, _NcclUniqueId, ctypes.c_int]
        self.rccl.ncclCommInitRank.restype = ctypes.c_int
        with torch.cuda.device(self.device):
            rc = self.rccl.ncclCommInitRank(ctypes.byref(self.comm), world, uid, rank)
        if not self._all_agree(1 if rc == 0 else 0, group):
            self.why = "ncclCommInitRank returned %d" % rc
            return
        fn = ctypes.cast(self.rccl.ncclAllReduce, ctypes.c_void_p)
        with torch.cuda.device(self.device):
            rc = self.lib.ltr_overlap_create(fn, self.comm, self.depth, ctypes.byref(self.handle))
        if not self._all_agree(1 if rc == 0 else 0, group):
            self.why = "ltr_overlap_create returned %d" % rc
            return
        self.ok = bool(self._all_agree(1 if self._self_check(group) else 0, group))
        self.why = "ok" if self.ok else "self-check against torch.distributed failed: %s" % getattr(self, "_check_note", "?")

    def _all_agree(self, flag, group):
        if not dist.is_initialized():
            return int(flag)
        t = torch.tensor([int(flag)], dtype=torch.int32, device=self.device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
        return int(t.item())

    def _self_check(self, group):
        """One all-reduce through the raw communicator against the same through torch.distributed."""
        try:
            rank = dist.get_rank(group) if dist.is_initialized() else 0
            v = (torch.arange(self.F + 2, dtype=torch.float32, device=self.device) + 1.0) * float(rank + 1)
            want = v.clone()
            if dist.is_initialized():
                dist.all_reduce(want, op=dist.ReduceOp.SUM, group=group)
            torch.cuda.synchronize(self.device)
            st = torch.cuda.current_stream(self.device).cuda_stream
            self.rccl.ncclAllReduce.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int,
                                                ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
            self.rccl.ncclAllReduce.restype = ctypes.c_int
            rc = self.rccl.ncclAllReduce(v.data_ptr(), v.data_ptr(), self.F + 2, 7, 0, self.comm, st)
            torch.cuda.synchronize(self.device)
            self._check_note = "rc %d, max abs diff %g" % (rc, float((v - want).abs().max()))
            return rc == 0 and bool(torch.equal(v, want))
        except Exception as exc:  # pragma: no cover - depends on the runtime
            self._check_note = repr(exc)
            return False

    def step(self, i, kind, sigma, X, W, bias, rel, rel_dtype, n, grad_out, B, L, F, loss, workspace, accumulate=False):
        """Step i through ltr_linear_step_f32: its bucket is i % depth, its all-reduce runs under step i + 1."""
        k = i % max(1, self.depth)
        st = torch.cuda.current_stream(self.device).cuda_stream
        with torch.cuda.device(self.device):      # (status word, exchange areas and launches on THIS device)
            self._C.check(self.lib.ltr_linear_step_f32(
                kind, sigma, X.data_ptr(), W.data_ptr(), bias.data_ptr(), rel.data_ptr(), rel_dtype, n.data_ptr(),
                None if grad_out is None else grad_out.data_ptr(), B, L, F, loss.data_ptr(), self.buckets[k].data_ptr(),
                1 if accumulate else 0, workspace.data_ptr(), workspace.numel() * workspace.element_size(),
                self.handle, k, st))

    def sgd_step(self, kind, sigma, X, W, bias, rel, rel_dtype, n, grad_out, B, L, F, lr, loss, workspace):
        """One synchronous-SGD step through ltr_linear_sgd_step_f32 (in-stream handles only): kernels, the step's
        all-reduce behind them on the same stream, then W -= lr * dW, bias -= lr * db -- the next step scores with
        weights that waited for this step's collective (examples/01-basic-usage.py:72-75, sharded)."""
        st = torch.cuda.current_stream(self.device).cuda_stream
        with torch.cuda.device(self.device):      # (status word, exchange areas and launches on THIS device)
            self._C.check(self.lib.ltr_linear_sgd_step_f32(
                kind, sigma, X.data_ptr(), W.data_ptr(), bias.data_ptr(), rel.data_ptr(), rel_dtype, n.data_ptr(),
                None if grad_out is None else grad_out.data_ptr(), B, L, F, float(lr), loss.data_ptr(),
                self.buckets[0].data_ptr(), workspace.data_ptr(), workspace.numel() * workspace.element_size(),
                self.handle, st))

    def result(self, i):
        """Step i's summed bucket; the current stream is made to wait for its all-reduce."""
        k = i % max(1, self.depth)
        self._C.check(self.lib.ltr_overlap_wait(self.handle, k, torch.cuda.current_stream(self.device).cuda_stream))
        return self.buckets[k]

    def flush(self):
        if self.handle:
            self._C.check(self.lib.ltr_overlap_flush(self.handle))

    def close(self):
        if self.handle:
            self.lib.ltr_overlap_destroy(self.handle)
            self.handle = ctypes.c_void_p(None)
        if self.comm and self.rccl is not None:
            try:
                self.rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
                self.rccl.ncclCommDestroy(self.comm)
            except Exception:  # pragma: no cover
                pass
            self.comm = ctypes.c_void_p(None)
        self.ok = False


# ---------------------------------------------------------------------------------------------
# The same per-step exchange without ANY collective library in the step: the mailbox all-reduce of the C ABI
# (include/ltr_hip.h: ltr_mailbox_*): peers' mailboxes mapped through HIP IPC, one small kernel per rank stores
# the bucket's tagged granules into every peer and adds what arrives in rank order -- bit-identical sums on all
# ranks, one hop over xGMI, and inside ltr_linear_sgd_step_f32 the weight update rides in the same kernel.
# ---------------------------------------------------------------------------------------------
class MailboxOverlap:
    """In-stream gradient all-reduce of the fused step through the mailbox all-reduce.

        mb = MailboxOverlap(F, count=B, device=dev)       # needs an initialised process group (any backend)
        if mb.ok:
            for batch in batches:
                mb.sgd_step(kind, sigma, X, W, bias, rel, rel_dtype, n, grad_out, B, L, F, lr, loss, workspace)
        mb.close()

    `ok` is False -- on EVERY rank -- when the mailboxes could not be set up (IPC handle not obtainable or not
    mappable, e.g. ranks on different nodes) or the self-check against torch.distributed disagrees; callers then
    fall back to RcclOverlap / torch.distributed.  Interface as RcclOverlap(depth=0)."""

    def __init__(self, F, count, device, group=None):
        from . import _C
        self.F, self.depth, self.device = int(F), 0, torch.device(device)
        self.lib = _C.lib()
        self._C = _C
        self.handle = ctypes.c_void_p(None)       # overlap handle
        self.mbox = ctypes.c_void_p(None)         # mailbox handle (= comm of the overlap handle)
        self.ok = False
        self.why = "not built"
        self.group = group
        self.buckets = [torch.zeros(self.F + 3, dtype=torch.float32, device=self.device)]
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        cnt = torch.tensor([float(count)], dtype=torch.float32, device=self.device)
        if dist.is_initialized():
            dist.all_reduce(cnt, op=dist.ReduceOp.SUM, group=group)
        self.global_count = float(cnt.item())
        self.buckets[0][self.F + 2:self.F + 3].copy_(cnt)
        if world > 16:
            self.why = "more than 16 ranks"
            return
        blob = ctypes.create_string_buffer(64)
        with torch.cuda.device(self.device):
            rc = self.lib.ltr_mailbox_create(rank, world, self.F + 2, ctypes.byref(self.mbox), blob)
        if not self._all_agree(1 if rc == 0 else 0):
            self.why = "ltr_mailbox_create returned %d (hipIpcGetMemHandle?)" % rc
            return
        handles = [None] * world
        if dist.is_initialized() and world > 1:
            dist.all_gather_object(handles, blob.raw, group=group)
        else:
            handles = [blob.raw]
        allh = ctypes.create_string_buffer(b"".join(handles), 64 * world)
        with torch.cuda.device(self.device):
            rc = self.lib.ltr_mailbox_connect(self.mbox, allh)
        if not self._all_agree(1 if rc == 0 else 0):
            self.why = "ltr_mailbox_connect returned %d (hipIpcOpenMemHandle: ranks on one node, one process per GPU?)" % rc
            return
        fn = ctypes.cast(self.lib.ltr_mailbox_allreduce, ctypes.c_void_p)
        with torch.cuda.device(self.device):
            rc = self.lib.ltr_overlap_create(fn, self.mbox, 0, ctypes.byref(self.handle))
        if not self._all_agree(1 if rc == 0 else 0):
            self.why = "ltr_overlap_create returned %d" % rc
            return
        self.ok = bool(self._all_agree(1 if self._self_check() else 0))
        self.why = "ok" if self.ok else "self-check against torch.distributed failed: %s" % getattr(self, "_check_note", "?")

    def _all_agree(self, flag):
        if not dist.is_initialized():
            return int(flag)
        t = torch.tensor([int(flag)], dtype=torch.int32, device=self.device)
        if dist.get_backend(self.group) == "gloo":
            t = t.cpu()
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
        return int(t.item())

    def allreduce_(self, vec):
        """In-place sum of a float32 device vector of at most F + 2 elements over the ranks (current stream)."""
        st = torch.cuda.current_stream(self.device).cuda_stream
        with torch.cuda.device(self.device):
            rc = self.lib.ltr_mailbox_allreduce(vec.data_ptr(), vec.data_ptr(), vec.numel(), 7, 0, self.mbox, st)
        if rc != 0:
            raise RuntimeError("ltr_mailbox_allreduce failed")
        return vec

    def _self_check(self):
        """Three all-reduces (both halves of the mailbox, and once more) against torch.distributed, under a SHORT
        time budget (a peer that is not there is not going to come).  Whatever happens, the sticky device status is
        left CLEAN: a failed check must not make the fall-back (RcclOverlap / plain steps) fail with the
        LTR_ERR_TIMEOUT this check provoked."""
        ok = True
        budget_ms = int(os.environ.get("LTR_MAILBOX_CHECK_TIMEOUT_MS", "5000"))
        previous_ms = 0
        try:
            previous_ms = int(self.lib.ltr_mailbox_set_timeout_ms(budget_ms))      # (returns the budget it replaces)
            rank = dist.get_rank(self.group) if dist.is_initialized() else 0
            for rep in range(3):
                v = (torch.arange(self.F + 2, dtype=torch.float32, device=self.device) + 1.0 + rep) * float(rank + 1) * 0.37
                want = v.clone()
                if dist.is_initialized():
                    w = want.cpu() if dist.get_backend(self.group) == "gloo" else want
                    dist.all_reduce(w, op=dist.ReduceOp.SUM, group=self.group)
                    want = w.to(self.device)
                try:                       # (local failures only: the torch.distributed calls around stay in step)
                    self.allreduce_(v)
                    torch.cuda.synchronize(self.device)
                    # (the mailbox adds in rank order, the library in its own: equal to a few ulps, not bit for bit)
                    ok = ok and bool(torch.allclose(v, want, rtol=1e-5, atol=1e-6))
                    self._check_note = "max abs diff %g" % float((v - want).abs().max())
                except Exception as exc:  # pragma: no cover - depends on the runtime
                    self._check_note = repr(exc)
                    ok = False
                # a timed-out all-reduce is not repeated (one budget, not three) -- but the decision to stop is taken by ALL
                # ranks together: a failure seen by one rank only (a mapping visible in one direction, a late start) must not
                # leave the others in the next repetition's all_reduce while this one goes on to _all_agree
                ok = bool(self._all_agree(1 if ok else 0))
                if not ok:
                    break
            ok = ok and self.lib.ltr_device_status(0) == 0
        except Exception as exc:  # pragma: no cover - depends on the runtime
            self._check_note = repr(exc)
            ok = False
        finally:
            if previous_ms > 0:
                self.lib.ltr_mailbox_set_timeout_ms(previous_ms)
            if not ok:
                try:
                    torch.cuda.synchronize(self.device)
                except Exception:  # pragma: no cover
                    pass
                self.lib.ltr_device_status(1)      # the fall-back starts clean
        return ok

    def sgd_step(self, kind, sigma, X, W, bias, rel, rel_dtype, n, grad_out, B, L, F, lr, loss, workspace):
        st = torch.cuda.current_stream(self.device).cuda_stream
        with torch.cuda.device(self.device):      # (status word, exchange areas and launches on THIS device)
            self._C.check(self.lib.ltr_linear_sgd_step_f32(
                kind, sigma, X.data_ptr(), W.data_ptr(), bias.data_ptr(), rel.data_ptr(), rel_dtype, n.data_ptr(),
                None if grad_out is None else grad_out.data_ptr(), B, L, F, float(lr), loss.data_ptr(),
                self.buckets[0].data_ptr(), workspace.data_ptr(), workspace.numel() * workspace.element_size(),
                self.handle, st))

    def step(self, i, kind, sigma, X, W, bias, rel, rel_dtype, n, grad_out, B, L, F, loss, workspace, accumulate=False):
        st = torch.cuda.current_stream(self.device).cuda_stream
        with torch.cuda.device(self.device):      # (status word, exchange areas and launches on THIS device)
            self._C.check(self.lib.ltr_linear_step_f32(
                kind, sigma, X.data_ptr(), W.data_ptr(), bias.data_ptr(), rel.data_ptr(), rel_dtype, n.data_ptr(),
                None if grad_out is None else grad_out.data_ptr(), B, L, F, loss.data_ptr(), self.buckets[0].data_ptr(),
                1 if accumulate else 0, workspace.data_ptr(), workspace.numel() * workspace.element_size(),
                self.handle, 0, st))

    def result(self, i=0):
        return self.buckets[0]

    def flush(self):
        torch.cuda.synchronize(self.device)

    def close(self):
        if self.handle:
            self.lib.ltr_overlap_destroy(self.handle)
            self.handle = ctypes.c_void_p(None)
        if self.mbox:
            if dist.is_initialized():
                try:
                    torch.cuda.synchronize(self.device)
                    dist.barrier(group=self.group)       # nobody unmaps a mailbox a peer may still write
                except Exception:  # pragma: no cover
                    pass
            self.lib.ltr_mailbox_destroy(self.mbox)
            self.mbox = ctypes.c_void_p(None)
        self.ok = False
