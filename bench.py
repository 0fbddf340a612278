#!/usr/bin/env python
"""bench.py -- headline benchmark of the MI355X ranking-loss hot path.

    python bench.py --gpus N --steps K --warmup W            (N=1: run directly)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): queries/sec of loss forward+backward on the MSLR-WEB30K-shaped padded
batch B=1024, list_len=128, 136 features (configs[1]), plus achieved HBM GB/s of the dominant
kernel against the gfx950 roofline.  One "step" is the reference's training-step slice on one
batch (examples/01-basic-usage.py:70-75):

    loss = loss_fn(Linear(136, 1)(xs), ys, n).mean();  loss.backward()

computed by the fused HIP path (scores -> PairwiseHingeLoss -> d/dW, d/db) through the C ABI,
launched EAGERLY (no hipGraph in `value`), inputs resident in HBM.

Cold working set.  Consecutive steps take DIFFERENT batches: NBUF distinct synthetic batches whose
feature tensors total more than the 256 MiB Infinity Cache are visited in rotation, so every step
streams its features from HBM (a single 71 MB batch re-read every step would sit in the cache and
flatter the roofline).  `roofline.frac` is computed on the bytes a launch MUST move -- only the
n[b] real rows of every query are read -- next to the padded-formula figure of SURVEY.md 8(d);
`roofline.full_lists` is the n == list_len twin, where both figures coincide.

Timing.  The timed region holds max(K, enough steps for >= 50 ms) steps, bracketed by barrier +
synchronize; it is repeated 5 times and the MEDIAN repeat is reported (max over ranks per repeat).

N > 1: one process per GPU (the driver's `torch.distributed.run` launch; run BARE, `bench.py --gpus N` starts its N ranks
itself; WORLD_SIZE must equal --gpus, and the line's n_gpus is the number of ranks that ran, config.devices where).  Queries are
independent units -> no data-path collective.  Weak scaling by default (every rank owns its own B queries per step); `--shard`
splits BASELINE's global batch over the ranks instead (C4: 32 queries / GPU at N = 8).  The only exchange is the scorer
gradient, ONE all-reduce PER STEP of [dW (F) | db | loss_sum] (north_star / SURVEY.md 8(e); config.allreduce_every = 1).  The
headline step is ONE launch per rank at every N (ltr_linear_sgd_lazy_step_dp_f32: the all-reduce rides in the fused launch's
reducer workgroups, through HIP-IPC mapped peer mailboxes over xGMI); the same steps with the mailbox all-reduce as a launch of
its own and with an in-stream ncclAllReduce (RCCL) are timed next to it (extra.mailbox_three_launch_step_us,
extra.rccl_instream_step_us); a trial under a short time budget decides, on every rank alike, whether the headline falls back to
them (config.allreduce says which ran).  `--accum K` (K > 1) is the gradient-accumulation variant, reported in `extra` at N > 1.

Prints ONE JSON line on rank 0 (metric, value, ..., roofline, cpu_baseline, extra).
"""
import argparse
import json
import math
import os
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E peak (MI355X_MICROARCH.md); ~6300 achievable
CU_COUNT = 256
CLOCK_HZ = 2.4e9            # max shader clock (MI355X_MICROARCH.md)
VALU_ISSUE_PER_CYCLE = 512  # wave-instructions per cycle: 1024 SIMDs x one wave64 instruction per 2 cycles
                            # (MI355X_MICROARCH.md, SIMD-32).  Measured (profiles/r05_valu_issue.txt): a SIMD with
                            # >= 2 waves issues v_add_f32 every 2.26 cycles, v_fma_f32 2.64, DPP moves 4.4, v_pk_fma_f32
                            # 5.1; ONE wave alone issues every 4.9-5.9 cycles.  Rounds 1-4 used 256 (4 cycles, inferred
                            # from SQ_ACTIVE_INST_VALU counting quad-cycles): every valu_issue_frac of those rounds
                            # is twice the figure on this basis.
L3_BYTES = 256 * 2 ** 20

WORKLOADS = {
    # name: (B, L, F, loss kind)   -- BASELINE.json configs
    "c2": (1024, 128, 136, "hinge"),
    "c3": (1024, 128, 136, "ndcg2"),
    "c4": (256, 1000, 220, "dcg_hinge"),
    "c5": (512, 512, 700, "hinge"),
    # (not a BASELINE config: C2's shape at half the batch, for FUNCTIONAL multi-rank runs with the ranks on one GPU --
    # tests/test_gpu_bench.py --, where the ranks' grids must be resident side by side)
    "c2s": (512, 128, 136, "hinge"),
}
PLAN_NAMES = {1: "linear_regtile_kernel", 2: "linear_cluster_kernel", 3: "linear_pairwise_kernel",
              4: "linear_parts_kernel"}


def synth(B, L, F, seed, device):
    """SURVEY.md section 8(d) recipe (CPU generator, then .to(device))."""
    g = torch.Generator().manual_seed(seed)
    scores = torch.randn(B, L, generator=g)
    relevance = torch.randint(0, 5, (B, L), generator=g)
    n = torch.randint(1, L + 1, (B,), generator=g)
    X = torch.randn(B, L, F, generator=g)
    return scores.to(device), relevance.to(device), n.to(device), X.to(device)


def nbuf_for(B, L, F):
    """Distinct batches in rotation: their feature tensors together exceed the Infinity Cache."""
    xbytes = 4 * B * L * F
    return int(min(8, max(2, math.ceil(1.25 * L3_BYTES / xbytes))))


def make_batches(B, L, F, nbuf, seed0, dev, full=False):
    out = []
    for i in range(nbuf):
        scores, rel, n, X = synth(B, L, F, seed0 + i, dev)
        if full:
            n = torch.full_like(n, L)
        out.append({"scores": scores, "rel": rel, "n": n, "X": X,
                    "rows": int(n.clamp(max=L).sum())})
    return out


def moved_bytes(batches, B, L, F):
    """Bytes one fused launch must move, averaged over the batches in rotation: the n[b] real rows
    (features 4F + int64 label 8 each), n 8 B per query, W/bias, loss 4 B and the (F+1) partials
    per query."""
    rows = sum(b["rows"] for b in batches) / float(len(batches))
    return rows * (4 * F + 8) + B * 8 + 4 * (F + 1) + B * 4 + 4 * (F + 1) * B


def padded_bytes(B, L, F):
    """algorithmic bytes per query of the fused step, padded formula."""
    return B * (4 * L * F + 8 * L + 8 + 4 + 4 * (F + 1)) + 4 * (F + 1)


def time_events(fn, iters):
    """Per-launch duration (us) of fn() with ONE HIP event pair per launch on torch's current
    stream (the stream the kernels are launched on).  Includes the event-record overhead
    (~4-5 us on this stack): an upper bound, reported next to the graph-batched figure."""
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(iters)]
    ends = [torch.cuda.Event(enable_timing=True) for _ in range(iters)]
    torch.cuda.synchronize()
    for i in range(iters):
        starts[i].record()
        fn(i)
        ends[i].record()
    torch.cuda.synchronize()
    per = sorted(s.elapsed_time(e) * 1e3 for s, e in zip(starts, ends))
    return sum(per) / len(per), per[len(per) // 2], per[0]


GRAPHS_OK = True     # cleared when a process group exists (see main)


def try_graph(step, warm=2):
    """Capture `step()` into a hipGraph; returns the replay function or None."""
    if not GRAPHS_OK:
        return None
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warm):
                step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            step()
        torch.cuda.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        return graph.replay
    except Exception as exc:  # pragma: no cover - depends on the runtime
        sys.stderr.write("[bench] hipGraph capture unavailable: %r\n" % (exc,))
        torch.cuda.synchronize()
        return None


def time_launches(fn, nbuf, rounds=4, replays=10):
    """Average launch duration (us) of fn(i), i cycling over the `nbuf` batches: rounds*nbuf
    back-to-back launches captured into a hipGraph, replayed `replays` times between ONE HIP event
    pair on the launch stream -- kernel + dependent-launch boundary, no host or event overhead.
    Eager back-to-back launches if capture is unavailable."""
    count = rounds * nbuf

    def many():
        for i in range(count):
            fn(i)
    replay = try_graph(many, warm=1)
    run = replay if replay is not None else many
    run()
    torch.cuda.synchronize()
    start = torch.cuda.Event(enable_timing=True)
    end = torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(replays):
        run()
    end.record()
    torch.cuda.synchronize()
    return start.elapsed_time(end) * 1e3 / (count * replays), replay is not None


def time_region(fn, steps, barrier, repeats=5, reduce_max=None, finish=None):
    """`repeats` timed regions of `steps` calls of fn(i), each bracketed by barrier + synchronize;
    returns the list of elapsed seconds (max over ranks when reduce_max is given).  finish(): work the steps left
    pending (the lazy step's last update), INSIDE the timed region."""
    out = []
    k = 0
    for _ in range(repeats):
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _i in range(steps):
            fn(k)
            k += 1
        if finish is not None:
            finish()
        torch.cuda.synchronize()
        barrier()
        dt = time.perf_counter() - t0
        if reduce_max is not None:
            dt = reduce_max(dt)
        out.append(dt)
    return out


def median(xs):
    s = sorted(xs)
    return s[len(s) // 2]


# learning rate of the timed synchronous-SGD step: small enough that the weights stay where the synthetic
# batches were drawn for over the ~10^5 steps of a run (the step's cost does not depend on the values)
SGD_LR = 1e-6


def bucket_views(flat, F):
    """The one all-reduced bucket of a step, (F + 3) floats: [dW (F) | db | loss_sum | count]."""
    return flat[:F], flat[F:F + 1], flat[F + 1:F + 2], flat[F + 2:F + 3]


class FusedStep:
    """The step through the C ABI: ltr_linear_partials_f32 + ltr_linear_reduce_accum_f32, writing
    the bucket [dW | db | loss_sum | count] (bucket_views)."""

    def __init__(self, kind, B, L, F, dev, n_gpus=1, seed=0):
        from pytorchltr_amd import _C
        self._C = _C
        self.lib = _C.lib()
        self.kind_id = getattr(_C, kind.upper())
        self.B, self.L, self.F = B, L, F
        g = torch.Generator().manual_seed(seed)
        bound = 1.0 / math.sqrt(F)
        self.W = ((torch.rand(F, generator=g) * 2 - 1) * bound).to(dev)
        self.bias = ((torch.rand(1, generator=g) * 2 - 1) * bound).to(dev)
        self.lossv = torch.empty(B, device=dev)
        self.ws_bytes = self.lib.ltr_linear_workspace_bytes(B, L, F)
        self.part = torch.empty(self.ws_bytes // 4, device=dev)
        self.flat = torch.zeros(F + 3, device=dev)
        self.flat[F + 2] = float(B)
        # uniform upstream gradient 1/(B*N): after the SUM all-reduce the bucket holds the gradient
        # of the GLOBAL mean loss -- what `.mean().backward()` gives on the unsharded batch
        self.go = torch.full((B,), 1.0 / (B * n_gpus), device=dev)
        self.plan = PLAN_NAMES.get(self.lib.ltr_linear_fused_plan(self.kind_id, B, L, F), "?")
        self.pending = 0
        self.mailbox = None                  # N > 1: the ltr_mailbox_* handle of the data-parallel lazy step
        self.lazy_scale = 1.0 / (B * n_gpus)
        self._W0, self._b0 = self.W.clone(), self.bias.clone()

    def reset_weights(self):
        self.W.copy_(self._W0)
        self.bias.copy_(self._b0)
        self.pending = 0

    @staticmethod
    def _stream():
        return torch.cuda.current_stream().cuda_stream

    def kernel(self, batch):
        self._C.check(self.lib.ltr_linear_partials_f32(
            self.kind_id, 1.0, batch["X"].data_ptr(), self.W.data_ptr(), self.bias.data_ptr(),
            batch["rel"].data_ptr(), self._C.LABEL_I64, batch["n"].data_ptr(), self.B, self.L, self.F,
            self.lossv.data_ptr(), None, self.part.data_ptr(), self._stream()))

    def reduce(self, accumulate=False, flat=None):
        fp = (self.flat if flat is None else flat).data_ptr()
        self._C.check(self.lib.ltr_linear_reduce_accum_f32(
            self.part.data_ptr(), self.go.data_ptr(), self.lossv.data_ptr(), self.B, self.F, fp,
            fp + 4 * self.F, fp + 4 * (self.F + 1), 1 if accumulate else 0, self._stream()))

    def step(self, batch, accumulate=False, flat=None):
        """The step in ONE C-ABI call (ltr_linear_step_f32 = partials + reduce into the bucket)."""
        self._C.check(self.lib.ltr_linear_step_f32(
            self.kind_id, 1.0, batch["X"].data_ptr(), self.W.data_ptr(), self.bias.data_ptr(),
            batch["rel"].data_ptr(), self._C.LABEL_I64, batch["n"].data_ptr(), self.go.data_ptr(), self.B, self.L,
            self.F, self.lossv.data_ptr(), (self.flat if flat is None else flat).data_ptr(), 1 if accumulate else 0,
            self.part.data_ptr(), self.ws_bytes, None, 0, self._stream()))

    def sgd_step(self, batch, lr=SGD_LR):
        """The reference's whole training step in ONE C-ABI call (ltr_linear_sgd_step_f32): partials + reduce into
        the bucket + W -= lr * dW, bias -= lr * db (at one rank the update rides in the reduction kernel)."""
        self._C.check(self.lib.ltr_linear_sgd_step_f32(
            self.kind_id, 1.0, batch["X"].data_ptr(), self.W.data_ptr(), self.bias.data_ptr(),
            batch["rel"].data_ptr(), self._C.LABEL_I64, batch["n"].data_ptr(), self.go.data_ptr(), self.B, self.L,
            self.F, float(lr), self.lossv.data_ptr(), self.flat.data_ptr(), self.part.data_ptr(), self.ws_bytes,
            None, self._stream()))

    def lazy_step(self, batch, lr=SGD_LR):
        """The same step, the update applied lazily (ltr_linear_sgd_lazy_step_dp_f32): ONE launch -- this batch's per-query
        gradient rows, and the previous batch's reduction + W -= lr * dW inside it; lazy_flush() applies the last one.
        With self.mailbox set (N > 1) the reducers all-reduce their column sums over the ranks inside the same launch."""
        self._C.check(self.lib.ltr_linear_sgd_lazy_step_dp_f32(
            self.kind_id, 1.0, batch["X"].data_ptr(), self.W.data_ptr(), self.bias.data_ptr(),
            batch["rel"].data_ptr(), self._C.LABEL_I64, batch["n"].data_ptr(), self.B, self.L, self.F, float(lr),
            self.lossv.data_ptr(), self.flat.data_ptr(), self.part.data_ptr(), self.ws_bytes, self.pending,
            self.lazy_scale, None, 0, self.mailbox, self._stream()))
        self.pending = self.B

    def lazy_flush(self, lr=SGD_LR):
        self._C.check(self.lib.ltr_linear_sgd_flush_dp_f32(
            self.kind_id, self.W.data_ptr(), self.bias.data_ptr(), self.pending, self.L, self.F, float(lr), self.lazy_scale, None, 0,
            self.lossv.data_ptr(), self.flat.data_ptr(), self.part.data_ptr(), self.mailbox, self._stream()))
        self.pending = 0

    def step_two_calls(self, batch, accumulate=False, flat=None):
        self.kernel(batch)
        self.reduce(accumulate, flat)


def lazy_kernel_us(fs, batches, nbuf, count=400):
    """Average duration (us) of the lazy step's launch itself: an event pair per launch, recorded by the command processor
    at the kernel's begin and end (include/ltr_hip.h: ltr_debug_kernel_events); None without the hook / the HIP runtime."""
    import ctypes
    lib = fs.lib
    if not hasattr(lib, "ltr_debug_kernel_events"):
        return None
    try:
        hip = ctypes.CDLL("libamdhip64.so")
    except OSError:
        return None
    evs = []
    for _ in range(2 * count):
        e = ctypes.c_void_p()
        if hip.hipEventCreate(ctypes.byref(e)) != 0:
            return None
        evs.append(e)
    for i in range(2 * nbuf):
        fs.lazy_step(batches[i % nbuf])
    for i in range(count):
        lib.ltr_debug_kernel_events(evs[2 * i], evs[2 * i + 1])
        fs.lazy_step(batches[i % nbuf])
    lib.ltr_debug_kernel_events(None, None)          # (disarmed, whatever the last launch did with the pair)
    fs.lazy_flush()
    torch.cuda.synchronize()
    per = []
    for i in range(count):
        ms = ctypes.c_float()
        if hip.hipEventElapsedTime(ctypes.byref(ms), evs[2 * i], evs[2 * i + 1]) == 0:
            per.append(ms.value * 1e3)
    for e in evs:
        hip.hipEventDestroy(e)
    if not per:
        return None
    per.sort()
    return {"avg": sum(per) / len(per), "median": per[len(per) // 2], "min": per[0], "count": len(per)}


PMC_STALE = set()     # workloads whose committed counters belong to older kernel sources



def pmc_record(workload):
    """PMC-derived per-launch figures of the workload's kernels (profiles/pmc_<workload>.json,
    written by scripts/pmc_to_json.py from separate rocprofv3 --pmc passes of this command)."""
    path = os.path.join(ROOT, "profiles", "pmc_%s.json" % workload)
    if os.path.exists(path):
        with open(path) as fh:
            doc = json.load(fh)
        # stale-counter guard: counters taken from other kernel sources than the tree's
        # are not printed -- `traffic` / `valu_issue_frac` then read null until the passes are re-run
        sys.path.insert(0, os.path.join(ROOT, "scripts"))
        try:
            from pmc_to_json import csrc_sha256
            if doc.get("_csrc_sha256") != csrc_sha256(ROOT):
                PMC_STALE.add(workload)
                return {}
        except Exception:  # pragma: no cover
            return {}
        return doc
    return {}


def valu_frac(valu_insts, us):
    if not valu_insts or not us:
        return None
    return valu_insts / (VALU_ISSUE_PER_CYCLE * CLOCK_HZ * us * 1e-6)


def measure_config(name, B, L, F, kind, dev, seed0, full=False, events=False, label=None):
    """Cold (rotating batches) timing of the fused step and its dominant kernel for one shape."""
    nbuf = nbuf_for(B, L, F)
    batches = make_batches(B, L, F, nbuf, seed0, dev, full=full)
    fs = FusedStep(kind, B, L, F, dev)
    for i in range(2 * nbuf):
        fs.step(batches[i % nbuf])
    torch.cuda.synchronize()
    k_us, graphed = time_launches(lambda i: fs.kernel(batches[i % nbuf]), nbuf)
    s_us, _ = time_launches(lambda i: fs.step(batches[i % nbuf]), nbuf)
    moved = moved_bytes(batches, B, L, F)
    padded = padded_bytes(B, L, F)
    pmc = pmc_record(name).get(fs.plan, {}) if not full else {}
    res = {
        "workload": label or "%s: Linear(%d,1) + %s, B=%d, list_len=%d, %s" % (
            name, F, kind, B, L, "n == list_len" if full else "n~U[1,%d]" % L),
        "plan": fs.plan, "batches_in_rotation": nbuf,
        "rotating_feature_bytes": nbuf * 4 * B * L * F,
        "kernel_us": k_us, "step_us": s_us, "step_queries_per_s": B / (s_us * 1e-6),
        "timing": "HIP events around %s back-to-back launches over %d rotating batches" % (
            "hipGraph-replayed" if graphed else "eager", nbuf),
        "moved_bytes_per_launch": moved, "padded_formula_bytes_per_launch": padded,
        "moved_GBs": moved / (k_us * 1e-6) / 1e9, "frac_moved": moved / (k_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
        "frac_padded_formula": padded / (k_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
        "traffic": pmc.get("hbm_bytes_per_launch"),
        "valu_issue_frac": valu_frac(pmc.get("valu_insts_per_launch"), k_us),
    }
    if events:
        avg, med, mn = time_events(lambda i: fs.kernel(batches[i % nbuf]), 100)
        res["kernel_us_single_launch_event_pair"] = {"avg": avg, "median": med, "min": mn}
    # the synchronous-SGD step as ONE launch (the lazy step: ltr_linear_sgd_lazy_step_f32) where the register tile takes the shape:
    # eager back-to-back steps + the flush, wall clock (a captured graph cannot carry the lazy step's tag) -- next to step_us,
    # the graph-timed two launches without the weight update
    if fs.plan == "linear_regtile_kernel" and hasattr(fs.lib, "ltr_linear_sgd_lazy_step_f32") and not full:
        for i in range(2 * nbuf):
            fs.lazy_step(batches[i % nbuf])
        fs.lazy_flush()
        torch.cuda.synchronize()
        per = []
        K = 2000
        for _ in range(3):
            t0 = time.perf_counter()
            for i in range(K):
                fs.lazy_step(batches[i % nbuf])
            fs.lazy_flush()
            torch.cuda.synchronize()
            per.append((time.perf_counter() - t0) / K * 1e6)
        res["lazy_sgd_step_us"] = sorted(per)[1]
        res["lazy_sgd_step_queries_per_s"] = B / (res["lazy_sgd_step_us"] * 1e-6)
    return res, fs, batches


def mlp_extra(kind, batches, no_graph):
    """the guide's 136-50-10-1 ReLU MLP + loss + backward as one fused MFMA step
    (ltr_mlp_pairwise_f32), next to the same step written with torch layers + the HIP loss module."""
    from pytorchltr_amd import _C, fused
    from pytorchltr_amd import loss as L_
    X0 = batches[0]["X"]
    B, L, F = X0.shape
    H1, H2 = 50, 10
    if not fused.mlp_supported(L, F, H1, H2):
        return None
    dev = X0.device
    nbuf = len(batches)
    torch.manual_seed(0)
    m = fused.FusedMLPLoss(F, kind, hidden=(H1, H2)).to(dev)
    lib = _C.lib()
    P = lib.ltr_mlp_param_count(F, H1, H2)
    ws_bytes = lib.ltr_mlp_workspace_bytes(B, F, H1, H2)
    ws = torch.empty(ws_bytes // 4, device=dev)
    grads = torch.empty(P, device=dev)
    lossv = torch.empty(B, device=dev)
    lsum = torch.zeros(1, device=dev)
    params = [p.detach().contiguous() for p in (m.l1.weight, m.l1.bias, m.l2.weight, m.l2.bias,
                                                m.l3.weight, m.l3.bias)]
    kind_id = getattr(_C, kind.upper())

    def launch(i):
        b = batches[i % nbuf]
        _C.check(lib.ltr_mlp_pairwise_f32(
            kind_id, 1.0, b["X"].data_ptr(), *[p.data_ptr() for p in params], b["rel"].data_ptr(),
            _C.LABEL_I64, b["n"].data_ptr(), None, B, L, F, H1, H2, lossv.data_ptr(), None,
            grads.data_ptr(), lsum.data_ptr(), ws.data_ptr(), ws_bytes,
            torch.cuda.current_stream().cuda_stream))

    for i in range(nbuf):
        launch(i)
    us, _ = time_launches(launch, nbuf, rounds=2, replays=10)
    docs = sum(b["rows"] for b in batches) / float(nbuf)
    flops_per_doc = 2 * (F * H1 + H1 * H2 + H2) + 2 * (F * H1 + 2 * H1 * H2 + H2)   # fwd + bwd, no dX
    out = {"step": "MLP %d-%d-%d-1 + %s fwd+bwd (2 launches: mlp_tile_kernel + mlp_reduce_kernel), "
                   "%d rotating batches" % (F, H1, H2, kind, nbuf),
           "us_per_step": us, "queries_per_s": B / (us * 1e-6),
           "useful_TFLOPs": docs * flops_per_doc / (us * 1e-6) / 1e12,
           "f32_mfma_peak_TFLOPs": 157.3,
           "frac_of_f32_mfma_peak": docs * flops_per_doc / (us * 1e-6) / 1e12 / 157.3}
    # the launch-geometry ceiling of that step: the tile kernel's grid, fills, LDS image, barriers and
    # MFMA streams with nothing else (the same query assignment; no labels, per-document layers 2-3, parking, pair pass, partial
    # vectors, reduction launch) -- ltr_debug_mlp_probe_f32, cold over the same rotation
    if F == 136 and L <= 128 and hasattr(lib, "ltr_debug_mlp_probe_f32"):
        pout = torch.empty(4096, device=dev)

        def probe(i):
            b = batches[i % nbuf]
            _C.check(lib.ltr_debug_mlp_probe_f32(b["X"].data_ptr(), *[p.data_ptr() for p in params], b["n"].data_ptr(),
                                                 B, L, F, H1, H2, pout.data_ptr(), torch.cuda.current_stream().cuda_stream))
        for i in range(nbuf):
            probe(i)
        pus, _ = time_launches(probe, nbuf, rounds=2, replays=10)
        # MFMAs the geometry issues: 80 forward + 88 backward per 32-row fill and wave, 4 waves; 2048 flop each
        fills = sum(int(((b["n"].clamp(max=L) + 31) // 32).clamp(min=1).sum()) for b in batches) / float(nbuf)
        mfma_flops = fills * 4 * 168 * 2048.0
        out["launch_ceiling"] = {
            "us": pus, "step_over_ceiling": us / pus,
            "useful_frac_of_f32_mfma_peak_at_ceiling": docs * flops_per_doc / (pus * 1e-6) / 1e12 / 157.3,
            "issued_mfma_TFLOPs_at_ceiling": mfma_flops / (pus * 1e-6) / 1e12,
            "issued_over_useful_flops": mfma_flops / (docs * flops_per_doc),
            "what": "mlp_tile_kernel<PROBE>: same grid (2 workgroups of 4 waves per CU), same 32-row fills requested / written to "
                    "the LDS image / read back, same barriers, same 168 MFMAs (v_mfma_f32_16x16x4_f32) per fill and wave; same "
                    "assignment of queries to workgroups; no labels, owner-wave layers 2-3, parking, pair pass, partial vectors or reduction launch"}
    loss_fn = {"hinge": L_.PairwiseHingeLoss, "dcg_hinge": L_.PairwiseDCGHingeLoss,
               "logistic": L_.PairwiseLogisticLoss, "arp1": L_.LambdaARPLoss1, "arp2": L_.LambdaARPLoss2,
               "ndcg1": L_.LambdaNDCGLoss1, "ndcg2": L_.LambdaNDCGLoss2}[kind]()
    ps = list(m.parameters())
    b0 = batches[0]

    def unfused():
        for p_ in ps:
            p_.grad = None
        loss_fn(m.score(b0["X"]), b0["rel"], b0["n"]).mean().backward()
    for _ in range(3):
        unfused()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        unfused()
    torch.cuda.synchronize()
    out["unfused_torch_layers_plus_loss_module_eager_us"] = (time.perf_counter() - t0) / 30 * 1e6

    # the drop-in module of the guide's step (FusedMLPLoss: nn.Linear parameters, autograd), eager:
    # what a training script gets per `loss = model(xs, ys, n); loss.backward()`
    def module_step(i):
        b = batches[i % nbuf]
        for p_ in ps:
            p_.grad = None
        m(b["X"], b["rel"], b["n"]).backward()
    for i in range(10):
        module_step(i)
    torch.cuda.synchronize()
    reps = []
    for _ in range(5):
        t0 = time.perf_counter()
        for i in range(40):
            module_step(i)
        torch.cuda.synchronize()
        reps.append((time.perf_counter() - t0) / 40 * 1e6)
    out["fused_module_autograd_eager_us"] = sorted(reps)[len(reps) // 2]
    return out


def cpu_baseline(kind, B, L, F, reps=5):
    """The reference-equivalent CPU path (materialising torch port) on this box's cores."""
    from oracle import materialized_torch as M
    ncpu = os.cpu_count() or 1
    _, relevance, n, X = synth(B, L, F, 0, "cpu")
    lin = torch.nn.Linear(F, 1)
    # The port is memory-copy bound; over-subscribing a many-core host makes it slower, so
    # report the BEST of a small thread sweep (the reference user would tune this too).
    best = None
    for threads in sorted({min(8, ncpu), min(32, ncpu), min(64, ncpu), ncpu}):
        torch.set_num_threads(threads)
        M.linear_step(kind, X, lin.weight, lin.bias, relevance, n)      # warm-up
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            M.linear_step(kind, X, lin.weight, lin.bias, relevance, n)
            ts.append(time.perf_counter() - t0)
            if ts[-1] > 3.0:
                break
        med = sorted(ts)[len(ts) // 2]
        if best is None or med < best[0]:
            best = (med, threads)
    med, cores = best
    torch.set_num_threads(cores)
    scores = torch.randn(B, L)
    M.loss_step(kind, scores, relevance, n)
    t0 = time.perf_counter()
    M.loss_step(kind, scores, relevance, n)
    loss_only = time.perf_counter() - t0
    # the scalar C oracle, one core, for information
    from oracle import ltr_oracle as O
    t0 = time.perf_counter()
    O.pairwise_loss(kind, scores.numpy(), relevance.numpy(), n.numpy())
    c_scalar = time.perf_counter() - t0
    return {
        "value": B / med, "unit": "queries/s", "cores": cores, "kind": "port",
        "sample": "median of %d full steps (Linear(%d,1)+%s fwd+bwd, B=%d, L=%d) of "
                  "oracle/materialized_torch.py; best of a torch-thread sweep on %d host CPUs = %d "
                  "threads" % (reps, F, kind, B, L, ncpu, cores),
        "ms_per_step": med * 1e3, "host_cpus": ncpu,
        "loss_only_queries_per_s": B / loss_only,
        "c_oracle_scalar_1core_queries_per_s": B / c_scalar,
    }


def lazy_dp_trial(fs, mb, batches, nbuf, dist, dev):
    """A few data-parallel lazy steps under a SHORT time budget before they are timed: the launch's reducers wait for their
    peers inside the fused kernel, which has only ever run between processes on one device (tests/test_gpu_lazy_dp.py) -- if
    anything is off (a poll gives up, the ranks' weights differ) every rank falls back to the three-launch mailbox step.
    Returns None when fine, else the reason; the weights are put back either way."""
    lib = fs.lib
    why = None
    old = lib.ltr_mailbox_set_timeout_ms(int(os.environ.get("LTR_MAILBOX_CHECK_TIMEOUT_MS", "5000")))
    fs.mailbox = mb.mbox
    try:
        for i in range(2 * nbuf):
            fs.lazy_step(batches[i % nbuf])
        fs.lazy_flush()
        torch.cuda.synchronize()
        if lib.ltr_device_status(0) != 0:
            why = "device status %d (an in-launch wait gave up)" % lib.ltr_device_status(0)
        elif not bool(torch.isfinite(fs.W).all()):
            why = "non-finite weights"
    except Exception as exc:  # pragma: no cover - depends on the runtime
        why = repr(exc)[:160]
    finally:
        if old > 0:
            lib.ltr_mailbox_set_timeout_ms(old)
    if why is None:
        wb = [None] * dist.get_world_size()
        dist.all_gather_object(wb, torch.cat([fs.W, fs.bias]).cpu().numpy().tobytes())
        if not all(x == wb[0] for x in wb):
            why = "weights differ across ranks"
    else:
        dist.all_gather_object([None] * dist.get_world_size(), b"")          # (the same collectives on every rank)
    if not mb._all_agree(1 if why is None else 0):
        why = why or "another rank's trial failed"
    try:
        torch.cuda.synchronize()
    except Exception:  # pragma: no cover
        pass
    if why is not None:
        lib.ltr_device_status(1)
        fs.mailbox = None
    fs.reset_weights()
    return why


def allreduce_probe():
    """Internal (--allreduce-probe): one rank, RCCL process group forced; times the eager
    all-reduce of the (F+3)-float bucket and the step with it.  Printed as one JSON line."""
    import torch.distributed as dist
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 300))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    B, L, F, kind = WORKLOADS["c2"]
    batches = make_batches(B, L, F, 2, 500, dev)
    fs = FusedStep(kind, B, L, F, dev)
    for i in range(20):
        fs.step(batches[i % 2])
        dist.all_reduce(fs.flat)
    torch.cuda.synchronize()
    out = {}
    t0 = time.perf_counter()
    for i in range(500):
        dist.all_reduce(fs.flat)
    torch.cuda.synchronize()
    out["allreduce_only_us"] = (time.perf_counter() - t0) / 500 * 1e6
    for accum in (1, 8):
        t0 = time.perf_counter()
        for i in range(800):
            fs.step(batches[i % 2], accumulate=(i % accum) != 0)
            if i % accum == accum - 1:
                dist.all_reduce(fs.flat)
        torch.cuda.synchronize()
        out["step_with_allreduce_every_%d_us" % accum] = (time.perf_counter() - t0) / 800 * 1e6
    t0 = time.perf_counter()
    for i in range(800):
        fs.step(batches[i % 2])
    torch.cuda.synchronize()
    out["step_without_allreduce_us"] = (time.perf_counter() - t0) / 800 * 1e6
    # the N > 1 headline mode: one all-reduce per step, double-buffered and asynchronous
    from pytorchltr_amd.distributed import OverlappedBucketAllReduce
    red = OverlappedBucketAllReduce(F, count=B, device=dev)
    for i in range(40):
        fs.step(batches[i % 2], flat=red.acquire(i))
        red.launch(i)
    red.flush()
    torch.cuda.synchronize()
    reps = []
    for _ in range(3):
        t0 = time.perf_counter()
        for i in range(800):
            fs.step(batches[i % 2], flat=red.acquire(i))
            red.launch(i)
        red.flush()
        torch.cuda.synchronize()
        reps.append((time.perf_counter() - t0) / 800 * 1e6)
    out["step_with_overlapped_allreduce_every_1_us"] = sorted(reps)[1]
    from pytorchltr_amd.distributed import RcclOverlap
    from pytorchltr_amd import _C
    for depth in (0, 4):
        rawd = RcclOverlap(F, count=B, device=dev, depth=depth)
        if rawd.ok:
            def dstep(i):
                b = batches[i % 2]
                rawd.step(i, fs.kind_id, 1.0, b["X"], fs.W, fs.bias, b["rel"], _C.LABEL_I64, b["n"], fs.go, B, L, F,
                          fs.lossv, fs.part)
            for i in range(40):
                dstep(i)
            rawd.flush()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(800):
                dstep(i)
            t_host = time.perf_counter() - t0
            rawd.flush()
            torch.cuda.synchronize()
            out["raw_rccl_depth%d_us" % depth] = (time.perf_counter() - t0) / 800 * 1e6
            out["raw_rccl_depth%d_host_enqueue_us" % depth] = t_host / 800 * 1e6
        rawd.close()
    raw = RcclOverlap(F, count=B, device=dev)
    out["raw_rccl_communicator"] = bool(raw.ok)
    out["raw_rccl_note"] = raw.why
    if raw.ok:
        def rstep(i):
            b = batches[i % 2]
            raw.step(i, fs.kind_id, 1.0, b["X"], fs.W, fs.bias, b["rel"], _C.LABEL_I64, b["n"], fs.go, B, L, F,
                     fs.lossv, fs.part)
        for i in range(40):
            rstep(i)
        raw.flush()
        torch.cuda.synchronize()
        reps = []
        for _ in range(3):
            t0 = time.perf_counter()
            for i in range(800):
                rstep(i)
            raw.flush()
            torch.cuda.synchronize()
            reps.append((time.perf_counter() - t0) / 800 * 1e6)
        out["step_with_raw_rccl_overlapped_allreduce_every_1_us"] = sorted(reps)[1]
        # the result is the same as without the exchange at one rank
        ref = fs.flat.clone()
        fs.step(batches[0])
        torch.cuda.synchronize()
        rstep(0)
        got = raw.result(0)
        torch.cuda.synchronize()
        out["raw_rccl_matches_local_step"] = bool(torch.equal(got[:F + 2], fs.flat[:F + 2]))
    raw.close()
    t0 = time.perf_counter()
    for i in range(500):
        w = dist.all_reduce(fs.flat, async_op=True)
    w.wait()
    torch.cuda.synchronize()
    out["allreduce_async_host_call_us"] = (time.perf_counter() - t0) / 500 * 1e6
    dist.destroy_process_group()
    print(json.dumps(out))


def note(msg):
    """Progress on stderr (LTR_BENCH_VERBOSE=1): where a multi-rank run is, should it stall."""
    if os.environ.get("LTR_BENCH_VERBOSE") == "1":
        sys.stderr.write("[bench r%s %.1fs] %s\n" % (os.environ.get("RANK", "0"), time.perf_counter(), msg))
        sys.stderr.flush()


def spawn_ranks(n):
    """`python bench.py --gpus N` run bare (no launcher): N ranks of this very command, one per GPU (RANK = LOCAL_RANK = 0 .. N-1,
    rendezvous on 127.0.0.1), rank 0's JSON line passed through.  The driver's `torch.distributed.run` launch sets WORLD_SIZE
    and never comes here."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "MASTER_ADDR": "127.0.0.1",
                    "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    line, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(line.decode())
    sys.stdout.flush()
    if any(rcs):
        raise SystemExit("bench.py --gpus %d: rank exit codes %r" % (n, rcs))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--shard", action="store_true",
                    help="strong scaling: BASELINE's global batch split over the ranks (default: weak, B per rank)")
    ap.add_argument("--accum", type=int, default=1,
                    help="N > 1: micro-batch steps accumulated per gradient all-reduce (1 = one overlapped "
                         "all-reduce per step, the headline; > 1 = gradient-accumulation variant)")
    ap.add_argument("--no-graph", action="store_true", help="skip the hipGraph replay figures")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="only the headline step + roofline")
    ap.add_argument("--full-lists", action="store_true", help="n == list_len for every query in the timed step")
    ap.add_argument("--allreduce-probe", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.allreduce_probe:
        return allreduce_probe()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return spawn_ranks(args.gpus)        # run bare: one rank per GPU, started here
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        raise SystemExit("bench.py --gpus %d launched with WORLD_SIZE=%s: the launcher's rank count and --gpus must agree "
                         "(the line's n_gpus is the number of ranks that ran)" % (args.gpus, os.environ.get("WORLD_SIZE")))

    # stdout carries exactly ONE line, the JSON result: RCCL prints a version banner to stdout when
    # a communicator is created, and other libraries may chat too -- send file descriptor 1 to
    # stderr for the duration of the run and write the result to the saved descriptor at the end
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (no CPU fallback in the product path)")
    backend = os.environ.get("LTR_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        # (LTR_BENCH_BACKEND=gloo: the N > 1 code path run functionally on a box with fewer GPUs than ranks -- ranks share
        # devices round robin; RCCL itself wants one device per rank)
        local_rank %= torch.cuda.device_count()
    elif local_rank >= torch.cuda.device_count():
        raise SystemExit("rank %d: LOCAL_RANK %d but %d GPU(s) visible (one rank per GPU; LTR_BENCH_BACKEND=gloo shares devices "
                         "for functional runs)" % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("LTR_BENCH_FORCE_DIST") == "1":    # force: smoke-test RCCL at N=1
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # (LTR_BENCH_BACKEND=gloo with every rank on LOCAL_RANK 0: the N > 1 code path -- mailbox all-reduce, SGD step,
        # max-over-ranks timing -- run functionally by several processes on ONE GPU; tests/test_gpu_bench.py)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    # n_gpus = the ranks that actually run (and the devices they sit on, for the line)
    n_gpus = dist.get_world_size() if dist is not None else 1
    devices = ["%s:%d %s" % (os.uname().nodename, local_rank, torch.cuda.get_device_name(dev))]
    if dist is not None and n_gpus > 1:
        gathered = [None] * n_gpus
        dist.all_gather_object(gathered, devices[0])
        devices = gathered
    if dist is not None:
        # ProcessGroupNCCL's watchdog thread polls HIP events; an event query while another
        # thread is capturing aborts the process ("operation not permitted when stream is
        # capturing").  With a process group alive everything is launched eagerly.
        global GRAPHS_OK
        GRAPHS_OK = False

    def barrier():
        if dist is not None:
            dist.barrier()

    def reduce_max(dt):
        if dist is None:
            return dt
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    from pytorchltr_amd import _C
    _C.lib()                                            # fail loudly if the extension is missing

    Bg, L, F, kind = WORKLOADS[args.workload]
    if args.shard:
        if Bg % n_gpus != 0:
            raise SystemExit("--shard: global batch %d is not divisible by %d ranks" % (Bg, n_gpus))
        B = Bg // n_gpus
    else:
        B = Bg
    nbuf = nbuf_for(B, L, F)
    batches = make_batches(B, L, F, nbuf, 1000 * rank, dev, full=args.full_lists)
    fs = FusedStep(kind, B, L, F, dev, n_gpus=n_gpus)
    accum = max(1, args.accum) if dist is not None else 1

    # ---- the timed step: fused scorer + loss forward, backward to dW/db, eager, through the C ABI ----
    red = None
    allreduce_impl = None
    finish = None
    lazy_headline = dist is None and os.environ.get("LTR_BENCH_EAGER_STEP") != "1" and hasattr(_C.lib(), "ltr_linear_sgd_lazy_step_f32")
    if lazy_headline:
        # one C-ABI call and ONE launch per step: the update of step k rides in step k + 1's launch (bit-identical weights:
        # tests/test_gpu_step.py); every timed region ends with the flush of its last update
        def step(i):
            fs.lazy_step(batches[i % nbuf])

        def finish():
            fs.lazy_flush()
    elif dist is None:
        def step(i):
            fs.sgd_step(batches[i % nbuf])
    elif accum == 1:
        # ONE gradient all-reduce per step (synchronous SGD: step i + 1 scores with the weights step i's all-reduced gradient
        # produced -- the collective is on the critical path).  LTR_BENCH_ALLREDUCE selects how it is carried:
        #   "lazy" (default)  ltr_linear_sgd_lazy_step_dp_f32: ONE launch per step and rank, as at N = 1 -- the reducer
        #                     workgroups in front of the fused launch all-reduce their column sums through the peers' HIP-IPC
        #                     mapped mailboxes, the first query's workgroup writes the weights;
        #   "mailbox"         ltr_linear_sgd_step_f32 + mailbox handle: kernel, reduction, mailbox all-reduce + update (3 launches);
        #   "instream"        ... + RCCL handle: kernel, reduction, ncclAllReduce on the step's stream, update (4 launches);
        #   "overlap" / "c10d" RCCL on a side stream (delayed gradient) / torch.distributed per step.
        # Each falls back to the next on EVERY rank when it cannot be set up or its check fails; the line says which ran,
        # and extra carries the others next to it (below).
        from pytorchltr_amd.distributed import MailboxOverlap, OverlappedBucketAllReduce, RcclOverlap
        mode = os.environ.get("LTR_BENCH_ALLREDUCE", "lazy")
        raw = None
        fell_back = []
        if mode in ("lazy", "mailbox"):
            raw = MailboxOverlap(F, count=B, device=dev)
            if not raw.ok:
                fell_back.append("mailbox all-reduce unavailable (%s)" % raw.why)
                raw.close()
                raw, mode = None, "instream"
        if raw is not None and mode == "lazy":
            # (ranks sharing ONE device -- functional runs only -- must be resident side by side: a launch whose reducers wait for a
            # peer whose launch cannot become resident never ends.  One rank per GPU, as the driver launches, is never affected.)
            shared = max(devices.count(d) for d in devices)
            if shared > 1 and shared * (B + (F + 4) // 4 + 1) > 4 * CU_COUNT:
                fell_back.append("%d ranks share a device and their launches do not fit it together" % shared)
                mode = "mailbox"
        if raw is not None and mode == "lazy":
            note("mailbox up; trial of the data-parallel lazy step")
            why = lazy_dp_trial(fs, raw, batches, nbuf, dist, dev)
            note("trial: %s" % (why or "ok"))
            if why is not None:
                fell_back.append("data-parallel lazy step: %s" % why)
                mode = "mailbox"
        if raw is None and mode != "c10d":
            raw = RcclOverlap(F, count=B, device=dev, depth=(0 if mode == "instream" else 2))
        if fell_back:
            sys.stderr.write("[bench] rank %d: %s -> %s\n" % (rank, "; ".join(fell_back), mode))
        if raw is not None and raw.ok:
            red = raw
            if mode == "lazy":
                allreduce_impl = ("ltr_linear_sgd_lazy_step_dp_f32: ONE launch per step and rank -- the reducer workgroups in front of "
                                  "the fused launch sum the previous step's gradient rows, post their column sums as tagged granules "
                                  "into every peer's HIP-IPC-mapped mailbox, add the arrivals in rank order and hand the new weights "
                                  "to the launch (no collective library, no extra launch in the step)")
                fs.mailbox = raw.mbox

                def step(i):
                    fs.lazy_step(batches[i % nbuf])

                def finish():
                    fs.lazy_flush()
            else:
                if mode == "mailbox":
                    allreduce_impl = ("ltr_linear_sgd_step_f32 + mailbox all-reduce: one kernel per rank stores the bucket's tagged "
                                      "granules into every peer's HIP-IPC-mapped mailbox, adds the arrivals in rank order and applies "
                                      "the weight update (3 launches per step, no collective library in the step)")
                else:
                    allreduce_impl = ("ltr_linear_step_f32 + %s handle: ncclAllReduce %s (RCCL communicator of its own)" % (
                        ("in-stream", "on the step's stream behind its kernels") if mode == "instream" else
                        ("overlap", "on a side stream under the next step")))

                def step(i):
                    b = batches[i % nbuf]
                    if mode in ("instream", "mailbox"):      # synchronous SGD: kernels, all-reduce, W -= lr * dW -- one C-ABI call
                        raw.sgd_step(fs.kind_id, 1.0, b["X"], fs.W, fs.bias, b["rel"], _C.LABEL_I64, b["n"], fs.go, B, L, F,
                                     SGD_LR, fs.lossv, fs.part)
                    else:
                        raw.step(i, fs.kind_id, 1.0, b["X"], fs.W, fs.bias, b["rel"], _C.LABEL_I64, b["n"], fs.go, B, L, F,
                                 fs.lossv, fs.part)
        else:
            if raw is not None:
                sys.stderr.write("[bench] raw RCCL communicator unavailable (%s): torch.distributed all_reduce\n" % raw.why)
                fell_back.append("raw RCCL communicator unavailable (%s)" % raw.why)
                raw.close()
            mode = "c10d"
            red = OverlappedBucketAllReduce(F, count=B, device=dev, depth=1)
            allreduce_impl = "torch.distributed all_reduce per step (blocking)"

            def step(i):
                flat = red.acquire(i)
                fs.step(batches[i % nbuf], flat=flat)
                red.launch(i)
                g = red.result(i)
                fs.W.add_(g[:F], alpha=-SGD_LR)           # the optimiser step behind the collective
                fs.bias.add_(g[F:F + 1], alpha=-SGD_LR)
    else:
        ar_view = fs.flat[:F + 2]                # (the count slot is not all-reduced again and again)
        fs.flat[F + 2] = float(B * n_gpus)

        def step(i):
            j = i % accum
            fs.step(batches[i % nbuf], accumulate=(j != 0))
            if j == accum - 1:
                dist.all_reduce(ar_view, op=dist.ReduceOp.SUM)

    note("warm-up")
    for i in range(max(args.warmup, 2 * nbuf)):
        step(i)
    if finish is not None:
        finish()
    torch.cuda.synchronize()
    note("estimating the step")
    # enough steps for a >= 50 ms timed region (aim at 60), a multiple of the all-reduce period
    est_steps = 25 * accum
    ests = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(est_steps):
            step(i)
        if finish is not None:
            finish()
        torch.cuda.synchronize()
        ests.append((time.perf_counter() - t0) / est_steps)
    est = min(ests)
    steps_timed = max(args.steps, int(math.ceil(0.06 / max(est, 1e-7))))
    if dist is not None:
        t = torch.tensor([steps_timed], dtype=torch.int64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        steps_timed = int(t.item())
    steps_timed = ((steps_timed + accum - 1) // accum) * accum
    note("timing %d steps x 5" % steps_timed)
    regions = time_region(step, steps_timed, barrier, repeats=5, reduce_max=reduce_max, finish=finish)
    note("timed")
    elapsed = median(regions)
    value = n_gpus * B * steps_timed / elapsed

    # N > 1: every rank must hold the same weights after the same steps (synchronous SGD), and the other ways of carrying the
    # step's all-reduce are timed next to the headline's -- the mailbox step as three launches and the in-stream ncclAllReduce
    # (RCCL over xGMI: north_star's wording) -- so that one hardware run yields them all
    dp_extra = None
    if dist is not None and accum == 1:
        dp_extra = {"allreduce_mode": mode, "fell_back": fell_back}
        torch.cuda.synchronize()
        wb = [None] * n_gpus
        dist.all_gather_object(wb, torch.cat([fs.W, fs.bias]).cpu().numpy().tobytes())
        dp_extra["weights_bit_identical_across_ranks"] = all(x == wb[0] for x in wb)
        dp_extra["device_status"] = int(fs.lib.ltr_device_status(0))
        nalt = max(20, steps_timed // 4)

        def time_alt(fn, fin=None):
            for i in range(2 * nbuf):
                fn(i)
            if fin is not None:
                fin()
            r = time_region(fn, nalt, barrier, repeats=3, reduce_max=reduce_max, finish=fin)
            return median(r) / nalt * 1e6
        if mode == "lazy":
            dp_extra["lazy_one_launch_step_us"] = elapsed / steps_timed * 1e6

            def mstep(i):
                b = batches[i % nbuf]
                red.sgd_step(fs.kind_id, 1.0, b["X"], fs.W, fs.bias, b["rel"], _C.LABEL_I64, b["n"], fs.go, B, L, F, SGD_LR, fs.lossv, fs.part)
            dp_extra["mailbox_three_launch_step_us"] = time_alt(mstep)
        if mode != "instream" and backend == "nccl" and n_gpus > 1:
            rc_ = RcclOverlap(F, count=B, device=dev, depth=0)
            if rc_.ok:
                def rstep(i):
                    b = batches[i % nbuf]
                    rc_.sgd_step(fs.kind_id, 1.0, b["X"], fs.W, fs.bias, b["rel"], _C.LABEL_I64, b["n"], fs.go, B, L, F, SGD_LR, fs.lossv, fs.part)
                dp_extra["rccl_instream_step_us"] = time_alt(rstep)
                dp_extra["rccl_instream_what"] = ("ltr_linear_sgd_step_f32 + in-stream RCCL handle: kernel, reduction, ncclAllReduce on the "
                                                  "step's stream, update (4 launches)")
                rc_.flush()
            else:
                dp_extra["rccl_instream_step_us"] = None
                dp_extra["rccl_instream_what"] = "raw RCCL communicator unavailable: %s" % rc_.why
            rc_.close()
        elif mode == "instream":
            dp_extra["rccl_instream_step_us"] = elapsed / steps_timed * 1e6
    if red is not None:
        red.flush()
        if hasattr(red, "close"):
            red.close()
        fs.mailbox = None

    # N > 1, headline mode: the gradient-accumulation variant (one all-reduce per 8 steps) next to it
    accum8 = None
    if dist is not None and accum == 1:
        ar_view = fs.flat[:F + 2]
        fs.flat[F + 2] = float(B * n_gpus)

        def step8(i):
            j = i % 8
            fs.step(batches[i % nbuf], accumulate=(j != 0))
            if j == 7:
                dist.all_reduce(ar_view, op=dist.ReduceOp.SUM)
        n8 = max(8, (steps_timed // 2 // 8) * 8)
        for i in range(16):
            step8(i)
        r8 = time_region(step8, n8, barrier, repeats=3, reduce_max=reduce_max)
        accum8 = {"queries_per_s": n_gpus * B * n8 / median(r8), "ms_per_step": median(r8) / n8 * 1e3,
                  "allreduce_every": 8, "what": "gradient accumulation: 8 micro-batch steps per all-reduce"}

    # the same step without the weight update (what rounds 1-3 reported as `value`)
    noupd = None
    if dist is None:
        for i in range(2 * nbuf):
            fs.step(batches[i % nbuf])
        r0 = time_region(lambda i: fs.step(batches[i % nbuf]), max(1, steps_timed // 2), barrier, repeats=3, reduce_max=reduce_max)
        noupd = {"queries_per_s": n_gpus * B * max(1, steps_timed // 2) / median(r0),
                 "ms_per_step": median(r0) / max(1, steps_timed // 2) * 1e3,
                 "what": "ltr_linear_step_f32: the same two launches, W left alone (rounds 1-3 reported this)"}

    # the step as rounds 4-5 timed it: two launches per step (ltr_linear_sgd_step_f32: kernel, then reduction + update)
    eager2 = None
    if lazy_headline:
        for i in range(2 * nbuf):
            fs.sgd_step(batches[i % nbuf])
        r1 = time_region(lambda i: fs.sgd_step(batches[i % nbuf]), max(1, steps_timed // 2), barrier, repeats=3, reduce_max=reduce_max)
        eager2 = {"queries_per_s": n_gpus * B * max(1, steps_timed // 2) / median(r1),
                  "ms_per_step": median(r1) / max(1, steps_timed // 2) * 1e3,
                  "what": "ltr_linear_sgd_step_f32: the same steps as two launches each (kernel; reduction + update) -- the "
                          "headline of rounds 4-5; same weights bit for bit"}

    out = None
    if rank == 0:
        extra = {"timed_regions_s": regions, "steps_per_region": steps_timed}
        if noupd is not None:
            extra["step_without_weight_update"] = noupd
        if eager2 is not None:
            extra["two_launch_step"] = eager2
        _ = pmc_record(args.workload)
        extra["pmc_counters"] = ("profiles/pmc_<workload>.json (separate rocprofv3 --pmc passes); refused when the kernel "
                                 "sources changed since: stale for %s" % (sorted(PMC_STALE) or "none"))
        if accum8 is not None:
            extra["gradient_accumulation_8"] = accum8
        if dp_extra is not None:
            extra["data_parallel"] = dp_extra
            for k_ in ("rccl_instream_step_us", "mailbox_three_launch_step_us", "lazy_one_launch_step_us"):
                if k_ in dp_extra:
                    extra[k_] = dp_extra[k_]
        # ---- roofline of the dominant kernel: fused scorer+loss, cold, timed live with HIP events ----
        k_us, graphed = time_launches(lambda i: fs.kernel(batches[i % nbuf]), nbuf)
        k_evt = time_events(lambda i: fs.kernel(batches[i % nbuf]), 100)
        moved = moved_bytes(batches, B, L, F)
        padded = padded_bytes(B, L, F)
        pmc = pmc_record(args.workload).get(fs.plan, {}) if not args.full_lists else {}
        roofline = {
            "bound": "hbm", "achieved": moved / (k_us * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": moved / (k_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
            "traffic": pmc.get("hbm_bytes_per_launch"),
            "basis": "bytes the launch must move: the n[b] real rows of each query (features + int64 "
                     "labels), n, W, loss and the (F+1) partials per query; %d batches in rotation "
                     "(%.0f MB of features > 256 MiB Infinity Cache), so every launch streams from HBM"
                     % (nbuf, nbuf * 4 * B * L * F / 1e6),
            "kernel": "%s<%s> (via ltr_linear_partials_f32)" % (fs.plan, kind),
            "kernel_us_avg": k_us,
            "timing": "HIP events around %s back-to-back launches over %d rotating batches" % (
                "hipGraph-replayed" if graphed else "eager", nbuf),
            "kernel_us_single_launch_event_pair": {"avg": k_evt[0], "median": k_evt[1], "min": k_evt[2]},
            "moved_bytes_per_launch": moved,
            "padded_formula": {"bytes_per_launch": padded, "achieved": padded / (k_us * 1e-6) / 1e9,
                               "frac": padded / (k_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                               "note": "SURVEY.md 8(d) bytes (padded rows counted although never read)"},
            "valu_issue_frac": valu_frac(pmc.get("valu_insts_per_launch"), k_us),
        }
        if lazy_headline:
            # the launch of the timed step is this kernel WITH the previous step's reduction and update in it: its own duration,
            # live, from event pairs the command processor records at the kernel's begin and end (ltr_debug_kernel_events ->
            # hipExtLaunchKernelGGL), over back-to-back lazy steps on the rotating batches
            lz = lazy_kernel_us(fs, batches, nbuf)
            if lz is not None:
                roofline["plain_kernel"] = {"kernel_us_avg": k_us, "frac": roofline["frac"], "achieved": roofline["achieved"],
                                            "what": "the same kernel launched through ltr_linear_partials_f32 (no pending update in the "
                                                    "launch): what rounds 1-5 reported here; " + roofline["timing"]}
                roofline["kernel"] = "%s<%s> (via ltr_linear_sgd_lazy_step_f32: + the previous step's reduction and update)" % (fs.plan, kind)
                roofline["kernel_us_avg"] = lz["avg"]
                roofline["timing"] = ("start / stop events recorded at the kernel's begin and end (hipExtLaunchKernelGGL) of %d back-to-back "
                                      "eager lazy steps over %d rotating batches" % (lz["count"], nbuf))
                roofline["kernel_us_median_min"] = [lz["median"], lz["min"]]
                roofline["achieved"] = moved / (lz["avg"] * 1e-6) / 1e9
                roofline["frac"] = roofline["achieved"] / HBM_PEAK_GBS
        # the same bytes against the whole STEP (kernel + reduction launch + update): what HBM delivers per step
        step_s = elapsed / steps_timed
        roofline["step_level"] = {"us_per_step": step_s * 1e6, "frac": moved / step_s / 1e9 / HBM_PEAK_GBS,
                                  "note": "must-move bytes / whole-step time / 8 TB/s (the per-kernel frac leaves the reduction "
                                          "launch and the launch boundaries out)"}
        if lazy_headline and "plain_kernel" in roofline:
            # ONE launch per step: the step's wall time bounds the launch's duration from above on every clock -- rocprofv3's
            # kernel trace reads 0.3-0.4 us more per launch than the command processor's event pairs (profiles/README.md), so the
            # headline fraction is taken on the clock that can only under-state it, and the event-pair figure is kept next to it
            roofline["frac_event_pairs"] = roofline["frac"]
            roofline["achieved_event_pairs"] = roofline["achieved"]
            roofline["kernel_us_event_pairs"] = roofline["kernel_us_avg"]
            roofline["kernel_us_avg"] = step_s * 1e6
            roofline["achieved"] = moved / step_s / 1e9
            roofline["frac"] = roofline["achieved"] / HBM_PEAK_GBS
            roofline["timing"] += ("; `kernel_us_avg` / `achieved` / `frac` are taken on the STEP's wall time (one launch per step: an "
                                   "upper bound of the launch's duration on every clock, rocprofv3's included), the event-pair "
                                   "figures are kernel_us_event_pairs / frac_event_pairs")
        # the HBM rate this box reaches on a plain device copy (read + write of 1 GiB, hipMemcpy DtoD through torch), for scale
        try:
            src_ = torch.empty(256 * 2 ** 20, dtype=torch.float32, device=dev).normal_()
            dst_ = torch.empty_like(src_)
            for _ in range(3):
                dst_.copy_(src_)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                dst_.copy_(src_)
            e1.record()
            torch.cuda.synchronize()
            copy_gbs = 10 * 2 * src_.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
            roofline["peak_measured"] = copy_gbs
            roofline["frac_of_measured"] = roofline["achieved"] / copy_gbs
            roofline["peak_measured_what"] = "device-to-device copy of 1 GiB (read + write bytes / time), this box, this run"
            del src_, dst_
            torch.cuda.empty_cache()
        except Exception as exc:  # pragma: no cover
            roofline["peak_measured"] = None
            roofline["peak_measured_what"] = repr(exc)[:120]
        # what ONE ROUND of workgroups can stream at this batch size: the same grid, workgroup size and
        # ragged spans with nothing behind the loads (ltr_debug_stream_probe_f32)
        if fs.plan == "linear_regtile_kernel" and L * (F // 4) <= 19 * 512 and F % 4 == 0:
            pout = torch.empty(B * 8, device=dev)

            def probe(i):
                bt = batches[i % nbuf]
                _C.check(_C.lib().ltr_debug_stream_probe_f32(bt["X"].data_ptr(), bt["n"].data_ptr(), B, L, F,
                                                             pout.data_ptr(), torch.cuda.current_stream().cuda_stream))
            for i in range(2 * nbuf):
                probe(i)
            p_us, _ = time_launches(probe, nbuf)
            rows = sum(b_["rows"] for b_ in batches) / float(len(batches))
            pbytes = rows * 4 * F + B * 8 + B * 32
            roofline["launch_ceiling"] = {
                "us": p_us, "bytes_per_launch": pbytes, "GBs": pbytes / (p_us * 1e-6) / 1e9,
                "frac_of_peak": pbytes / (p_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                "kernel_over_ceiling": k_us / p_us,
                "what": "stream_probe_kernel: same grid (one 512-thread workgroup per query), same ragged spans, "
                        "16-byte buffer loads all in flight, one dword store per wave -- no scores, no pair pass, "
                        "no dW; cold, same rotating batches, same timing method"}
        if n_gpus == 1 and not args.full_lists and not args.no_extra:
            full_res, _, _ = measure_config(args.workload, B, L, F, kind, dev, 2000, full=True)
            roofline["full_lists"] = {k: full_res[k] for k in (
                "workload", "kernel_us", "step_us", "moved_bytes_per_launch", "moved_GBs", "frac_moved",
                "batches_in_rotation", "timing")}
            roofline["full_lists"]["frac"] = full_res["frac_moved"]

        if not args.no_graph and dist is None:
            rp = try_graph(lambda: [fs.step(batches[i]) for i in range(nbuf)])
            if rp is not None:
                rg = time_region(lambda i: rp(), max(1, steps_timed // nbuf), lambda: None, repeats=3)
                extra["hipgraph_replay_queries_per_s"] = B * nbuf * max(1, steps_timed // nbuf) / median(rg)

        if n_gpus == 1:
            extra["loss_kernel"] = loss_kernel_extra(args.workload, kind, B, L, dev, batches)
        if n_gpus == 1 and not args.no_extra:
            extra.update(n1_extras(args, kind, B, L, F, dev, batches, steps_timed))
        cpu = None
        if n_gpus == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(kind, B, L, F)

        out = {
            "metric": "queries/sec loss fwd+bwd (B=%d, list_len=%d) + achieved HBM GB/s" % (Bg, L),
            "value": value, "unit": "queries/s", "n_gpus": n_gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / steps_timed * 1e3,
            "higher_is_better": True, "scaling": "strong" if args.shard else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: Linear(%d,1) scorer + %s loss fwd+bwd, B=%d/GPU, "
                                   "list_len=%d, n~U[1,%d]%s" % (args.workload, F, kind, B, L, L,
                                                                " (full lists)" if args.full_lists else ""),
                       "global_batch": n_gpus * B, "list_len": L, "features": F, "loss": kind,
                       "mode": "eager", "parallelism": "dp%d" % n_gpus, "devices": devices,
                       "step": (("synchronous SGD step in one C-ABI call AND ONE LAUNCH (ltr_linear_sgd_lazy_step_f32): fused scorer + "
                                 "loss kernel whose first workgroups sum the PREVIOUS step's per-query gradient rows into [dW | db | "
                                 "loss_sum], apply W -= lr * dW (lr %g) and hand the new weights to every workgroup before the dot "
                                 "products; each timed region ends with ltr_linear_sgd_flush_f32 (the last step's update): as many "
                                 "updates as steps, weights bit-identical to the two-launch step (extra.two_launch_step) -- "
                                 "examples/01-basic-usage.py:66-75" % SGD_LR) if lazy_headline else
                                ("data-parallel synchronous SGD step in one C-ABI call AND ONE LAUNCH per rank "
                                 "(ltr_linear_sgd_lazy_step_dp_f32): as at N = 1, and the reducer workgroups all-reduce their column "
                                 "sums over the ranks inside the launch (lr %g); each timed region ends with "
                                 "ltr_linear_sgd_flush_dp_f32 -- examples/01-basic-usage.py:66-75, sharded" % SGD_LR)
                                if (dist is not None and accum == 1 and mode == "lazy") else
                                ("synchronous SGD step in one C-ABI call (ltr_linear_sgd_step_f32): fused scorer + loss "
                                "kernel, cross-query reduction into [dW | db | loss_sum]%s, W -= lr * dW (lr %g) -- "
                                "examples/01-basic-usage.py:66-75" % (
                                    ", the step's gradient all-reduce" if dist is not None else
                                    " with the update riding in it", SGD_LR))) if accum == 1 else
                               "gradient accumulation (no weight update in the timed region)",
                       "batches_in_rotation": nbuf, "steps_timed": steps_timed, "repeats": 5,
                       "statistic": "median of 5 timed regions",
                       "allreduce_every": accum if dist is not None else None,
                       "allreduce": (None if dist is None else
                                     ("one per step: " + str(allreduce_impl) if accum == 1 else
                                      "blocking, once per %d accumulated steps" % accum))},
            "roofline": roofline, "cpu_baseline": cpu, "extra": extra,
        }
    barrier()
    if dist is not None:
        dist.destroy_process_group()
    sys.stdout.flush()
    if out is not None:
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    os.close(result_fd)


def loss_kernel_extra(workload, kind, B, L, dev, batches):
    """The loss kernel alone (the literal "loss fwd+bwd" on precomputed scores): VALU-issue bound
    (DESIGN.md section 4), so both an HBM and a VALU-issue fraction are reported."""
    from pytorchltr_amd import _C
    lib = _C.lib()
    nbuf = len(batches)
    dsc = torch.empty(B, L, device=dev)
    lossv = torch.empty(B, device=dev)
    kid = getattr(_C, kind.upper())

    def launch_loss(i):
        b = batches[i % nbuf]
        _C.check(lib.ltr_pairwise_loss_f32(kid, 1.0, b["scores"].data_ptr(), b["rel"].data_ptr(), _C.LABEL_I64,
                                           b["n"].data_ptr(), B, L, lossv.data_ptr(), dsc.data_ptr(),
                                           torch.cuda.current_stream().cuda_stream))
    for i in range(10):
        launch_loss(i)
    l_us, _ = time_launches(launch_loss, nbuf)
    loss_bytes = B * (16 * L + 16)
    pmc = pmc_record(workload).get("pairwise_loss_kernel", {})
    return {"kernel": "pairwise_loss_kernel<%s>" % kind, "us_avg": l_us, "bound": "valu",
            "algorithmic_bytes_per_launch": loss_bytes,
            "achieved_GBs": loss_bytes / (l_us * 1e-6) / 1e9,
            "frac_of_hbm_peak": loss_bytes / (l_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
            "valu_issue_frac": valu_frac(pmc.get("valu_insts_per_launch"), l_us),
            "valu_basis": "SQ_INSTS_VALU per launch (profiles/pmc_%s.json) / (512 wave-instructions per "
                          "cycle x 2.4 GHz x duration)" % workload}


def n1_extras(args, kind, B, L, F, dev, batches, steps_timed):
    """Everything else the N=1 line carries: the reference-signature (drop-in) paths, the loss-only
    kernel, the other BASELINE configs, the MLP scorer, the forced-dist all-reduce probe."""
    from pytorchltr_amd import _C
    from pytorchltr_amd import loss as L_
    from pytorchltr_amd.evaluation import ndcg
    from pytorchltr_amd.fused import FusedLinearLoss, LinearScorer
    lib = _C.lib()
    extra = {}
    nbuf = len(batches)
    b0 = batches[0]
    loss_cls = {"hinge": L_.PairwiseHingeLoss, "dcg_hinge": L_.PairwiseDCGHingeLoss,
                "logistic": L_.PairwiseLogisticLoss, "arp1": L_.LambdaARPLoss1,
                "arp2": L_.LambdaARPLoss2, "ndcg1": L_.LambdaNDCGLoss1,
                "ndcg2": L_.LambdaNDCGLoss2}[kind]
    loss_fn = loss_cls()
    reps = max(200, min(steps_timed, 2000))

    def measure(stepfn):
        for _ in range(20):
            stepfn()
        res = {"eager_queries_per_s": B * reps / median(time_region(lambda i: stepfn(), reps, lambda: None, repeats=3))}
        res["eager_us_per_step"] = B / res["eager_queries_per_s"] * 1e6
        rp = None if args.no_graph else try_graph(stepfn)
        if rp is not None:
            res["hipgraph_queries_per_s"] = B * reps / median(time_region(lambda i: rp(), reps, lambda: None, repeats=3))
        return res

    # (a) the fused op as an autograd module: FusedLinearLoss(...)(xs, ys, n).mean().backward()
    fused = FusedLinearLoss(F, kind).to(dev)
    params = [fused.weight, fused.bias]

    def module_step():
        for p_ in params:
            p_.grad = None
        fused(b0["X"], b0["rel"], b0["n"]).mean().backward()
    extra["fused_module_autograd"] = measure(module_step)

    # (b) the reference's own user code, unfused drop-in: torch Linear + our loss module
    lin = torch.nn.Linear(F, 1).to(dev)

    def dropin_step():
        lin.weight.grad = None
        lin.bias.grad = None
        loss_fn(lin(b0["X"]), b0["rel"], b0["n"]).mean().backward()
    extra["dropin_linear_plus_loss_module"] = measure(dropin_step)

    # (b2) the same user code after the one-liner `model = use_linear_scorer(model)`: the nn.Linear(F, 1)
    # scorer swapped for the package's streaming scorer, parameters shared
    from pytorchltr_amd.fused import use_linear_scorer
    scorer = use_linear_scorer(torch.nn.Linear(F, 1).to(dev))
    assert isinstance(scorer, LinearScorer)

    def dropin_scorer_step():
        scorer.weight.grad = None
        scorer.bias.grad = None
        loss_fn(scorer(b0["X"]), b0["rel"], b0["n"]).mean().backward()
    extra["dropin_use_linear_scorer_plus_loss_module"] = measure(dropin_scorer_step)

    # (b2') the reference's WHOLE loop body as the user writes it -- loss = loss_fn(model(xs), ys, n).mean(); optimizer.zero_grad();
    # loss.backward(); optimizer.step() -- with the two changed lines: model = use_linear_scorer(model) and
    # pytorchltr_amd.optim.SGD for torch.optim.SGD (one kernel launch per step for the scorer: the lazy step; + .mean() and its
    # backward, which the user wrote), next to the same loop with torch.optim.SGD
    try:
        from pytorchltr_amd.optim import SGD as LazyOptSGD
        for name_, make_opt in (("dropin_loop_lazy_sgd", lambda ps: LazyOptSGD(ps, lr=1e-6)),
                                ("dropin_loop_torch_sgd", lambda ps: torch.optim.SGD(ps, lr=1e-6))):
            m_ = use_linear_scorer(torch.nn.Linear(F, 1).to(dev))
            o_ = make_opt(m_.parameters())

            def loop_step():
                bt = batches[loop_step.i % nbuf]
                loop_step.i += 1
                loss_ = loss_fn(m_(bt["X"]), bt["rel"], bt["n"]).mean()
                o_.zero_grad()
                loss_.backward()
                o_.step()
            loop_step.i = 0
            for _ in range(20):
                loop_step()
            t_ = median(time_region(lambda i: loop_step(), reps, lambda: None, repeats=3)) / reps
            extra[name_] = {"eager_us_per_step": t_ * 1e6, "eager_queries_per_s": B / t_,
                            "what": "examples/01-basic-usage.py:70-75 verbatim, rotating cold batches, optimizer step included"}
            if hasattr(o_, "flush"):
                o_.flush()
            del m_, o_
    except Exception as exc:  # pragma: no cover
        extra["dropin_loop_lazy_sgd"] = {"error": repr(exc)[:200]}

    # (b3) the whole training step of examples/01-basic-usage.py:66-75 as the user writes it -- model = use_linear_scorer(
    # nn.Linear(F, 1)), loss_fn(model(xs), ys, n).mean().backward(), torch.optim.SGD step -- captured ONCE by
    # pytorchltr_amd.GraphedStep and replayed per batch (the eager figures above vary 2x between
    # boxes, the replayed step does not).  Timed with the copy of each batch into the static buffers left out
    # (graph replay only: the batch is already on the device) and with it (step(xs, ys, n) as documented).
    if not args.no_graph:
        try:
            from pytorchltr_amd.graphed import GraphedStep
            gmodel = use_linear_scorer(torch.nn.Linear(F, 1).to(dev))
            from pytorchltr_amd.optim import SGD as LazyOptSGD_
            gopt = LazyOptSGD_(gmodel.parameters(), lr=1e-6)
            gstep = GraphedStep(gmodel, gopt, lambda xs, ys, n_: loss_fn(gmodel(xs), ys, n_).mean(),
                                example_batch=(b0["X"], b0["rel"], b0["n"]))
            for _ in range(20):
                gstep._graph.replay()
            t_replay = median(time_region(lambda i: gstep._graph.replay(), reps, lambda: None, repeats=3)) / reps
            for _ in range(5):
                gstep(b0["X"], b0["rel"], b0["n"])
            t_call = median(time_region(lambda i: gstep(batches[i % nbuf]["X"], batches[i % nbuf]["rel"], batches[i % nbuf]["n"]),
                                        max(50, reps // 4), lambda: None, repeats=3)) / max(50, reps // 4)
            extra["graphed_step_dropin_sgd"] = {
                "what": "GraphedStep(use_linear_scorer(nn.Linear(F,1)), pytorchltr_amd.optim.SGD, loss_fn(model(xs), ys, n).mean()): forward "
                        "+ backward + optimizer.step() replayed as one hipGraph (the previous step's reduction + update, the fused kernel, "
                        ".mean() and its backward)",
                "replay_us_per_step": t_replay * 1e6, "replay_queries_per_s": B / t_replay,
                "call_with_batch_copy_us_per_step": t_call * 1e6, "call_with_batch_copy_queries_per_s": B / t_call,
                "batch_copy_bytes": int(b0["X"].numel() * 4 + b0["rel"].numel() * b0["rel"].element_size() + b0["n"].numel() * 8)}
        except Exception as exc:  # pragma: no cover - depends on the runtime's capture support
            extra["graphed_step_dropin_sgd"] = {"error": repr(exc)[:200]}

    # (c) loss only (the literal "loss fwd+bwd" on precomputed scores), reference call signature
    sc = b0["scores"].clone().requires_grad_(True)

    # the floor of PyTorch's own eager autograd on this host for a graph of the same depth: a
    # NATIVE elementwise op in place of the loss (no custom Function at all)
    def native_step():
        sc.grad = None
        (sc * 2.0).sum(1).mean().backward()
    extra["eager_floor_native_ops"] = dict(measure(native_step),
                                           what="(scores * 2).sum(1).mean().backward(): torch ops only, same graph depth")

    def loss_step():
        sc.grad = None
        loss_fn(sc, b0["rel"], b0["n"]).mean().backward()
    extra["loss_only_module"] = measure(loss_step)

    # ---- the other BASELINE configs, cold, same method (configs[2..4]; per-GPU shards of C4 / C5) ----
    cfgs = {}
    todo = [("c3", WORKLOADS["c3"], None), ("c4", WORKLOADS["c4"], None), ("c4_shard8", (32, 1000, 220, "dcg_hinge"), "c4"),
            ("c5", WORKLOADS["c5"], None), ("c5_shard8", (64, 512, 700, "hinge"), "c5")]
    for name, (b_, l_, f_, k_), base in todo:
        if name == args.workload:
            continue
        try:
            res, fs_, bt_ = measure_config(base or name, b_, l_, f_, k_, dev, 3000,
                                           label="%s: Linear(%d,1) + %s, B=%d, list_len=%d, n~U[1,%d]%s" % (
                                               name, f_, k_, b_, l_, l_,
                                               " (the per-GPU shard of %s at 8 GPUs)" % base if base else ""))
            if base:
                res["traffic"] = None
                res["valu_issue_frac"] = None
            if name == "c3":
                # the evaluation half of configs[2]: ndcg@10 on the same batch
                sc3 = bt_[0]["scores"]
                mout = torch.empty(b_, device=dev)

                def launch_ndcg(i):
                    bb = bt_[i % len(bt_)]
                    _C.check(lib.ltr_dcg_f32(bb["scores"].data_ptr(), bb["rel"].data_ptr(), _C.LABEL_I64,
                                             bb["n"].data_ptr(), b_, l_, 10, 1, 1, mout.data_ptr(),
                                             torch.cuda.current_stream().cuda_stream))
                for i in range(5):
                    launch_ndcg(i)
                m_us, _ = time_launches(launch_ndcg, len(bt_))
                res["ndcg_at_10_kernel_us"] = m_us
                with torch.no_grad():
                    for _ in range(20):
                        ndcg(sc3, bt_[0]["rel"], bt_[0]["n"], k=10)
                    res["ndcg_at_10_eager_call_us"] = median(time_region(
                        lambda i: ndcg(sc3, bt_[0]["rel"], bt_[0]["n"], k=10), 200, lambda: None, repeats=3)) / 200 * 1e6
            cfgs[name] = res
            del fs_, bt_
            torch.cuda.empty_cache()
        except Exception as exc:  # pragma: no cover
            cfgs[name] = {"error": repr(exc)}
    extra["configs"] = cfgs

    # ---- shapes OFF the BASELINE grid whose plan changed in round 4 (same cold method; kernel us only): lists of 136..256
    # documents at MSLR / Istella widths (19- / 24-sweep register tiles), Yahoo-shaped short lists with 700 features (parts
    # kernel), the NDCG kinds on long lists at small batches (cluster / parts kernel with the rank exchange) ----
    off = {}
    for name, (b_, l_, f_, k_) in (("mslr_len256_hinge", (1024, 256, 136, "hinge")), ("mslr_len256_ndcg2", (1024, 256, 136, "ndcg2")),
                                   ("istella_len200_hinge", (1024, 200, 220, "hinge")), ("yahoo_len128_hinge", (2048, 128, 700, "hinge")),
                                   ("c4_shape_ndcg2", (256, 1000, 220, "ndcg2")), ("c4_shard8_ndcg2", (32, 1000, 220, "ndcg2"))):
        try:
            res, fs_, bt_ = measure_config(name, b_, l_, f_, k_, dev, 5000)
            off[name] = {"shape": [b_, l_, f_], "loss": k_, "plan": res["plan"], "kernel_us": res["kernel_us"],
                         "step_us": res["step_us"], "frac_moved": res["frac_moved"]}
            del fs_, bt_
            torch.cuda.empty_cache()
        except Exception as exc:  # pragma: no cover
            off[name] = {"error": repr(exc)}
    extra["off_grid"] = off

    mlp = mlp_extra(kind, batches, args.no_graph)
    if mlp is not None:
        extra["mlp_scorer_fused"] = mlp

    # ---- multi-GPU is unmeasured here (1-GPU box): cost of the per-step collective at one rank ----
    try:
        env = dict(os.environ)
        env.pop("LTR_BENCH_FORCE_DIST", None)
        env["MASTER_PORT"] = str(29700 + os.getpid() % 200)
        pr = subprocess.run([sys.executable, os.path.abspath(__file__), "--allreduce-probe"], env=env,
                            stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=120)
        line = [ln for ln in pr.stdout.decode().splitlines() if ln.startswith("{")]
        extra["allreduce_probe_1rank"] = json.loads(line[-1]) if line else {"error": "no output, rc=%d" % pr.returncode}
        extra["allreduce_probe_1rank"]["note"] = ("RCCL process group with ONE rank on this box; scaling over xGMI is "
                                                  "unmeasured by the builder (no multi-GPU box)")
    except Exception as exc:  # pragma: no cover
        extra["allreduce_probe_1rank"] = {"error": repr(exc)}
    return extra


if __name__ == "__main__":
    main()
