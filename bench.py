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

computed by the fused HIP path (scores -> PairwiseHingeLoss -> d/dW, d/db), inputs resident in
HBM.  N > 1: one process per GPU, every rank owns its own B queries (weak scaling; queries are
independent, no data-path collective) and the scorer gradients + (loss sum, count) are
all-reduced in one RCCL bucket per step.

Prints ONE JSON line on rank 0.  The same line carries
  roofline      the dominant kernel (fused scorer+loss) timed live with HIP events,
                algorithmic bytes / launch duration against 8 TB/s HBM3E;
  cpu_baseline  the materialising CPU port of the reference path (oracle/materialized_torch.py)
                timed on this box's host cores (rank 0, N=1 only);
  extra         loss-only drop-in numbers, eager vs hipGraph, per-kernel times.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E peak (MI355X_MICROARCH.md); ~6300 achievable

WORKLOADS = {
    # name: (B, L, F, loss kind)   -- BASELINE.json configs
    "c2": (1024, 128, 136, "hinge"),
    "c3": (1024, 128, 136, "ndcg2"),
    "c4": (256, 1000, 220, "dcg_hinge"),
    "c5": (512, 512, 700, "hinge"),
}


def synth(B, L, F, seed, device):
    """SURVEY.md section 8(d) recipe (CPU generator, then .to(device))."""
    g = torch.Generator().manual_seed(seed)
    scores = torch.randn(B, L, generator=g)
    relevance = torch.randint(0, 5, (B, L), generator=g)
    n = torch.randint(1, L + 1, (B,), generator=g)
    X = torch.randn(B, L, F, generator=g)
    return scores.to(device), relevance.to(device), n.to(device), X.to(device)


def fused_kernel_name(kind_id, B, L, F):
    """Which kernel ltr_linear_partials_f32 dispatches to for this shape (asks the library)."""
    from pytorchltr_amd import _C
    plan = _C.lib().ltr_linear_fused_plan(kind_id, B, L, F)
    return {_C.PLAN_REGISTER_TILE: "linear_regtile_kernel", _C.PLAN_CLUSTER: "linear_cluster_kernel",
            _C.PLAN_GENERAL: "linear_pairwise_kernel"}.get(plan, "?")


def time_events(fn, iters):
    """Per-launch duration (us) of fn() with ONE HIP event pair per launch, on the stream the
    kernels are launched on (torch's current stream).  Includes the event-record overhead
    (~4-5 us on this stack), so it is an upper bound; see time_launches for the figure used."""
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(iters)]
    ends = [torch.cuda.Event(enable_timing=True) for _ in range(iters)]
    torch.cuda.synchronize()
    for i in range(iters):
        starts[i].record()
        fn()
        ends[i].record()
    torch.cuda.synchronize()
    per = sorted(s.elapsed_time(e) * 1e3 for s, e in zip(starts, ends))
    return sum(per) / len(per), per[len(per) // 2], per[0]


def time_launches(fn, per_graph=20, replays=10):
    """Average launch duration (us): `per_graph` back-to-back launches of fn() captured into a
    hipGraph, replayed `replays` times between ONE HIP event pair on the launch stream; the
    average therefore contains the kernel plus its dependent-launch boundary, not host or event
    overhead.  Falls back to eager back-to-back launches if capture is unavailable."""
    def many():
        for _ in range(per_graph):
            fn()
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
    return start.elapsed_time(end) * 1e3 / (per_graph * replays), replay is not None


def time_wall(fn, steps, barrier):
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    barrier()
    return time.perf_counter() - t0


GRAPHS_OK = True     # cleared when a process group exists (see main)


def try_graph(step, warm=3):
    """Capture `step` (forward + backward) into a hipGraph; returns replay fn or None."""
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


def mlp_extra(kind, X, relevance, n, no_graph):
    """the guide's 136-50-10-1 ReLU MLP + loss + backward as one fused MFMA step
    (ltr_mlp_pairwise_f32), next to the same step written with torch layers + the HIP loss module."""
    from pytorchltr_amd import _C, fused
    from pytorchltr_amd import loss as L_
    B, L, F = X.shape
    H1, H2 = 50, 10
    if not fused.mlp_supported(L, F, H1, H2):
        return None
    dev = X.device
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

    def launch():
        _C.check(lib.ltr_mlp_pairwise_f32(
            kind_id, 1.0, X.data_ptr(), *[p.data_ptr() for p in params], relevance.data_ptr(),
            _C.LABEL_I64, n.data_ptr(), None, B, L, F, H1, H2, lossv.data_ptr(), None,
            grads.data_ptr(), lsum.data_ptr(), ws.data_ptr(), ws_bytes,
            torch.cuda.current_stream().cuda_stream))

    for _ in range(5):
        launch()
    us, graphed = time_launches(launch, per_graph=10, replays=10)
    docs = int(n.clamp(max=L).sum())
    flops_per_doc = 2 * (F * H1 + H1 * H2 + H2) + 2 * (F * H1 + 2 * H1 * H2 + H2)   # fwd + bwd, no dX
    out = {"step": "MLP %d-%d-%d-1 + %s fwd+bwd (2 launches: mlp_pairwise_kernel + mlp_reduce_kernel)" % (F, H1, H2, kind),
           "us_per_step": us, "queries_per_s": B / (us * 1e-6),
           "useful_TFLOPs": docs * flops_per_doc / (us * 1e-6) / 1e12,
           "f32_mfma_peak_TFLOPs": 157.3,
           "frac_of_f32_mfma_peak": docs * flops_per_doc / (us * 1e-6) / 1e12 / 157.3}
    loss_fn = {"hinge": L_.PairwiseHingeLoss, "dcg_hinge": L_.PairwiseDCGHingeLoss,
               "logistic": L_.PairwiseLogisticLoss, "arp1": L_.LambdaARPLoss1, "arp2": L_.LambdaARPLoss2,
               "ndcg1": L_.LambdaNDCGLoss1, "ndcg2": L_.LambdaNDCGLoss2}[kind]()
    ps = list(m.parameters())

    def unfused():
        for p_ in ps:
            p_.grad = None
        loss_fn(m.score(X), relevance, n).mean().backward()
    for _ in range(5):
        unfused()
    res = {"eager_us": time_wall(unfused, 50, lambda: None) / 50 * 1e6}
    rp = None if no_graph else try_graph(unfused)
    if rp is not None:
        res["hipgraph_us"] = time_wall(rp, 100, lambda: None) / 100 * 1e6
    out["unfused_torch_layers_plus_loss_module"] = res
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-graph", action="store_true", help="time the eager autograd path only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--full-lists", action="store_true", help="n == list_len for every query")
    args = ap.parse_args()

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
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("LTR_BENCH_FORCE_DIST") == "1":    # force: smoke-test RCCL at N=1
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    n_gpus = world
    if dist is not None:
        # ProcessGroupNCCL's watchdog thread polls HIP events; an event query while another
        # thread is capturing aborts the process ("operation not permitted when stream is
        # capturing").  With a process group alive, launch eagerly -- the direct C-ABI step is
        # host-cheap (3 launches + 1 collective) and was as fast as graph replay at N=1.
        global GRAPHS_OK
        GRAPHS_OK = os.environ.get("LTR_BENCH_DIST_GRAPH", "") in ("kernels", "all")   # experiments only

    def barrier():
        if dist is not None:
            dist.barrier()

    from pytorchltr_amd import _C
    from pytorchltr_amd import loss as L_
    from pytorchltr_amd.fused import FusedLinearLoss
    _C.lib()                                            # fail loudly if the extension is missing

    B, L, F, kind = WORKLOADS[args.workload]
    scores, relevance, n, X = synth(B, L, F, 1000 * rank, dev)
    if args.full_lists:
        n = torch.full_like(n, L)
    fused = FusedLinearLoss(F, kind).to(dev)
    lib = _C.lib()
    kind_id = getattr(_C, kind.upper())
    W = fused.weight.detach().reshape(F).contiguous()
    bvec = fused.bias.detach().reshape(1).contiguous()
    lossv = torch.empty(B, device=dev)
    ws_bytes = lib.ltr_linear_workspace_bytes(B, L, F)
    part = torch.empty(ws_bytes // 4, device=dev)
    # [dW (F) | db (1) | loss_sum | count]: the one bucket that is all-reduced when N > 1
    flat = torch.zeros(F + 3, device=dev)
    flat[F + 2] = float(B)
    # uniform upstream gradient 1/(B*N): after the SUM all-reduce the bucket holds the gradient
    # of the GLOBAL mean loss -- what `.mean().backward()` gives on the unsharded batch
    go = torch.full((B,), 1.0 / (B * n_gpus), device=dev)

    def cur_stream():
        # resolved per call: inside hipGraph capture the current stream is the capture stream
        return torch.cuda.current_stream().cuda_stream

    # ---- the step: fused scorer + loss forward, backward to dW/db, through the C ABI ----
    def fwd_bwd():
        st = cur_stream()
        _C.check(lib.ltr_linear_partials_f32(
            kind_id, 1.0, X.data_ptr(), W.data_ptr(), bvec.data_ptr(), relevance.data_ptr(),
            _C.LABEL_I64, n.data_ptr(), B, L, F, lossv.data_ptr(), None, part.data_ptr(), st))
        _C.check(lib.ltr_linear_reduce_loss_f32(part.data_ptr(), go.data_ptr(), lossv.data_ptr(), B, F,
                                                flat.data_ptr(), flat.data_ptr() + 4 * F,
                                                flat.data_ptr() + 4 * (F + 1), st))

    def fwd_bwd_allreduce():
        fwd_bwd()
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)

    results = {}
    modes = {}
    if dist is None:
        modes["eager"] = fwd_bwd
        replay = None if args.no_graph else try_graph(fwd_bwd)
        if replay is not None:
            modes["hipgraph"] = replay
    else:
        modes["eager"] = fwd_bwd_allreduce
        dist_graph = os.environ.get("LTR_BENCH_DIST_GRAPH", "")
        if dist_graph == "kernels":                 # graph of the two kernels + eager collective
            replay = try_graph(fwd_bwd)
            if replay is not None:
                def graph_then_allreduce():
                    replay()
                    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
                modes["hipgraph_kernels+eager_allreduce"] = graph_then_allreduce
        elif dist_graph == "all":                   # the collective captured too
            replay = try_graph(fwd_bwd_allreduce)
            if replay is not None:
                modes["hipgraph_with_allreduce"] = replay
    for name, fn in modes.items():
        for _ in range(args.warmup):
            fn()
        elapsed = time_wall(fn, args.steps, barrier)
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        if dist is not None:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        results[name] = float(t.item())
    mode = min(results, key=results.get)
    elapsed = results[mode]
    value = n_gpus * B * args.steps / elapsed

    out = None
    if rank == 0:
        # ---- roofline of the dominant kernel: fused scorer+loss, timed live with HIP events ----
        def launch_fused():
            _C.check(lib.ltr_linear_partials_f32(
                kind_id, 1.0, X.data_ptr(), W.data_ptr(), bvec.data_ptr(),
                relevance.data_ptr(), _C.LABEL_I64, n.data_ptr(), B, L, F, lossv.data_ptr(), None,
                part.data_ptr(), cur_stream()))

        for _ in range(10):
            launch_fused()
        k_evt_avg, k_evt_med, k_evt_min = time_events(launch_fused, 100)
        k_avg, k_graphed = time_launches(launch_fused)
        # algorithmic bytes per query of the fused step: features 4LF read
        # once + labels 8L + n 8 + W/bias 4(F+1) amortised per launch + loss 4 + partials 4(F+1)
        alg_bytes = B * (4 * L * F + 8 * L + 8 + 4 + 4 * (F + 1)) + 4 * (F + 1)
        achieved = alg_bytes / (k_avg * 1e-6) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_%s.json" % args.workload)
        if os.path.exists(tpath):
            with open(tpath) as fh:
                traffic = json.load(fh).get("hbm_bytes_per_launch")
        roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                    "kernel": "%s<%s> (via ltr_linear_partials_f32)" % (fused_kernel_name(kind_id, B, L, F), kind),
                    "kernel_us_avg": k_avg,
                    "timing": "HIP events around %s back-to-back launches" % ("hipGraph-replayed" if k_graphed else "eager"),
                    "kernel_us_single_launch_event_pair": {"avg": k_evt_avg, "median": k_evt_med, "min": k_evt_min},
                    "algorithmic_bytes_per_launch": alg_bytes}

        # ---- loss-only drop-in path (the literal "loss fwd+bwd" of the metric) ----
        loss_cls = {"hinge": L_.PairwiseHingeLoss, "dcg_hinge": L_.PairwiseDCGHingeLoss,
                    "logistic": L_.PairwiseLogisticLoss, "arp1": L_.LambdaARPLoss1,
                    "arp2": L_.LambdaARPLoss2, "ndcg1": L_.LambdaNDCGLoss1,
                    "ndcg2": L_.LambdaNDCGLoss2}[kind]
        loss_fn = loss_cls()
        sc = scores.clone().requires_grad_(True)

        def loss_step():
            sc.grad = None
            loss_fn(sc, relevance, n).mean().backward()

        extra = {"step_seconds": results}

        def measure(stepfn):
            for _ in range(args.warmup):
                stepfn()
            res = {"eager_queries_per_s": B * args.steps / time_wall(stepfn, args.steps, lambda: None)}
            rp = None if args.no_graph else try_graph(stepfn)
            if rp is not None:
                res["hipgraph_queries_per_s"] = B * args.steps / time_wall(rp, args.steps, lambda: None)
            return res

        # (a) the fused op as an autograd module: FusedLinearLoss(...)(xs, ys, n).mean().backward()
        params = [fused.weight, fused.bias]

        def module_step():
            for p_ in params:
                p_.grad = None
            fused(X, relevance, n).mean().backward()
        extra["fused_module_autograd"] = measure(module_step)

        # (b) the reference's own user code, unfused drop-in: torch Linear + our loss module
        lin = torch.nn.Linear(F, 1).to(dev)

        def dropin_step():
            lin.weight.grad = None
            lin.bias.grad = None
            loss_fn(lin(X), relevance, n).mean().backward()
        extra["dropin_linear_plus_loss_module"] = measure(dropin_step)

        # (b2) the same user code with the package's streaming scorer in place of nn.Linear
        from pytorchltr_amd.fused import LinearScorer
        scorer = LinearScorer(F).to(dev)

        def dropin_scorer_step():
            scorer.weight.grad = None
            scorer.bias.grad = None
            loss_fn(scorer(X, n), relevance, n).mean().backward()
        extra["dropin_linearscorer_plus_loss_module"] = measure(dropin_scorer_step)

        # (c) loss only (the literal "loss fwd+bwd" on precomputed scores)
        extra["loss_only_module"] = measure(loss_step)
        dsc = torch.empty(B, L, device=dev)

        def launch_loss():
            _C.check(lib.ltr_pairwise_loss_f32(
                getattr(_C, kind.upper()), 1.0, scores.data_ptr(), relevance.data_ptr(),
                _C.LABEL_I64, n.data_ptr(), B, L, lossv.data_ptr(), dsc.data_ptr(), cur_stream()))

        for _ in range(10):
            launch_loss()
        l_evt_avg, l_evt_med, l_evt_min = time_events(launch_loss, 100)
        l_avg, _ = time_launches(launch_loss)
        loss_bytes = B * (16 * L + 16)
        extra["loss_kernel"] = {"kernel": "pairwise_loss_kernel<%s>" % kind, "us_avg": l_avg,
                                "us_single_launch_event_pair": {"avg": l_evt_avg, "median": l_evt_med, "min": l_evt_min},
                                "algorithmic_bytes_per_launch": loss_bytes,
                                "achieved_GBs": loss_bytes / (l_avg * 1e-6) / 1e9,
                                "frac_of_hbm_peak": loss_bytes / (l_avg * 1e-6) / 1e9 / HBM_PEAK_GBS}

        if n_gpus == 1:
            mlp = mlp_extra(kind, X, relevance, n, args.no_graph)
            if mlp is not None:
                extra["mlp_scorer_fused"] = mlp

        cpu = None
        if n_gpus == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(kind, B, L, F)

        out = {
            "metric": "queries/sec loss fwd+bwd (B=%d, list_len=%d) + achieved HBM GB/s" % (B, L),
            "value": value, "unit": "queries/s", "n_gpus": n_gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "%s: Linear(%d,1) scorer + %s loss fwd+bwd, B=%d/GPU, "
                                   "list_len=%d, n~U[1,%d]%s" % (args.workload, F, kind, B, L, L,
                                                                " (full lists)" if args.full_lists else ""),
                       "global_batch": n_gpus * B, "list_len": L, "features": F, "loss": kind,
                       "mode": mode, "parallelism": "dp%d" % n_gpus},
            "roofline": roofline, "cpu_baseline": cpu, "extra": extra,
        }
    barrier()
    if dist is not None:
        dist.destroy_process_group()
    sys.stdout.flush()
    if out is not None:
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    os.close(result_fd)


if __name__ == "__main__":
    main()
