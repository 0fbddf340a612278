"""CPU tier: the N>1 path with world_size=2 over gloo.  Queries are sharded across ranks with no
data-path collective; one flattened all-reduce turns per-rank sum-gradients into the global
mean gradient.  The per-rank loss here is the CPU oracle (the HIP kernels need a GPU); what is
under test is the sharding and the exchange -- pytorchltr_amd.distributed is backend-agnostic."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pytorchltr_amd.distributed import (allreduce_metric, allreduce_step, shard_batch,
                                        shard_bounds)
from tests.conftest import synth


def test_shard_bounds_cover_and_balance():
    for total in (0, 1, 7, 8, 1024, 1025):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(8, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, B, L, F, out_dir):
    from oracle import ltr_oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        s, y, n, X, W, b = synth(B, L, 77, F=F)
        Xs, ys, ns = shard_batch((X, y, n))
        lo, hi = shard_bounds(B, rank, world)
        assert Xs.shape[0] == hi - lo and torch.equal(ns, n[lo:hi])
        # local step: gradients of the SUM of this rank's per-query losses
        loss, _, dW, db = O.linear_pairwise("logistic", Xs.numpy(), W.numpy(), float(b[0]),
                                            ys.numpy(), ns.numpy(), np.ones(hi - lo))
        gW = torch.tensor(dW, dtype=torch.float32)
        gb = torch.tensor([db], dtype=torch.float32)
        mean_loss, count = allreduce_step([gW, gb], torch.tensor(float(loss.sum())), hi - lo)
        metric = allreduce_metric(torch.tensor(O.ndcg(
            (Xs @ W + b).numpy(), ys.numpy(), ns.numpy(), k=10), dtype=torch.float32))
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), gW=gW.numpy(), gb=gb.numpy(),
                 mean_loss=float(mean_loss), count=float(count), metric=float(metric))
    finally:
        dist.destroy_process_group()


def test_two_rank_gradient_allreduce_equals_single_process(tmp_path):
    from oracle import ltr_oracle as O
    B, L, F = 13, 12, 6                       # odd B: unequal shards
    port = _free_port()
    mp.spawn(_worker, args=(2, port, B, L, F, str(tmp_path)), nprocs=2, join=True)
    s, y, n, X, W, b = synth(B, L, 77, F=F)
    loss, scores, dW, db = O.linear_pairwise("logistic", X.numpy(), W.numpy(), float(b[0]),
                                             y.numpy(), n.numpy(), np.full(B, 1.0 / B))
    want_metric = O.ndcg(scores, y.numpy(), n.numpy(), k=10).mean()
    for rank in range(2):
        got = np.load(os.path.join(str(tmp_path), "rank%d.npz" % rank))
        assert np.allclose(got["gW"], dW, rtol=1e-5, atol=1e-6)        # == .mean().backward() unsharded
        assert np.allclose(got["gb"], db, rtol=1e-5, atol=1e-6)
        assert got["mean_loss"] == pytest.approx(loss.mean(), rel=1e-5)
        assert got["count"] == B
        assert got["metric"] == pytest.approx(want_metric, rel=1e-5)


def test_allreduce_step_is_identity_without_process_group():
    g = [torch.tensor([2.0, 4.0]), torch.tensor([6.0])]
    mean_loss, count = allreduce_step(g, torch.tensor(10.0), 2)
    assert torch.equal(g[0], torch.tensor([1.0, 2.0])) and torch.equal(g[1], torch.tensor([3.0]))
    assert float(mean_loss) == 5.0 and float(count) == 2.0


def _bucket_worker(rank, world, port, B, L, F, accum, out_dir):
    """bench.py's N > 1 step on CPU: `accum` micro-batches accumulated into the bucket
    [dW | db | loss_sum | count] with upstream gradient 1/(B*N), then ONE all-reduce."""
    import bench
    from oracle import ltr_oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        flat = torch.zeros(F + 3)
        dW, db, lsum, cnt = bench.bucket_views(flat, F)
        cnt[0] = float(B)
        go = np.full(B, 1.0 / (B * world))
        for j in range(accum):
            s, y, n, X, W, b = synth(B, L, 1000 * rank + j, F=F)       # bench: seeds 1000*rank + i
            W0 = synth(B, L, 0, F=F)[4]                               # replicated scorer weights
            loss, _, gW, gb = O.linear_pairwise("hinge", X.numpy(), W0.numpy(), 0.25, y.numpy(), n.numpy(), go)
            upd = (torch.tensor(gW, dtype=torch.float32), torch.tensor([gb], dtype=torch.float32),
                   torch.tensor([loss.sum()], dtype=torch.float32))
            for view, val in zip((dW, db, lsum), upd):                # reduce kernel: accumulate = (j != 0)
                view.copy_(val if j == 0 else view + val)
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        np.save(os.path.join(out_dir, "bucket%d.npy" % rank), flat.numpy())
    finally:
        dist.destroy_process_group()


def test_bench_bucket_layout_with_accumulation_two_ranks(tmp_path):
    """After the SUM all-reduce the bucket holds, per micro-batch, the gradient of the GLOBAL mean
    loss -- i.e. what `accum` calls of `.mean().backward()` on the unsharded batches accumulate --
    plus the loss total and the query count of all ranks."""
    from oracle import ltr_oracle as O
    B, L, F, accum, world = 6, 10, 4, 3, 2
    port = _free_port()
    mp.spawn(_bucket_worker, args=(world, port, B, L, F, accum, str(tmp_path)), nprocs=world, join=True)
    W0 = synth(B, L, 0, F=F)[4]
    want_dW = np.zeros(F)
    want_db = 0.0
    want_loss = 0.0
    for j in range(accum):
        Xs, ys, ns = [], [], []
        for rank in range(world):
            s, y, n, X, W, b = synth(B, L, 1000 * rank + j, F=F)
            Xs.append(X), ys.append(y), ns.append(n)
        Xg, yg, ng = torch.cat(Xs), torch.cat(ys), torch.cat(ns)
        loss, _, gW, gb = O.linear_pairwise("hinge", Xg.numpy(), W0.numpy(), 0.25, yg.numpy(), ng.numpy(),
                                            np.full(world * B, 1.0 / (world * B)))
        want_dW += gW
        want_db += gb
        want_loss += loss.sum()
    import bench
    for rank in range(world):
        flat = torch.tensor(np.load(os.path.join(str(tmp_path), "bucket%d.npy" % rank)))
        dW, db, lsum, cnt = bench.bucket_views(flat, F)
        assert np.allclose(dW.numpy(), want_dW, rtol=1e-5, atol=1e-6)
        assert float(db) == pytest.approx(want_db, rel=1e-5, abs=1e-6)
        assert float(lsum) == pytest.approx(want_loss, rel=1e-5)
        assert float(cnt) == world * B


def _overlap_worker(rank, world, port, B, L, F, steps, out_dir):
    """bench.py's N > 1 headline step on CPU: one all-reduce per step through the double-buffered,
    asynchronous bucket, next to a plain blocking all-reduce of the same local values."""
    import bench
    from oracle import ltr_oracle as O
    from pytorchltr_amd.distributed import OverlappedBucketAllReduce
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        Bl = B + rank                                                  # unequal shards: the count is summed once
        red = OverlappedBucketAllReduce(F, count=Bl, device="cpu")
        total = red.global_count
        W0 = synth(B, L, 0, F=F)[4]
        got, want = [], []
        for i in range(steps):
            s, y, n, X, W, b = synth(Bl, L, 1000 * rank + i, F=F)
            go = np.full(Bl, 1.0 / total)
            loss, _, gW, gb = O.linear_pairwise("logistic", X.numpy(), W0.numpy(), 0.25, y.numpy(), n.numpy(), go)
            local = torch.tensor(np.r_[gW, gb, loss.sum()], dtype=torch.float32)
            flat = red.acquire(i)                                      # (waits for step i - 2's collective)
            dW, db, lsum, cnt = bench.bucket_views(flat, F)
            dW.copy_(local[:F]); db.copy_(local[F:F + 1]); lsum.copy_(local[F + 1:F + 2])   # "the reduce kernel"
            red.launch(i)
            ref = local.clone()
            dist.all_reduce(ref, op=dist.ReduceOp.SUM)                 # the un-overlapped exchange
            want.append(ref.numpy().copy())
            if i >= 1:
                got.append(red.result(i - 1).numpy().copy())          # the optimiser reads step i - 1 here
        got.append(red.result(steps - 1).numpy().copy())
        red.flush()
        np.savez(os.path.join(out_dir, "overlap%d.npz" % rank), got=np.array(got), want=np.array(want), total=total)
    finally:
        dist.destroy_process_group()


def test_overlapped_bucket_equals_blocking_allreduce_two_ranks(tmp_path):
    """The double-buffered asynchronous bucket gives, step for step, exactly what a blocking all-reduce
    of the same local values gives (bit-identical: same collective, same operands), and the count slot
    holds the global query count without being reduced again every step."""
    B, L, F, steps, world = 5, 9, 4, 6, 2
    port = _free_port()
    mp.spawn(_overlap_worker, args=(world, port, B, L, F, steps, str(tmp_path)), nprocs=world, join=True)
    for rank in range(world):
        z = np.load(os.path.join(str(tmp_path), "overlap%d.npz" % rank))
        assert float(z["total"]) == 2 * B + 1
        assert z["got"].shape == (steps, F + 3)
        assert np.array_equal(z["got"][:, :F + 2], z["want"])
        assert np.all(z["got"][:, F + 2] == 2 * B + 1)
    a = np.load(os.path.join(str(tmp_path), "overlap0.npz"))["got"]
    b = np.load(os.path.join(str(tmp_path), "overlap1.npz"))["got"]
    assert np.array_equal(a, b)
