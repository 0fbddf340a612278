"""GPU tier: the data-parallel step entry points against the oracle (ltr_linear_step_f32 and the
overlap handle had only HIP-against-HIP checks inside a bench subprocess).

A second rank is played by `ltr_debug_fake_allreduce`, an ltr_allreduce_fn that adds a known bucket -- the
step of the OTHER shard, computed beforehand -- on the stream it is given: the all-reduced bucket of shard 1
must then be the gradient of the mean loss over the CONCATENATED batch, which the oracle computes directly
(reference step: loss_fn(Linear(F,1)(xs), ys, n).mean().backward(), examples/01-basic-usage.py:66-75, sharded
by queries as SURVEY.md 8(e) prescribes).  In-stream handle (depth 0), helper-thread handle (depth 2, three
rotating slots' worth of steps), and the synchronous-SGD step ltr_linear_sgd_step_f32 over several updates.
Tolerances: gradients 2e-5 of the largest entry + 1e-6, loss sums rtol 1e-5 (fp32 sums in another order)."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import ltr_oracle as O
from tests.conftest import synth

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _shards(B1, B2, L, F, seed):
    s, y, n, X, W, b = synth(B1 + B2, L, seed, F=F)
    return (X, y, n, W, b), (X[:B1], y[:B1], n[:B1]), (X[B1:], y[B1:], n[B1:])


def _oracle(kind, X, W, b, y, n):
    B = X.shape[0]
    loss, _, dW, db = O.linear_pairwise(kind, X.numpy(), W.numpy(), float(b[0]), y.numpy(), n.numpy(),
                                        np.full(B, 1.0 / B))
    return loss, dW, db


def _step(lib, _C, kind_id, Xd, Wd, bd, yd, nd, go, lossv, bucket, ws, handle, slot):
    B, L, F = Xd.shape
    _C.check(lib.ltr_linear_step_f32(kind_id, 1.0, Xd.data_ptr(), Wd.data_ptr(), bd.data_ptr(), yd.data_ptr(),
                                     _C.LABEL_I64, nd.data_ptr(), go.data_ptr(), B, L, F, lossv.data_ptr(),
                                     bucket.data_ptr(), 0, ws.data_ptr(), ws.numel() * 4, handle, slot,
                                     _C.stream_of(Xd)))


@pytest.mark.parametrize("shape", [("hinge", 70, 42, 128, 136), ("ndcg2", 33, 31, 100, 64), ("dcg_hinge", 20, 12, 1000, 220),
                                   ("logistic", 40, 30, 512, 700)])
@pytest.mark.parametrize("depth", [0, 2])
def test_step_with_the_other_ranks_bucket_added_in_the_allreduce(shape, depth):
    from pytorchltr_amd import _C
    lib = _C.lib()
    dev = _dev()
    kind, B1, B2, L, F = shape
    kind_id = getattr(_C, kind.upper())
    (X, y, n, W, b), (X1, y1, n1), (X2, y2, n2) = _shards(B1, B2, L, F, 11)
    Wd, bd = W.to(dev), b.to(dev)
    go1 = torch.full((B1,), 1.0 / (B1 + B2), device=dev)
    go2 = torch.full((B2,), 1.0 / (B1 + B2), device=dev)
    d1 = [t.to(dev) for t in (X1, y1, n1)]
    d2 = [t.to(dev) for t in (X2, y2, n2)]
    ws = torch.empty(max(lib.ltr_linear_workspace_bytes(B1, L, F), lib.ltr_linear_workspace_bytes(B2, L, F)) // 4 + 64,
                     device=dev)
    loss1, loss2 = torch.empty(B1, device=dev), torch.empty(B2, device=dev)
    other = torch.zeros(F + 2, device=dev)
    _step(lib, _C, kind_id, d2[0], Wd, bd, d2[1], d2[2], go2, loss2, other, ws, None, 0)     # "rank 1"
    torch.cuda.synchronize()
    want_loss, want_dW, want_db = _oracle(kind, X, W, b, y, n)
    fn = ctypes.cast(lib.ltr_debug_fake_allreduce, ctypes.c_void_p)
    handle = ctypes.c_void_p(None)
    _C.check(lib.ltr_overlap_create(fn, ctypes.c_void_p(other.data_ptr()), depth, ctypes.byref(handle)))
    try:
        buckets = [torch.zeros(F + 2, device=dev) for _ in range(max(1, depth))]
        wss = [torch.empty_like(ws) for _ in range(max(1, depth))]
        for i in range(5 if depth else 2):
            k = i % max(1, depth)
            _step(lib, _C, kind_id, d1[0], Wd, bd, d1[1], d1[2], go1, loss1, buckets[k], wss[k], handle, k)
            # (the stream waits for this slot's all-reduce -- on the helper's side stream when depth > 0)
            _C.check(lib.ltr_overlap_wait(handle, k, _C.stream_of(d1[0])))
            torch.cuda.synchronize()
            got = buckets[k].cpu().numpy()
            tol = 2e-5 * max(1.0, float(np.max(np.abs(want_dW)))) + 1e-6
            assert np.max(np.abs(got[:F] - want_dW)) < tol, (i, k)
            assert abs(got[F] - want_db) < tol, (i, k)
            assert np.isclose(got[F + 1], float(np.sum(want_loss)), rtol=2e-5, atol=1e-4), (i, k)
        _C.check(lib.ltr_overlap_flush(handle))
    finally:
        lib.ltr_overlap_destroy(handle)
    _C.device_status()


@pytest.mark.parametrize("shape", [("hinge", 96, 0, 128, 136), ("hinge", 70, 42, 128, 136), ("dcg_hinge", 20, 12, 1000, 220),
                                   ("ndcg2", 64, 0, 128, 136)])
def test_sgd_step_follows_the_oracles_trajectory(shape):
    """ltr_linear_sgd_step_f32 over four updates: W_{k+1} = W_k - lr * d mean loss / dW at W_k, against the same
    recursion on the oracle (fp64 gradients, fp32 weights); with a second shard added through the in-stream
    handle the gradient is the concatenated batch's, and the update waits for it."""
    from pytorchltr_amd import _C
    lib = _C.lib()
    dev = _dev()
    kind, B1, B2, L, F = shape
    kind_id = getattr(_C, kind.upper())
    (X, y, n, W, b), (X1, y1, n1), (X2, y2, n2) = _shards(B1, B2, L, F, 5)
    lr = 0.05
    Wd, bd = W.clone().to(dev), b.clone().to(dev)
    d1 = [t.to(dev) for t in (X1, y1, n1)]
    go1 = torch.full((B1,), 1.0 / (B1 + B2), device=dev)
    ws = torch.empty(lib.ltr_linear_workspace_bytes(max(B1, B2, 1), L, F) // 4 + 64, device=dev)
    loss1 = torch.empty(B1, device=dev)
    bucket = torch.zeros(F + 2, device=dev)
    handle = ctypes.c_void_p(None)
    other = torch.zeros(F + 2, device=dev)
    if B2:
        d2 = [t.to(dev) for t in (X2, y2, n2)]
        go2 = torch.full((B2,), 1.0 / (B1 + B2), device=dev)
        loss2 = torch.empty(B2, device=dev)
        fn = ctypes.cast(lib.ltr_debug_fake_allreduce, ctypes.c_void_p)
        _C.check(lib.ltr_overlap_create(fn, ctypes.c_void_p(other.data_ptr()), 0, ctypes.byref(handle)))
    Wh, bh = W.clone(), b.clone()
    try:
        for k in range(4):
            if B2:      # "rank 1" computes its bucket with the CURRENT weights (no update of its own: one model)
                _step(lib, _C, kind_id, d2[0], Wd, bd, d2[1], d2[2], go2, loss2, other, ws, None, 0)
            _C.check(lib.ltr_linear_sgd_step_f32(kind_id, 1.0, d1[0].data_ptr(), Wd.data_ptr(), bd.data_ptr(),
                                                 d1[1].data_ptr(), _C.LABEL_I64, d1[2].data_ptr(), go1.data_ptr(),
                                                 B1, L, F, lr, loss1.data_ptr(), bucket.data_ptr(), ws.data_ptr(),
                                                 ws.numel() * 4, handle if B2 else None, _C.stream_of(d1[0])))
            want_loss, want_dW, want_db = _oracle(kind, X, Wh, bh, y, n)
            torch.cuda.synchronize()
            got = bucket.cpu().numpy()
            tol = 5e-5 * max(1.0, float(np.max(np.abs(want_dW)))) + 1e-6
            assert np.max(np.abs(got[:F] - want_dW)) < tol, k
            Wh = (Wh.double() - lr * torch.from_numpy(want_dW)).float()
            bh = (bh.double() - lr * want_db).float()
            assert np.allclose(Wd.cpu().numpy(), Wh.numpy(), rtol=1e-4, atol=1e-5), k
            assert np.allclose(bd.cpu().numpy(), bh.numpy(), rtol=1e-4, atol=1e-5), k
    finally:
        if B2:
            lib.ltr_overlap_destroy(handle)
    _C.device_status()


def test_sgd_step_refuses_a_deferred_allreduce():
    from pytorchltr_amd import _C
    lib = _C.lib()
    dev = _dev()
    s, y, n, X, W, b = synth(8, 16, 3, F=8)
    t = [v.to(dev) for v in (X, W, b, y, n)]
    other = torch.zeros(10, device=dev)
    handle = ctypes.c_void_p(None)
    fn = ctypes.cast(lib.ltr_debug_fake_allreduce, ctypes.c_void_p)
    _C.check(lib.ltr_overlap_create(fn, ctypes.c_void_p(other.data_ptr()), 2, ctypes.byref(handle)))
    try:
        ws = torch.empty(lib.ltr_linear_workspace_bytes(8, 16, 8) // 4 + 64, device=dev)
        rc = lib.ltr_linear_sgd_step_f32(_C.HINGE, 1.0, t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), t[3].data_ptr(),
                                         _C.LABEL_I64, t[4].data_ptr(), None, 8, 16, 8, 0.1, torch.empty(8, device=dev).data_ptr(),
                                         other.data_ptr(), ws.data_ptr(), ws.numel() * 4, handle, _C.stream_of(t[0]))
        assert rc == _C.ERR_CONFIG
    finally:
        lib.ltr_overlap_destroy(handle)


@pytest.mark.parametrize("shape", [("hinge", 1024, 128, 136), ("hinge", 300, 128, 136), ("ndcg2", 512, 128, 136),
                                   ("logistic", 257, 100, 220), ("dcg_hinge", 600, 60, 64), ("arp2", 96, 200, 136),
                                   ("hinge", 300, 10, 700), ("ndcg1", 1, 128, 136), ("hinge", 40, 1000, 220),
                                   # more than 1024 pending queries: several reducer workgroups per column group, inside the launch
                                   ("hinge", 2500, 64, 136), ("logistic", 5000, 40, 64), ("logistic", 4096, 100, 136)])
def test_lazy_sgd_steps_are_the_eager_steps_bit_for_bit(shape):
    """ltr_linear_sgd_lazy_step_f32: step k + 1's launch applies step k's update itself (its first workgroups reduce the
    pending batch's partial rows in front of their tile burst and hand the new weights over as tagged granules), the last
    update by ltr_linear_sgd_flush_f32.  Over six batches of three in rotation: the weights after every flush point, every
    step's bucket [dW | db | loss sum] and per-query losses are BIT-IDENTICAL to ltr_linear_sgd_step_f32's (which the test
    above holds against the oracle's trajectory) -- on the register-tile shapes (the update rides in the launch) and on
    shapes that flush first and run the plain launch (the last one: the cluster kernel)."""
    from pytorchltr_amd import _C
    lib = _C.lib()
    dev = _dev()
    kind, B, L, F = shape
    kind_id = getattr(_C, kind.upper())
    lr = 0.05
    batches = []
    for i in range(3):
        s, y, n, X, W, b = synth(B, L, 11 + i, F=F)
        batches.append([t.to(dev) for t in (X, y, n)])
    _, _, _, _, W0, b0 = synth(B, L, 11, F=F)
    st = _C.stream_of(batches[0][0])
    nws = lib.ltr_linear_workspace_bytes(B, L, F)

    def run(lazy):
        Wd, bd = W0.clone().to(dev), b0.clone().to(dev)
        ws = torch.full((nws // 4 + 64,), float("nan"), device=dev)
        loss = torch.empty(B, device=dev)
        bucket = torch.zeros(F + 2, device=dev)
        trace = []
        pending = 0
        for k in range(6):
            Xd, yd, nd = batches[k % 3]
            if lazy:
                _C.check(lib.ltr_linear_sgd_lazy_step_f32(kind_id, 1.0, Xd.data_ptr(), Wd.data_ptr(), bd.data_ptr(), yd.data_ptr(),
                                                          _C.LABEL_I64, nd.data_ptr(), B, L, F, lr, loss.data_ptr(), bucket.data_ptr(),
                                                          ws.data_ptr(), ws.numel() * 4, pending, st))
                if pending:
                    trace.append(("bucket", k - 1, bucket.clone()))      # (the previous step's, written by this launch)
                pending = B
                trace.append(("loss", k, loss.clone()))
                if k in (2, 5):        # read the weights: flush first
                    _C.check(lib.ltr_linear_sgd_flush_f32(kind_id, Wd.data_ptr(), bd.data_ptr(), pending, L, F, lr, loss.data_ptr(),
                                                          bucket.data_ptr(), ws.data_ptr(), st))
                    pending = 0
                    trace.append(("bucket", k, bucket.clone()))
                    trace.append(("W", k, Wd.clone(), bd.clone()))
            else:
                _C.check(lib.ltr_linear_sgd_step_f32(kind_id, 1.0, Xd.data_ptr(), Wd.data_ptr(), bd.data_ptr(), yd.data_ptr(),
                                                     _C.LABEL_I64, nd.data_ptr(), None, B, L, F, lr, loss.data_ptr(),
                                                     bucket.data_ptr(), ws.data_ptr(), ws.numel() * 4, None, st))
                trace.append(("loss", k, loss.clone()))
                trace.append(("bucket", k, bucket.clone()))
                if k in (2, 5):
                    trace.append(("W", k, Wd.clone(), bd.clone()))
        torch.cuda.synchronize()
        _C.device_status()
        return {(e[0], e[1]): [t.cpu().numpy() for t in e[2:]] for e in trace}

    eager, lazy = run(False), run(True)
    assert set(eager) == set(lazy)
    # beyond 1024 pending queries several reducer workgroups share a column group (round 6: a reducer never sums more than ~1024
    # rows) and the last to arrive adds their partial sums in split order: another -- fixed -- summation order than the strided
    # reduction launch's, so the same numbers to rounding, and the same BITS run after run
    exact = B <= 1024
    for key in sorted(eager):
        for a, b2 in zip(eager[key], lazy[key]):
            assert np.all(np.isfinite(a)), key
            if exact:
                assert np.array_equal(a, b2), key
            else:
                assert np.allclose(a, b2, rtol=1e-4, atol=1e-5 * max(1.0, float(np.abs(a).max()))), (key, float(np.abs(a - b2).max()), float(np.abs(a).max()))
    if not exact:
        again = run(True)
        for key in sorted(lazy):
            for a, b2 in zip(lazy[key], again[key]):
                assert np.array_equal(a, b2), key


def test_lazy_sgd_module_trains_like_the_reference_loop():
    """pytorchltr_amd.fused.LazySGD over five batches: the weights after flush() equal torch.optim.SGD on
    loss_fn(nn.Linear(F, 1)(xs), ys, n).mean() through this package's loss module (itself held against the reference's
    vectors) to fp32 summation order, and the returned mean loss / gradient are the last batch's."""
    from pytorchltr_amd.fused import LazySGD
    from pytorchltr_amd.loss import PairwiseLogisticLoss
    dev = _dev()
    B, L, F = 300, 100, 136            # (the logistic loss: smooth -- the hinge's trajectory forks at every pair on the margin)
    lr = 0.05
    data = []
    for i in range(5):
        s, y, n, X, W, b = synth(B, L, 31 + i, F=F)
        data.append((X.to(dev), y.to(dev), n.to(dev)))
    _, _, _, _, W0, b0 = synth(B, L, 31, F=F)
    w, bias = W0.clone().to(dev), b0.clone().to(dev)
    opt = LazySGD(w, bias, lr, loss="logistic")
    for xs, ys, n in data:
        opt.step(xs, ys, n)
    mean_loss, grad = opt.flush()
    model = torch.nn.Linear(F, 1).to(dev)
    with torch.no_grad():
        model.weight.copy_(W0.reshape(1, F))
        model.bias.copy_(b0)
    sgd = torch.optim.SGD(model.parameters(), lr=lr)
    loss_fn = PairwiseLogisticLoss()
    last = None
    for xs, ys, n in data:
        sgd.zero_grad()
        last = loss_fn(model(xs), ys, n).mean()
        last.backward()
        sgd.step()
    torch.cuda.synchronize()
    assert torch.allclose(w, model.weight.detach().reshape(F), rtol=1e-4, atol=1e-6)
    assert torch.allclose(bias, model.bias.detach(), rtol=1e-4, atol=1e-6)
    assert torch.allclose(mean_loss, last.detach(), rtol=1e-5)
    assert torch.allclose(grad[:F], model.weight.grad.reshape(F), rtol=1e-4, atol=1e-6)


def test_lazy_sgd_step_that_never_gets_its_weights_raises_the_status():
    """The weights' hand-over inside the lazy launch is a bounded wait: with the waits forced to give up (ltr_debug_force_timeout)
    the step scores with NaN weights and raises LTR_ERR_TIMEOUT in the sticky device status instead of hanging; once the status
    has been cleared the next steps run clean."""
    from pytorchltr_amd import _C
    lib = _C.lib()
    dev = _dev()
    B, L, F = 300, 128, 136
    s, y, n, X, W, b = synth(B, L, 3, F=F)
    Xd, yd, nd, Wd, bd = X.to(dev), y.to(dev), n.to(dev), W.clone().to(dev), b.clone().to(dev)
    ws = torch.zeros(lib.ltr_linear_workspace_bytes(B, L, F) // 4 + 64, device=dev)
    loss = torch.empty(B, device=dev)
    bucket = torch.zeros(F + 2, device=dev)
    st = _C.stream_of(Xd)

    def step(pending):
        return lib.ltr_linear_sgd_lazy_step_f32(_C.HINGE, 1.0, Xd.data_ptr(), Wd.data_ptr(), bd.data_ptr(), yd.data_ptr(), _C.LABEL_I64,
                                                nd.data_ptr(), B, L, F, 0.01, loss.data_ptr(), bucket.data_ptr(), ws.data_ptr(),
                                                ws.numel() * 4, pending, st)
    assert step(0) == 0
    lib.ltr_debug_force_timeout(1)
    try:
        assert step(B) == 0
        torch.cuda.synchronize()
    finally:
        lib.ltr_debug_force_timeout(0)
    assert lib.ltr_device_status(1) == _C.ERR_TIMEOUT       # (the step scored with NaN weights: whatever it wrote is invalid)
    # a clean restart
    Wd.copy_(W.to(dev)); bd.copy_(b.to(dev))
    assert step(0) == 0 and step(B) == 0
    torch.cuda.synchronize()
    _C.device_status()
    assert np.all(np.isfinite(loss.cpu().numpy())) and np.all(np.isfinite(Wd.cpu().numpy()))
