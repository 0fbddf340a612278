"""GPU tier: the persistent multi-batch launch ltr_linear_sgd_steps_f32 (K synchronous-SGD steps, one workgroup per
query position resident over all batches, the gradient reduction and the weight update done by the workgroups
themselves between the steps) against

  * the ORACLE's trajectory: W_{k+1} = W_k - lr * d mean loss / dW at W_k over K different batches
    (loss_fn(Linear(F,1)(xs), ys, n).mean().backward(); optimizer.step() -- examples/01-basic-usage.py:66-75),
    every step's mean gradient, loss sum and per-query losses;
  * the per-step entry point ltr_linear_sgd_step_f32 over the same batches (equal up to fp32 summation order);
  * itself: two runs are bit-identical (fixed summation order, no atomics);

at BASELINE's C2 shape (1024 x 128 x 136, every workgroup slot of the chip taken), on ragged / full / empty lists,
several kinds and feature widths, more steps than one launch holds, and with the waits forced to give up
(LTR_ERR_TIMEOUT, weights untouched).  Tolerances: gradients 5e-5 of the largest entry, weights rtol 1e-4."""
import numpy as np
import pytest
import torch

from oracle import ltr_oracle as O
from tests.conftest import synth

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _batches(K, B, L, F, seed, pattern="ragged"):
    out = []
    W = b = None
    for k in range(K):
        s, y, n, X, W_, b_ = synth(B, L, seed + 17 * k, F=F)
        if W is None:
            W, b = W_, b_
        if pattern == "full":
            n = torch.full_like(n, L)
        elif pattern == "holes":
            n[::7] = 0
            n[1::11] = 1
        out.append((X, y, n))
    return out, W, b


def _oracle_along(kind, batches, W, b, lr, grads):
    """The oracle's step at the weights the KERNEL held at each step -- W_k rebuilt in fp32 from the gradients it
    reported (W_{k+1} = W_k - lr * g_k, the kernel's own update) -- so that a hinge pair that flips under an ulp of
    weight difference after a few steps does not turn a trajectory test into a chaos test; the free-running oracle
    trajectory (its own gradients) is compared at the end with a tolerance that allows for such flips."""
    Wk, bk = W.clone().numpy().astype(np.float32), np.float32(b[0])
    Wh, bh = W.clone(), b.clone()
    steps = []
    F = W.numel()
    for k, (X, y, n) in enumerate(batches):
        B = X.shape[0]
        loss, _, dW, db = O.linear_pairwise(kind, X.numpy(), Wk, float(bk), y.numpy(), n.numpy(), np.full(B, 1.0 / B))
        steps.append((loss, dW, db))
        Wk = (Wk - np.float32(lr) * grads[k, :F].astype(np.float32)).astype(np.float32)
        bk = np.float32(bk - np.float32(lr) * np.float32(grads[k, F]))
        lo, _, dWo, dbo = O.linear_pairwise(kind, X.numpy(), Wh.numpy(), float(bh[0]), y.numpy(), n.numpy(), np.full(B, 1.0 / B))
        Wh = (Wh.double() - lr * torch.from_numpy(dWo)).float()
        bh = (bh.double() - lr * dbo).float()
    return steps, Wk, bk, Wh, bh


def _run_persistent(kind, batches, W, b, lr, dev):
    from pytorchltr_amd.fused import linear_sgd_steps
    Wd, bd = W.clone().to(dev), b.clone().to(dev)
    dbat = [(X.to(dev), y.to(dev), n.to(dev)) for X, y, n in batches]
    mean_loss, grads, losses = linear_sgd_steps(dbat, Wd, bd, lr, loss=kind, return_losses=True)
    torch.cuda.synchronize()
    return Wd.cpu(), bd.cpu(), mean_loss.cpu().numpy(), grads.cpu().numpy(), losses.cpu().numpy()


def _check(kind, K, B, L, F, seed, pattern="ragged", lr=0.05, want_plan=1):
    from pytorchltr_amd import _C
    lib = _C.lib()
    dev = _dev()
    assert lib.ltr_linear_sgd_steps_plan(getattr(_C, kind.upper()), B, L, F) == want_plan
    batches, W, b = _batches(K, B, L, F, seed, pattern)
    Wp, bp, mean_loss, grads, losses = _run_persistent(kind, batches, W, b, lr, dev)
    _C.device_status()
    steps, Wk, bk, Wh, bh = _oracle_along(kind, batches, W, b, lr, grads)
    for k, (loss, dW, db) in enumerate(steps):
        tol = 5e-5 * max(1.0, float(np.max(np.abs(dW)))) + 1e-6
        assert np.max(np.abs(grads[k, :F] - dW)) < tol, k
        assert abs(grads[k, F] - db) < tol, k
        assert np.allclose(losses[k], loss, rtol=2e-5, atol=1e-5), k
        assert np.isclose(mean_loss[k], float(np.mean(loss)), rtol=2e-5, atol=1e-6), k
    # the weights that left the kernel = its own updates applied to W (one fma rounding per step at most) ...
    assert np.allclose(Wp.numpy(), Wk, rtol=2e-6, atol=2e-6)
    assert abs(float(bp[0]) - float(bk)) < 1e-6
    # ... and the oracle's free-running trajectory (its own fp64 gradients, its own weights): pairs at their margin flip
    # on the way and move a gradient entry by x / B each, so the two trajectories agree to a fraction of one update
    upd = float(np.max(np.abs(Wh.numpy() - W.numpy()))) + 1e-6
    assert np.max(np.abs(Wp.numpy() - Wh.numpy())) < 2e-2 * upd
    assert abs(float(bp[0]) - float(bh[0])) < 2e-2 * max(upd, abs(float(bh[0] - b[0])))
    # the per-step entry point over the same batches
    Wd, bd = W.clone().to(dev), b.clone().to(dev)
    ws = torch.empty(lib.ltr_linear_workspace_bytes(B, L, F) // 4 + 64, device=dev)
    lossv = torch.empty(B, device=dev)
    bucket = torch.zeros(F + 2, device=dev)
    for X, y, n in batches:
        Xd, yd, nd = X.to(dev), y.to(dev), n.to(dev)
        _C.check(lib.ltr_linear_sgd_step_f32(getattr(_C, kind.upper()), 1.0, Xd.data_ptr(), Wd.data_ptr(), bd.data_ptr(), yd.data_ptr(),
                                             _C.LABEL_I64, nd.data_ptr(), None, B, L, F, lr, lossv.data_ptr(), bucket.data_ptr(),
                                             ws.data_ptr(), ws.numel() * 4, None, _C.stream_of(Xd)))
        torch.cuda.synchronize()
    assert np.max(np.abs(Wp.numpy() - Wd.cpu().numpy())) < 2e-2 * upd
    # and deterministic: a second run, bit for bit
    Wp2, bp2, _, grads2, losses2 = _run_persistent(kind, batches, W, b, lr, dev)
    assert torch.equal(Wp, Wp2) and torch.equal(bp, bp2)
    assert np.array_equal(grads, grads2) and np.array_equal(losses, losses2)


def test_c2_shape_every_workgroup_slot_taken():
    """BASELINE's C2: 1024 queries x 128 documents x 136 features, hinge -- four 512-thread workgroups on each of the
    256 CUs, all resident; six steps over six different batches."""
    _check("hinge", 6, 1024, 128, 136, 0)


@pytest.mark.parametrize("case", [("hinge", 5, 1024, 128, 136, "full"), ("dcg_hinge", 4, 700, 100, 136, "holes"),
                                  ("logistic", 4, 512, 128, 136, "ragged"), ("arp1", 3, 300, 64, 64, "ragged"),
                                  ("arp2", 3, 600, 40, 220, "ragged"), ("hinge", 4, 256, 256, 32, "ragged"),
                                  ("hinge", 3, 333, 37, 12, "holes")])
def test_kinds_widths_and_list_patterns(case):
    kind, K, B, L, F, pattern = case
    # (the log-sigmoid kinds with a small step: the reference's -- and the oracle's -- literal formulas overflow once the
    # weights have grown, SURVEY.md 8(c) "domain of parity")
    _check(kind, K, B, L, F, 3, pattern, lr=0.05 if kind in ("hinge", "dcg_hinge") else 0.001)


def test_more_steps_than_one_launch_holds():
    """K = 70 > 32 steps per launch: three launches, the weights carried from one to the next."""
    _check("hinge", 70, 256, 40, 24, 9, lr=0.01)


@pytest.mark.parametrize("case", [("ndcg2", 3, 256, 128, 136), ("hinge", 3, 200, 128, 136), ("hinge", 2, 1500, 64, 64),
                                  ("hinge", 2, 128, 300, 136), ("hinge", 2, 128, 64, 45)])
def test_shapes_the_persistent_kernel_declines_run_as_per_step_calls(case):
    """LambdaNDCG kinds, fewer queries than reducers need, more than can be resident, lists beyond the register tile,
    rows that are not whole float4: ltr_linear_sgd_steps_plan says 0 and the same entry point gives the same trajectory."""
    kind, K, B, L, F = case
    _check(kind, K, B, L, F, 4, want_plan=0)


def test_a_wait_that_gives_up_leaves_the_weights_alone():
    from pytorchltr_amd import _C
    from pytorchltr_amd.fused import linear_sgd_steps
    lib = _C.lib()
    dev = _dev()
    batches, W, b = _batches(3, 512, 64, 136, 2)
    Wd, bd = W.clone().to(dev), b.clone().to(dev)
    dbat = [(X.to(dev), y.to(dev), n.to(dev)) for X, y, n in batches]
    lib.ltr_debug_steps_force_timeout(1)
    try:
        linear_sgd_steps(dbat, Wd, bd, 0.05, loss="hinge")
        torch.cuda.synchronize()
    finally:
        lib.ltr_debug_steps_force_timeout(0)
    assert lib.ltr_device_status(0) != 0                       # LTR_ERR_TIMEOUT, sticky
    with pytest.raises(RuntimeError):                          # the next call reports it
        linear_sgd_steps(dbat, Wd, bd, 0.05, loss="hinge")
    assert torch.equal(Wd.cpu(), W) and torch.equal(bd.cpu(), b)
    assert lib.ltr_device_status(1) != 0
    # recovery
    _check("hinge", 3, 512, 64, 136, 2)
