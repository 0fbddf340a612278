"""GPU tier: the long-list fused configs at BASELINE's FULL sizes against the oracle on every row
(VERDICT r1 item 2): C4 256 x 1000 x 220 PairwiseDCGHingeLoss on the cluster kernel (a query spread
over several workgroups that wait for each other), C5 512 x 512 x 700 PairwiseHingeLoss on the
general kernel -- each run twice (bit-identical) and, for the cluster kernel, once more while a
second stream keeps the GPU busy; plus the failure mode of the in-launch waits: a wait that gives
up must surface as LTR_ERR_TIMEOUT, not as a silent NaN.
Reference: loss/pairwise_additive.py:51-90,116-133 composed with torch.nn.Linear
(examples/01-basic-usage.py:66-75)."""
import numpy as np
import pytest
import torch

from oracle import ltr_oracle as O
from tests.conftest import synth

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _run(kind, B, L, F, seed, want_plan, busy=False):
    from pytorchltr_amd import _C
    from pytorchltr_amd.fused import linear_loss_step
    dev = _dev()
    s, y, n, X, W, b = synth(B, L, seed, F=F)
    Xd, Wd, bd, yd, nd = X.to(dev), W.to(dev), b.to(dev), y.to(dev), n.to(dev)
    plan = _C.lib().ltr_linear_fused_plan(getattr(_C, kind.upper()), B, L, F)
    assert plan == want_plan, "shape %s takes plan %d, expected %d" % ((B, L, F), plan, want_plan)
    outs = []
    for rep in range(2):
        loss, dW, db = linear_loss_step(Xd, Wd, bd, yd, nd, loss=kind)
        outs.append((loss.cpu().numpy(), dW.cpu().numpy(), db.cpu().numpy()))
    if busy:
        # a second stream saturates the CUs with GEMMs while the step runs
        side = torch.cuda.Stream()
        a = torch.randn(4096, 4096, device=dev)
        with torch.cuda.stream(side):
            for _ in range(40):
                a = torch.tanh(a @ a) * 0.01
        for rep in range(3):
            loss, dW, db = linear_loss_step(Xd, Wd, bd, yd, nd, loss=kind)
            outs.append((loss.cpu().numpy(), dW.cpu().numpy(), db.cpu().numpy()))
        side.synchronize()
    _C.device_status()                                     # no wait gave up
    for o in outs[1:]:
        for got, first in zip(o, outs[0]):
            assert np.array_equal(got, first), "runs differ (kernel must be deterministic)"
    want_l, _, want_dW, want_db = O.linear_pairwise(kind, X.numpy(), W.numpy(), float(b[0]), y.numpy(),
                                                    n.numpy(), np.full(B, 1.0 / B))
    loss, dW, db = outs[0]
    assert np.all(np.isfinite(loss))
    assert np.allclose(loss, want_l, rtol=5e-4, atol=1e-5)             # every row
    tol = 2e-4 * max(1.0, float(np.max(np.abs(want_dW))))
    assert np.max(np.abs(dW - want_dW)) < tol
    assert abs(float(db[0]) - want_db) < tol


def test_c4_full_size_cluster_kernel_all_rows():
    from pytorchltr_amd import _C
    _run("dcg_hinge", 256, 1000, 220, 0, _C.PLAN_CLUSTER, busy=True)


def test_c4_shard_of_8_gpus_cluster_kernel():
    from pytorchltr_amd import _C
    _run("dcg_hinge", 32, 1000, 220, 1, _C.PLAN_CLUSTER)


def test_c5_full_size_general_kernel_all_rows():
    from pytorchltr_amd import _C
    _run("hinge", 512, 512, 700, 0, _C.PLAN_GENERAL)


def test_c5_shard_of_8_gpus():
    from pytorchltr_amd import _C
    plan = _C.lib().ltr_linear_fused_plan(_C.HINGE, 64, 512, 700)
    _run("hinge", 64, 512, 700, 2, plan)


def test_cluster_wait_timeout_is_an_error_not_a_silent_nan():
    """ltr_debug_force_timeout makes every in-launch wait give up: the step's outputs are poisoned
    AND the sticky status word turns the next ltr_linear_* call into LTR_ERR_TIMEOUT."""
    from pytorchltr_amd import _C
    from pytorchltr_amd.fused import linear_loss_step
    dev = _dev()
    B, L, F = 32, 1000, 220
    s, y, n, X, W, b = synth(B, L, 3, F=F)
    args = (X.to(dev), W.to(dev), b.to(dev), y.to(dev), n.to(dev))
    assert _C.lib().ltr_linear_fused_plan(_C.HINGE, B, L, F) == _C.PLAN_CLUSTER
    good = linear_loss_step(*args, loss="hinge")
    torch.cuda.synchronize()
    assert _C.lib().ltr_device_status(0) == 0
    lib = _C.lib()
    lib.ltr_debug_force_timeout(1)
    try:
        loss, dW, db = linear_loss_step(*args, loss="hinge")
        torch.cuda.synchronize()
    finally:
        lib.ltr_debug_force_timeout(0)
    multi = n.numpy() > 0
    assert np.all(np.isnan(loss.cpu().numpy()[multi])), "a failed wait must poison the loss"
    assert np.all(np.isnan(dW.cpu().numpy()))
    assert lib.ltr_device_status(0) == _C.ERR_TIMEOUT
    with pytest.raises(RuntimeError, match="gave up"):
        linear_loss_step(*args, loss="hinge")              # sticky: the NEXT call reports it
    with pytest.raises(RuntimeError, match="gave up"):
        _C.device_status()                                 # explicit query, clears the flag
    assert lib.ltr_device_status(0) == 0
    again = linear_loss_step(*args, loss="hinge")
    torch.cuda.synchronize()
    for a, g in zip(again, good):
        assert torch.equal(a, g)
