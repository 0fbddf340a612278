"""GPU tier: the long-list fused configs at BASELINE's FULL sizes against the oracle on every row
(VERDICT r1 item 2): C4 256 x 1000 x 220 PairwiseDCGHingeLoss on the cluster kernel (a query spread
over several workgroups that wait for each other), C5 512 x 512 x 700 PairwiseHingeLoss on the
parts kernel (round 3: features read once; the general kernel on a narrower twin) -- each run twice
(bit-identical) and, for the kernels whose workgroups wait for each other, three more times while a
second stream keeps the GPU busy; workspace reuse and graph replay of the parts kernel's self-resetting
control block; plus the failure mode of the in-launch waits: a wait that gives up must surface as
LTR_ERR_TIMEOUT, not as a silent NaN.
Reference: loss/pairwise_additive.py:51-90,116-133 composed with torch.nn.Linear
(examples/01-basic-usage.py:66-75)."""
import numpy as np
import pytest
import torch

from oracle import ltr_oracle as O
from tests.conftest import assert_rank_dependent_losses, synth

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _parts_kernel_wherever_it_applies(request):
    """The tests of the parts kernel's mechanics (exchange areas, tags, graph replay, the counting formulation, every
    kind) run on shapes smaller or narrower than the ones where the kernel is measured to pay and the dispatcher
    picks it (choose_parts_shape): for them ltr_debug_parts_all lets it take every shape it can.  The BASELINE
    shapes (C4, C5 and their shards) are tested on the plan the dispatcher really picks."""
    from pytorchltr_amd import _C
    name = request.node.name
    on = ("parts_kernel" in name or "exchange_area" in name) and "c5_full_size" not in name
    old = _C.lib().ltr_debug_parts_all(1 if on else 0)
    yield
    _C.lib().ltr_debug_parts_all(old)


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
    want_l, want_s, want_dW, want_db = O.linear_pairwise(kind, X.numpy(), W.numpy(), float(b[0]), y.numpy(),
                                                         n.numpy(), np.full(B, 1.0 / B))
    loss, dW, db = outs[0]
    assert np.all(np.isfinite(loss))
    if kind in ("ndcg1", "ndcg2"):
        # every row at 5e-4, a row that is off against the fp64 scores must be a TESTED rank flip of nearly tied fp32
        # scores (tests/conftest.py: assert_rank_dependent_losses) -- rounds 3-4 waved 3 % of the rows through at 5e-3
        assert_rank_dependent_losses(kind, loss, X, W, b, y, n, want_l, want_s, rtol=5e-4)
    else:
        assert np.isclose(loss, want_l, rtol=5e-4, atol=1e-5).all()               # every row
    tol = 2e-4 * max(1.0, float(np.max(np.abs(want_dW))))
    assert np.max(np.abs(dW - want_dW)) < tol
    assert abs(float(db[0]) - want_db) < tol


def test_c4_full_size_cluster_kernel_all_rows():
    from pytorchltr_amd import _C
    _run("dcg_hinge", 256, 1000, 220, 0, _C.PLAN_CLUSTER, busy=True)


def test_c4_shape_lambda_ndcg_on_the_cluster_kernel_with_the_rank_exchange():
    """Round 4: C4's shape with the LambdaNDCG losses -- the members of a cluster rank their own rows and exchange the
    ranks and their terms of maxDCG behind a FOURTH counter (one more in-launch wait): every row against the oracle,
    twice bit-identical, and three more times under a busy second stream."""
    from pytorchltr_amd import _C
    _run("ndcg2", 256, 1000, 220, 2, _C.PLAN_CLUSTER, busy=True)
    _run("ndcg1", 200, 700, 136, 3, _C.PLAN_CLUSTER, busy=True)
    # (a batch size at which LambdaNDCG2 and the logistic loss both decline the cluster kernel and LambdaNDCG1 takes it:
    # the workspace must be sized for the largest need over the kinds -- found by scripts/dev/fuzz_dispatch.py)
    _run("ndcg1", 272, 1000, 220, 5, _C.PLAN_CLUSTER)


def test_c4_shard_of_8_gpus_cluster_kernel():
    from pytorchltr_amd import _C
    _run("dcg_hinge", 32, 1000, 220, 1, _C.PLAN_CLUSTER)


def test_c5_full_size_parts_kernel_all_rows():
    """C5 at full size on the parts kernel (features read once, one exchange per query, persistent
    workgroups drawing tickets): every row against the oracle, run twice (bit-identical) and three more
    times while a second stream keeps the CUs busy (fewer of the persistent workgroups resident)."""
    from pytorchltr_amd import _C
    _run("hinge", 512, 512, 700, 0, _C.PLAN_PARTS, busy=True)


def test_c5_full_size_general_kernel_all_rows():
    """The same step on the general kernel (features read twice): LTR_DISABLE_PARTS is read once per
    process, so the general kernel is reached through the shape rule instead -- rows of 220 features."""
    from pytorchltr_amd import _C
    _run("hinge", 512, 512, 220, 0, _C.PLAN_GENERAL)


@pytest.mark.parametrize("shape", [("hinge", 48, 2000, 64), ("dcg_hinge", 20, 4000, 32), ("logistic", 300, 512, 700),
                                   ("arp1", 400, 600, 512), ("arp2", 600, 300, 448),
                                   # round 4: the rank-dependent kinds (every part ranks the whole query itself)
                                   ("ndcg2", 512, 512, 700), ("ndcg1", 512, 512, 700), ("ndcg2", 300, 300, 448),
                                   ("ndcg1", 150, 1000, 512), ("ndcg2", 90, 768, 640), ("ndcg2", 300, 400, 700),
                                   ("ndcg2", 128, 1000, 136), ("ndcg1", 100, 700, 220), ("ndcg2", 100, 400, 64),
                                   # short lists on wide rows (Yahoo-shaped: the general kernel reads the features twice)
                                   ("hinge", 600, 128, 700), ("logistic", 520, 256, 640), ("arp1", 300, 100, 512),
                                   # round 5: the LDS landing buffer (logistic / LambdaARP at two workgroups per CU: the next
                                   # part's first sweeps by LDS-DMA) -- persistent workgroups with many parts each, a DIRECT
                                   # launch (one part per workgroup: the sweeps still go through LDS), rows of one / two /
                                   # three column vectors per lane, a part of fewer rows than the landing buffer holds
                                   ("arp2", 512, 512, 700), ("logistic", 100, 512, 700), ("arp1", 700, 400, 256),
                                   ("logistic", 640, 300, 512), ("arp2", 900, 70, 700)])
def test_parts_kernel_shapes_all_rows(shape):
    """Long lists (beyond the symmetric pass) and wide rows, every kind, all rows."""
    from pytorchltr_amd import _C
    kind, B, L, F = shape
    _run(kind, B, L, F, 4, _C.PLAN_PARTS)


def test_parts_kernel_reuses_a_workspace_across_batches_and_shapes():
    """The control block and the tagged score granules live in the library's exchange area of the stream and
    reset themselves: the same area serves different batches, then a different shape (other offsets inside
    it), then the first shape again -- each step against the oracle; a stale granule or counter from an
    earlier launch would show up here.  The caller's workspace holds only outputs and may start as garbage."""
    from pytorchltr_amd import _C
    dev = _dev()
    lib = _C.lib()
    shapes = [(70, 512, 700), (70, 512, 700), (33, 2000, 64), (70, 512, 700), (200, 400, 512), (33, 2000, 64)]
    nbytes = max(lib.ltr_linear_workspace_bytes(*sh) for sh in shapes)
    ws = torch.empty(nbytes // 4 + 64, device=dev).fill_(float("nan"))
    for i, (B, L, F) in enumerate(shapes):
        assert lib.ltr_linear_fused_plan(_C.HINGE, B, L, F) == _C.PLAN_PARTS
        s, y, n, X, W, b = synth(B, L, 100 + i, F=F)
        Xd, Wd, bd, yd, nd = X.to(dev), W.to(dev), b.to(dev), y.to(dev), n.to(dev)
        loss = torch.empty(B, device=dev)
        dW, db = torch.empty(F, device=dev), torch.empty(1, device=dev)
        for rep in range(2):
            _C.check(lib.ltr_linear_pairwise_f32(_C.HINGE, 1.0, Xd.data_ptr(), Wd.data_ptr(), bd.data_ptr(),
                                                 yd.data_ptr(), _C.LABEL_I64, nd.data_ptr(), None, B, L, F,
                                                 loss.data_ptr(), None, dW.data_ptr(), db.data_ptr(),
                                                 ws.data_ptr(), nbytes, _C.stream_of(Xd)))
        want_l, _, want_dW, want_db = O.linear_pairwise("hinge", X.numpy(), W.numpy(), float(b[0]), y.numpy(),
                                                        n.numpy(), np.full(B, 1.0 / B))
        assert np.allclose(loss.cpu().numpy(), want_l, rtol=5e-4, atol=1e-5), (i, B, L, F)
        assert np.max(np.abs(dW.cpu().numpy() - want_dW)) < 2e-4 * max(1.0, float(np.max(np.abs(want_dW))))
    _C.device_status()


def test_parts_kernel_under_graph_replay():
    """hipGraph replay freezes the kernel arguments: the tag of the score granules must come from device
    state (the epoch in the control block), or a replay would accept the previous replay's scores."""
    from pytorchltr_amd import _C
    dev = _dev()
    lib = _C.lib()
    B, L, F = 80, 512, 700
    assert lib.ltr_linear_fused_plan(_C.DCG_HINGE, B, L, F) == _C.PLAN_PARTS
    s, y, n, X, W, b = synth(B, L, 31, F=F)
    Xd, Wd, bd, yd, nd = X.to(dev), W.to(dev), b.to(dev), y.to(dev), n.to(dev)
    nbytes = lib.ltr_linear_workspace_bytes(B, L, F)
    ws = torch.empty(nbytes // 4 + 64, device=dev)
    loss = torch.empty(B, device=dev)
    dW, db = torch.empty(F, device=dev), torch.empty(1, device=dev)

    def step():
        _C.check(lib.ltr_linear_pairwise_f32(_C.DCG_HINGE, 1.0, Xd.data_ptr(), Wd.data_ptr(), bd.data_ptr(),
                                             yd.data_ptr(), _C.LABEL_I64, nd.data_ptr(), None, B, L, F,
                                             loss.data_ptr(), None, dW.data_ptr(), db.data_ptr(),
                                             ws.data_ptr(), nbytes, _C.stream_of(Xd)))
    step()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    for rep in range(3):
        # new weights for every replay (same addresses): the scores change, stale granules would not
        Wd.copy_(W.to(dev) * (1.0 + rep))
        g.replay()
        torch.cuda.synchronize()
        want_l, _, want_dW, _ = O.linear_pairwise("dcg_hinge", X.numpy(), W.numpy() * (1.0 + rep), float(b[0]),
                                                  y.numpy(), n.numpy(), np.full(B, 1.0 / B))
        assert np.allclose(loss.cpu().numpy(), want_l, rtol=5e-4, atol=1e-5), rep
        assert np.max(np.abs(dW.cpu().numpy() - want_dW)) < 2e-4 * max(1.0, float(np.max(np.abs(want_dW)))), rep
    _C.device_status()


def test_exchange_area_tag_wraps_around():
    """The granules of the parts kernel carry a 32-bit tag that counts the launches on an exchange area (library-owned
    device memory, include/ltr_hip.h: ltr_exchange_release); the launch that uses the last tag, 2^32 - 1, zeroes the
    granule buffers on its way out and the next one starts over at 1.  ltr_debug_set_exchange_tag puts the area three
    launches before the wrap: every step on either side of it must match the oracle (a stale granule that carried a
    reused tag would be taken for a fresh score)."""
    from pytorchltr_amd import _C
    from pytorchltr_amd.fused import linear_loss_step
    dev = _dev()
    lib = _C.lib()
    B, L, F = 70, 512, 700
    assert lib.ltr_linear_fused_plan(_C.HINGE, B, L, F) == _C.PLAN_PARTS
    s, y, n, X, W, b = synth(B, L, 77, F=F)
    Xd, bd, yd, nd = X.to(dev), b.to(dev), y.to(dev), n.to(dev)
    linear_loss_step(Xd, W.to(dev), bd, yd, nd, loss="hinge")          # (the area exists)
    _C.check(lib.ltr_debug_set_exchange_tag(_C.stream_of(Xd), 0xFFFFFFFD))
    for rep in range(6):
        Wr = W * (1.0 + 0.25 * rep)
        loss, dW, db = linear_loss_step(Xd, Wr.to(dev), bd, yd, nd, loss="hinge")
        want_l, _, want_dW, _ = O.linear_pairwise("hinge", X.numpy(), Wr.numpy(), float(b[0]), y.numpy(), n.numpy(),
                                                  np.full(B, 1.0 / B))
        assert np.allclose(loss.cpu().numpy(), want_l, rtol=5e-4, atol=1e-5), rep
        assert np.max(np.abs(dW.cpu().numpy() - want_dW)) < 2e-4 * max(1.0, float(np.max(np.abs(want_dW)))), rep
    _C.device_status()


def test_parts_kernel_captured_without_a_warm_up_launch():
    """A call made under stream capture gets an exchange area of its own (allocated with the thread's capture mode
    relaxed): a graph captured as the very first use of a shape replays correctly, next to eager launches of the same
    shape on the same stream."""
    from pytorchltr_amd import _C
    dev = _dev()
    lib = _C.lib()
    B, L, F = 90, 520, 704
    assert lib.ltr_linear_fused_plan(_C.HINGE, B, L, F) == _C.PLAN_PARTS
    s, y, n, X, W, b = synth(B, L, 41, F=F)
    Xd, Wd, bd, yd, nd = X.to(dev), W.to(dev), b.to(dev), y.to(dev), n.to(dev)
    nbytes = lib.ltr_linear_workspace_bytes(B, L, F)
    ws = torch.full((nbytes // 4 + 64,), float("nan"), device=dev)     # (the caller's workspace may hold anything)
    loss = torch.empty(B, device=dev)
    dW, db = torch.empty(F, device=dev), torch.empty(1, device=dev)

    def step():
        _C.check(lib.ltr_linear_pairwise_f32(_C.HINGE, 1.0, Xd.data_ptr(), Wd.data_ptr(), bd.data_ptr(),
                                             yd.data_ptr(), _C.LABEL_I64, nd.data_ptr(), None, B, L, F,
                                             loss.data_ptr(), None, dW.data_ptr(), db.data_ptr(),
                                             ws.data_ptr(), nbytes, _C.stream_of(Xd)))
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    want_l, _, want_dW, _ = O.linear_pairwise("hinge", X.numpy(), W.numpy(), float(b[0]), y.numpy(), n.numpy(),
                                              np.full(B, 1.0 / B))
    for rep in range(3):
        loss.fill_(float("nan"))
        g.replay()
        torch.cuda.synchronize()
        assert np.allclose(loss.cpu().numpy(), want_l, rtol=5e-4, atol=1e-5), rep
        assert np.max(np.abs(dW.cpu().numpy() - want_dW)) < 2e-4 * max(1.0, float(np.max(np.abs(want_dW)))), rep
        step()                                                           # an eager launch in between
        torch.cuda.synchronize()
        assert np.allclose(loss.cpu().numpy(), want_l, rtol=5e-4, atol=1e-5), rep
    _C.device_status()


def test_c5_shard_of_8_gpus():
    from pytorchltr_amd import _C
    plan = _C.lib().ltr_linear_fused_plan(_C.HINGE, 64, 512, 700)
    _run("hinge", 64, 512, 700, 2, plan)


@pytest.mark.parametrize("shape", [(32, 1000, 220, "cluster"), (300, 512, 700, "parts")])
def test_cluster_wait_timeout_is_an_error_not_a_silent_nan(shape):
    """ltr_debug_force_timeout makes every in-launch wait give up: the step's outputs are poisoned
    AND the sticky status word turns the next ltr_linear_* call into LTR_ERR_TIMEOUT.  Both kernels
    whose workgroups wait for each other: the cluster kernel and the parts kernel.  Clearing the status marks
    the exchange areas dirty: they are zeroed before the next launch, which must be bit-identical to the first."""
    from pytorchltr_amd import _C
    from pytorchltr_amd.fused import linear_loss_step
    dev = _dev()
    B, L, F, which = shape
    s, y, n, X, W, b = synth(B, L, 3, F=F)
    args = (X.to(dev), W.to(dev), b.to(dev), y.to(dev), n.to(dev))
    assert _C.lib().ltr_linear_fused_plan(_C.HINGE, B, L, F) == (_C.PLAN_CLUSTER if which == "cluster" else _C.PLAN_PARTS)
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
    if which == "cluster":
        multi = n.numpy() > 0
        assert np.all(np.isnan(loss.cpu().numpy()[multi])), "a failed wait must poison the loss"
    else:
        # (a workgroup is poisoned from its first failed wait on: at least the queries of several parts
        # whose scores had not all arrived; single-part queries never wait)
        assert np.any(np.isnan(loss.cpu().numpy())), "a failed wait must poison the loss"
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


@pytest.mark.parametrize("shape", [("hinge", 100, 2000, 64), ("dcg_hinge", 60, 2048, 32), ("hinge", 300, 1000, 512),
                                   ("dcg_hinge", 280, 700, 700)])
def test_parts_kernel_ranked_hinge_all_rows(shape):
    """The hinge kinds on long lists take the counting formulation (buckets by score, per-grade count and sum
    tables, exact visits of the edge buckets) instead of the O(n^2) pair pass: every row against the oracle --
    the gradients are integers and must come out exactly, the loss to the long-list tolerance -- run twice,
    bit-identical.  Lists straddle the per-query threshold (rows * n), so both formulations are in the batch."""
    from pytorchltr_amd import _C
    kind, B, L, F = shape
    _run(kind, B, L, F, 6, _C.PLAN_PARTS)


@pytest.mark.parametrize("variant", ["float_labels", "grade_7", "tied_scores", "tiny_spread", "constant"])
def test_parts_kernel_ranked_hinge_falls_back(variant):
    """Queries the counting formulation does not take (labels that are not the integers 0..4, scores without
    spread) fall back to the pair pass, per query, and queries full of tied scores stay exact."""
    from pytorchltr_amd import _C
    from pytorchltr_amd.fused import linear_loss_step
    dev = _dev()
    B, L, F = 40, 2000, 64
    assert _C.lib().ltr_linear_fused_plan(_C.HINGE, B, L, F) == _C.PLAN_PARTS
    s, y, n, X, W, b = synth(B, L, 9, F=F)
    n = torch.clamp(n, min=1500)
    yy = y.clone()
    if variant == "float_labels":
        yy = y.float() + 0.5 * (torch.arange(B)[:, None] % 2)          # every other query: half-integer labels
    elif variant == "grade_7":
        yy[::3] = yy[::3] + 3                                            # grades up to 7 in a third of the queries
    elif variant == "tied_scores":
        X = torch.round(X * 2) / 2                                       # few distinct feature values ...
        W = torch.round(W * 8) / 8                                       # ... and weights: many exactly equal scores
    elif variant == "tiny_spread":
        X = X * 1e-7
    elif variant == "constant":
        X = torch.zeros_like(X)
    kind = "hinge"
    loss, dW, db = linear_loss_step(X.to(dev), W.to(dev), b.to(dev), yy.to(dev), n.to(dev), loss=kind)
    want_l, _, want_dW, want_db = O.linear_pairwise(kind, X.numpy(), W.numpy(), float(b[0]), yy.numpy(), n.numpy(),
                                                    np.full(B, 1.0 / B))
    assert np.allclose(loss.cpu().numpy(), want_l, rtol=5e-4, atol=1e-5), variant
    tol = 2e-4 * max(1.0, float(np.max(np.abs(want_dW))))
    assert np.max(np.abs(dW.cpu().numpy() - want_dW)) < tol, variant
    _C.device_status()
