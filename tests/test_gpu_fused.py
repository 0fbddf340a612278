"""GPU parity of the fused Linear(F,1)+loss kernel and of the drop-in training loop."""
import numpy as np
import pytest
import torch

from oracle import ltr_oracle as O
from tests.conftest import assert_rank_dependent_losses, load_golden, synth

pytestmark = pytest.mark.gpu
G = load_golden()
KINDS = list(O.KINDS)


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _check(kind, B, L, F, seed, sigma=1.0, full=False, grad_out=None):
    from pytorchltr_amd.fused import linear_loss_step
    dev = _dev()
    s, y, n, X, W, b = synth(B, L, seed, F=F)
    if full:
        n = torch.full_like(n, L)
    from pytorchltr_amd import loss as L_
    mod = {"hinge": L_.PairwiseHingeLoss, "dcg_hinge": L_.PairwiseDCGHingeLoss,
           "logistic": L_.PairwiseLogisticLoss, "arp1": L_.LambdaARPLoss1,
           "arp2": L_.LambdaARPLoss2, "ndcg1": L_.LambdaNDCGLoss1, "ndcg2": L_.LambdaNDCGLoss2}[kind]
    loss_mod = mod() if kind in ("hinge", "dcg_hinge") else mod(sigma)
    go = None if grad_out is None else grad_out.to(dev)
    loss, dW, db, scores = linear_loss_step(X.to(dev), W.to(dev), b.to(dev), y.to(dev), n.to(dev),
                                            loss=loss_mod, grad_out=go, return_scores=True)
    gout = np.full(B, 1.0 / B) if grad_out is None else grad_out.numpy().astype(np.float64)
    want_l, want_s, want_dW, want_db = O.linear_pairwise(kind, X.numpy(), W.numpy(), float(b[0]),
                                                         y.numpy(), n.numpy(), gout, sigma=sigma)
    assert np.allclose(scores.cpu().numpy(), want_s, rtol=1e-5, atol=1e-5)
    rtol = 5e-4 if L > 256 else 2e-5
    assert np.allclose(loss.cpu().numpy(), want_l, rtol=rtol, atol=1e-5), kind
    tol = 2e-5 * max(1.0, float(np.max(np.abs(want_dW)))) * (10 if L > 256 else 1)
    assert np.max(np.abs(dW.cpu().numpy() - want_dW)) < tol, kind
    assert abs(float(db.cpu()[0]) - want_db) < tol, kind
    # without the score output the single-pass register-tile kernel is eligible
    loss2, dW2, db2 = linear_loss_step(X.to(dev), W.to(dev), b.to(dev), y.to(dev), n.to(dev),
                                       loss=loss_mod, grad_out=go)
    assert np.allclose(loss2.cpu().numpy(), want_l, rtol=rtol, atol=1e-5), kind
    assert np.max(np.abs(dW2.cpu().numpy() - want_dW)) < tol, kind
    assert abs(float(db2.cpu()[0]) - want_db) < tol, kind


def _random_shapes(count, seed):
    import random
    rnd = random.Random(seed)
    out = []
    for _ in range(count):
        B = rnd.choice([1, 2, 3, 7, 33, 64, 130, 257, 600])
        L = rnd.choice([1, 2, 5, 31, 64, 65, 100, 128, 129, 200, 257, 300, 511, 777, 1000, 1030])
        F = rnd.choice([4, 8, 12, 36, 64, 136, 220, 260, 700])
        if B * L * F > 20_000_000:
            B = max(1, 20_000_000 // (L * F))
        out.append((B, L, F, rnd.choice(list(KINDS))))
    return out


@pytest.mark.parametrize("shape", _random_shapes(48, 20260929), ids=lambda s: "%dx%dx%d-%s" % s)
def test_fused_step_random_shapes(shape):
    """Seeded random (B, L, F, kind): whatever plan the dispatcher picks (register tile, cluster, parts,
    general kernel), every query's loss, the scores, dW and db against the fp64 oracle."""
    B, L, F, kind = shape
    _check(kind, B, L, F, seed=B + L + F)


def _check_ndcg_tolerant(kind, B, L, F):
    """Two nearly tied fp32 scores of a 200-document list can rank the other way round than the fp64 oracle's: such a row
    moves by ~1e-3 and must be a TESTED rank flip (assert_rank_dependent_losses); every row at the stated tolerance."""
    from pytorchltr_amd.fused import linear_loss_step
    dev = _dev()
    for seed, full in ((21, False), (22, True)):
        s_, y, n, X, W, b = synth(B, L, seed, F=F)
        if full:
            n = torch.full_like(n, L)
        loss, dW, db = linear_loss_step(X.to(dev), W.to(dev), b.to(dev), y.to(dev), n.to(dev), loss=kind)
        want_l, want_s, want_dW, want_db = O.linear_pairwise(kind, X.numpy(), W.numpy(), float(b[0]), y.numpy(), n.numpy(), np.full(B, 1.0 / B))
        assert_rank_dependent_losses(kind, loss.cpu().numpy(), X, W, b, y, n, want_l, want_s, rtol=2e-5)
        tol = 2e-4 * max(1.0, float(np.max(np.abs(want_dW))))
        assert np.max(np.abs(dW.cpu().numpy() - want_dW)) < tol and abs(float(db.cpu()[0]) - want_db) < tol


def _round4_shapes(count, seed):
    """Seeded shapes from the regimes whose dispatch changed in round 4 (scripts/dev/fuzz_dispatch.py is the long-running
    version: it found a workspace sized for two kinds only)."""
    import random
    rnd = random.Random(seed)
    out = []
    for _ in range(count):
        regime = rnd.choice(["wide", "wide_short", "rt1024", "ndcg_cluster", "ndcg_parts"])
        kind = rnd.choice(list(KINDS))
        if regime == "wide":
            B, L, F = rnd.choice([65, 130, 257, 300]), rnd.choice([300, 400, 512, 768, 1000]), rnd.choice([448, 512, 576, 640, 700])
        elif regime == "wide_short":
            B, L, F = rnd.choice([512, 520]), rnd.choice([100, 128, 200, 256]), rnd.choice([640, 700])
        elif regime == "rt1024":
            kind = rnd.choice(["hinge", "dcg_hinge", "logistic", "arp1", "arp2"])
            B, L, F = rnd.choice([3, 64, 257]), rnd.choice([129, 150, 181, 200, 255, 256]), rnd.choice([100, 120, 136, 160, 220])
        elif regime == "ndcg_cluster":
            kind = rnd.choice(["ndcg1", "ndcg2"])
            B, L, F = rnd.choice([5, 33, 100, 272, 380]), rnd.choice([257, 300, 512, 700, 1000, 1024]), rnd.choice([16, 64, 136, 220])
        else:
            kind = rnd.choice(["ndcg1", "ndcg2"])
            B, L, F = rnd.choice([70, 128, 190]), rnd.choice([400, 512, 600, 768, 1000]), rnd.choice([448, 512, 640, 700])
        if B * L * F > 70_000_000:
            B = max(1, 70_000_000 // (L * F))
        out.append((B, L, F, kind, rnd.randrange(4)))
    return out


@pytest.mark.parametrize("shape", _round4_shapes(40, 20260930), ids=lambda s: "%dx%dx%d-%s-p%d" % s)
def test_fused_step_round4_dispatch_regimes(shape):
    """Wide rows on the parts kernel (long and short lists, every kind), the 19- / 24-sweep register tiles, the NDCG kinds on
    the cluster kernel: whatever plan the dispatcher picks, losses and gradients against the fp64 oracle, twice
    bit-identical, ragged / full / mixed / short list-length patterns."""
    from pytorchltr_amd import _C
    from pytorchltr_amd.fused import linear_loss_step
    B, L, F, kind, pat = shape
    dev = _dev()
    s, y, n, X, W, b = synth(B, L, B + L + F + pat, F=F)
    g = torch.Generator().manual_seed(B * 7 + L)
    if pat == 1:
        n = torch.full_like(n, L)
    elif pat == 2:
        n = torch.where(torch.rand(B, generator=g) < 0.5, torch.full_like(n, L), n)
    elif pat == 3:
        n = torch.clamp(n, max=max(1, L // 3))
    n[0] = 0
    outs = []
    for rep in range(2):
        loss, dW, db = linear_loss_step(X.to(dev), W.to(dev), b.to(dev), y.to(dev), n.to(dev), loss=kind)
        outs.append((loss.cpu().numpy(), dW.cpu().numpy(), db.cpu().numpy()))
    _C.device_status()
    assert all(np.array_equal(a, c) for a, c in zip(outs[0], outs[1]))
    want_l, want_s, want_dW, want_db = O.linear_pairwise(kind, X.numpy(), W.numpy(), float(b[0]), y.numpy(), n.numpy(), np.full(B, 1.0 / B))
    loss, dW, db = outs[0]
    rtol = 5e-4 if L > 256 else 2e-5
    if kind in ("ndcg1", "ndcg2"):      # (a row off against the fp64 scores must be a tested rank flip of nearly tied fp32 scores)
        assert_rank_dependent_losses(kind, loss, X, W, b, y, n, want_l, want_s, rtol=rtol)
    else:
        assert np.isclose(loss, want_l, rtol=rtol, atol=1e-5).all()
    tol = 4e-4 * max(1.0, float(np.max(np.abs(want_dW))))
    assert np.max(np.abs(dW - want_dW)) < tol and abs(float(db[0]) - want_db) < tol


@pytest.mark.parametrize("kind", KINDS)
def test_fused_step_small_shapes(kind):
    _check(kind, 8, 16, 5, 1234)                 # scalar path (F % 4 != 0), Example3-like F
    _check(kind, 6, 37, 12, 5, sigma=2.0)        # vector path, odd L
    _check(kind, 5, 64, 136, 6)                  # C2 feature width


def test_many_more_queries_than_resident_workgroups():
    """B far above the resident workgroup count (several dispatch rounds per CU)."""
    _check("hinge", 2100, 24, 8, 13)
    _check("ndcg2", 1500, 40, 16, 14)


@pytest.mark.parametrize("kind", ["hinge", "logistic", "ndcg2"])
def test_fused_step_c2_shape(kind):
    _check(kind, 32, 128, 136, 3)
    _check(kind, 1024, 128, 136, 0)
    _check(kind, 16, 128, 136, 4, full=True)
    _check(kind, 16, 128, 136, 4, grad_out=torch.linspace(-0.5, 1.5, 16))


@pytest.mark.parametrize("kind", ["hinge", "ndcg2"])
def test_fused_step_c2_c3_full_lists_full_size(kind):
    """The n == list_len twin of C2 / C3 at FULL size through the fused path (every row against the
    oracle): the worst case for work, and the shape the roofline's `full_lists` figure is quoted on."""
    _check(kind, 1024, 128, 136, 0, full=True)


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("shape", [(150, 200, 220), (64, 216, 220), (70, 182, 220)])
def test_fused_step_24_sweep_register_tile(kind, shape):
    """Round 4: lists of 172 .. 216 documents at Istella's row width (Istella-S reaches 182) on a 24-sweep tile, every
    kind."""
    from pytorchltr_amd import _C
    B, L, F = shape
    assert _C.lib().ltr_linear_fused_plan(O.KINDS[kind], B, L, F) == _C.PLAN_REGISTER_TILE
    if kind in ("ndcg1", "ndcg2"):
        _check_ndcg_tolerant(kind, B, L, F)
        return
    _check(kind, B, L, F, 31)
    _check(kind, min(B, 40), L, F, 32, full=True)
    _check(kind, min(B, 40), L, F, 33, grad_out=torch.linspace(-0.5, 1.5, min(B, 40)))


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("shape", [(300, 256, 136), (260, 200, 136), (150, 162, 220), (90, 256, 100), (64, 129, 220)])
def test_fused_step_19_sweep_register_tile(kind, shape):
    """Round 4: lists of 136 .. 256 documents at MSLR / Istella row widths on the 19-sweep register tile (512 threads,
    two workgroups per CU; features read once instead of the general kernel's two passes), every kind, ragged and
    full lists, with an upstream gradient."""
    from pytorchltr_amd import _C
    B, L, F = shape
    assert _C.lib().ltr_linear_fused_plan(O.KINDS[kind], B, L, F) == _C.PLAN_REGISTER_TILE
    if kind in ("ndcg1", "ndcg2"):
        _check_ndcg_tolerant(kind, B, L, F)
        return
    _check(kind, B, L, F, 21)
    _check(kind, min(B, 40), L, F, 22, full=True)
    _check(kind, min(B, 40), L, F, 23, grad_out=torch.linspace(-0.5, 1.5, min(B, 40)))


@pytest.mark.parametrize("shape", [(6, 1000, 220, "dcg_hinge"), (6, 512, 700, "hinge"),
                                   (4, 300, 64, "ndcg1"), (3, 2000, 16, "arp2")])
def test_fused_step_large_tiles(shape):
    B, L, F, kind = shape
    _check(kind, B, L, F, 8)                     # tile does not fit LDS: re-read path


@pytest.mark.parametrize("shape", [(5, 8, 2048, "hinge"), (5, 12, 2048, "logistic"),
                                   (5, 16, 1024, "ndcg2"), (4, 18, 1024, "ndcg1")])
def test_fused_step_rows_as_wide_as_the_workgroup(shape):
    """A row with as many 16-byte column vectors as the workgroup has threads (C == T: F = 2048 on the
    512-thread register tile, F = 1024 on the 256-thread NDCG tile): the bias partial must still be
    written (it was stored by thread C, which does not exist there)."""
    B, L, F, kind = shape
    from pytorchltr_amd import _C
    assert _C.lib().ltr_linear_fused_plan(O.KINDS[kind], B, L, F) == _C.PLAN_REGISTER_TILE
    _check(kind, B, L, F, 17)
    _check(kind, B, L, F, 18, grad_out=torch.linspace(-1.0, 2.0, B))


@pytest.mark.parametrize("name", [c["name"] for c in G.by_op("linear_step")])
def test_fused_step_vs_reference_vectors(name):
    """Against what the real reference computed for loss_fn(Linear(X), y, n).mean().backward()."""
    from pytorchltr_amd.fused import linear_loss_step
    dev = _dev()
    case = G.cases[name]
    X, W, b = (torch.as_tensor(G.get(name, f)).to(dev) for f in ("X", "W", "b"))
    y, n = (torch.as_tensor(G.get(name, f)).to(dev) for f in ("relevance", "n"))
    for kind in case["kinds"]:
        loss, dW, db, scores = linear_loss_step(X, W, b, y, n, loss=kind, return_scores=True)
        assert np.allclose(scores.cpu().numpy(), G.get(name, kind + "/scores"), rtol=1e-5, atol=1e-5)
        assert np.allclose(loss.cpu().numpy(), G.get(name, kind + "/loss"), rtol=2e-5, atol=1e-5), kind
        ref_dW = G.get(name, kind + "/dW")
        tol = 3e-5 * max(1.0, float(np.max(np.abs(ref_dW))))
        assert np.max(np.abs(dW.cpu().numpy() - ref_dW)) < tol, kind
        assert abs(float(db.cpu()[0]) - float(G.get(name, kind + "/db")[0])) < tol, kind


def test_loss_sum_and_single_entry_point():
    """ltr_linear_reduce_loss_f32 totals the loss in the reduction launch; the one-call entry
    point ltr_linear_pairwise_f32 gives the same step."""
    from pytorchltr_amd import _C
    from pytorchltr_amd.fused import linear_loss_step
    dev = _dev()
    s, y, n, X, W, b = synth(40, 128, 21, F=136)
    X, W, b, y, n = X.to(dev), W.to(dev), b.to(dev), y.to(dev), n.to(dev)
    loss, dW, db, lsum = linear_loss_step(X, W, b, y, n, loss="logistic", return_loss_sum=True)
    assert float(lsum) == pytest.approx(float(loss.double().sum()), rel=1e-5)
    lib = _C.lib()
    B, L, F = X.shape
    ws_bytes = lib.ltr_linear_workspace_bytes(B, L, F)
    ws = torch.empty(ws_bytes // 4, device=dev)
    loss2, dW2, db2 = torch.empty(B, device=dev), torch.empty(F, device=dev), torch.empty(1, device=dev)
    _C.check(lib.ltr_linear_pairwise_f32(_C.LOGISTIC, 1.0, X.data_ptr(), W.data_ptr(), b.data_ptr(),
                                         y.data_ptr(), _C.LABEL_I64, n.data_ptr(), None, B, L, F,
                                         loss2.data_ptr(), None, dW2.data_ptr(), db2.data_ptr(),
                                         ws.data_ptr(), ws_bytes, _C.stream_of(X)))
    assert torch.equal(loss, loss2) and torch.equal(dW, dW2) and torch.equal(db, db2)
    assert lib.ltr_linear_pairwise_f32(_C.LOGISTIC, 1.0, X.data_ptr(), W.data_ptr(), b.data_ptr(),
                                       y.data_ptr(), _C.LABEL_I64, n.data_ptr(), None, B, L, F,
                                       loss2.data_ptr(), None, dW2.data_ptr(), db2.data_ptr(),
                                       ws.data_ptr(), 16, _C.stream_of(X)) == -5      # workspace too small


@pytest.mark.parametrize("B,F", [(4096, 136), (5003, 136), (70001, 46), (4100, 7), (9000, 1020), (4500, 1024)])
def test_reduction_of_many_rows(B, F):
    """Batches beyond 4096 queries: the reduction of the partial rows is two launches whose work scales with B (chunks of whole
    rows per workgroup, then one combine per column group -- csrc/ltr_linear.inc: launch_linear_reduce) instead of one workgroup
    per column walking all of them.  Same contract as the small-batch kernel: dW / db / loss sum, weighted rows, accumulate,
    deterministic.  (F = 1024: rows too wide for the split kernel's mapping -- the one-level kernel, at any B.)"""
    from pytorchltr_amd import _C
    dev = _dev()
    lib = _C.lib()
    g = torch.Generator().manual_seed(B + F)
    PF = (F + 4) & ~3
    ws_bytes = lib.ltr_linear_workspace_bytes(B, 8, F)
    assert ws_bytes >= B * PF * 4
    ws = torch.full((ws_bytes // 4,), float("nan"), device=dev)          # (the scratch behind the rows starts as garbage)
    rows = torch.randn(B, PF, generator=g)
    rows[:, F + 1:] = 0.0
    ws[:B * PF] = rows.reshape(-1).to(dev)
    go = torch.rand(B, generator=g)
    loss = torch.rand(B, generator=g)
    go_d, loss_d = go.to(dev), loss.to(dev)
    ref = (rows.double() * go.double()[:, None]).sum(0)
    scale = float(np.sqrt(B))
    outs = []
    for rep in range(2):
        dW, db, ls = torch.zeros(F, device=dev), torch.zeros(1, device=dev), torch.zeros(1, device=dev)
        _C.check(lib.ltr_linear_reduce_loss_f32(ws.data_ptr(), go_d.data_ptr(), loss_d.data_ptr(), B, F, dW.data_ptr(), db.data_ptr(),
                                                ls.data_ptr(), _C.stream_of(ws)))
        outs.append((dW.cpu(), db.cpu(), ls.cpu()))
    dW, db, ls = outs[0]
    assert all(torch.equal(a, b) for a, b in zip(outs[0], outs[1]))       # fixed order: bit-identical run to run
    assert np.max(np.abs(dW.double().numpy() - ref[:F].numpy())) < 2e-6 * scale * 4
    assert abs(float(db[0]) - float(ref[F])) < 2e-6 * scale * 4
    assert float(ls[0]) == pytest.approx(float(loss.double().sum()), rel=1e-5)
    # uniform weight 1/B (grad_out NULL), accumulated on top of what is there
    dW2, db2, ls2 = dW.to(dev).clone(), db.to(dev).clone(), ls.to(dev).clone()
    _C.check(lib.ltr_linear_reduce_accum_f32(ws.data_ptr(), None, loss_d.data_ptr(), B, F, dW2.data_ptr(), db2.data_ptr(), ls2.data_ptr(),
                                             1, _C.stream_of(ws)))
    ref2 = ref + rows.double().mean(0)
    assert np.max(np.abs(dW2.cpu().double().numpy() - ref2[:F].numpy())) < 2e-6 * scale * 4
    assert abs(float(db2.cpu()[0]) - float(ref2[F])) < 2e-6 * scale * 4
    assert float(ls2.cpu()[0]) == pytest.approx(2 * float(loss.double().sum()), rel=1e-5)
    # the upstream gradient as one device scalar (stride 0)
    sc = torch.full((1,), 0.25, device=dev)
    dW3, db3 = torch.empty(F, device=dev), torch.empty(1, device=dev)
    _C.check(lib.ltr_linear_reduce_bcast_f32(ws.data_ptr(), sc.data_ptr(), B, F, dW3.data_ptr(), db3.data_ptr(), _C.stream_of(ws)))
    ref3 = rows.double().sum(0) * 0.25
    assert np.max(np.abs(dW3.cpu().double().numpy() - ref3[:F].numpy())) < 2e-6 * scale * 4
    assert abs(float(db3.cpu()[0]) - float(ref3[F])) < 2e-6 * scale * 4


def test_fused_module_matches_unfused_dropin():
    """FusedLinearLoss == loss_fn(nn.Linear(F,1)(xs), ys, n) for arbitrary upstream weights."""
    from pytorchltr_amd.fused import FusedLinearLoss
    from pytorchltr_amd.loss import LambdaNDCGLoss2
    dev = _dev()
    s, y, n, X, W, b = synth(24, 50, 11, F=20)
    lin = torch.nn.Linear(20, 1).to(dev)
    fused = FusedLinearLoss(20, LambdaNDCGLoss2(sigma=0.7)).to(dev)
    fused.load_state_dict(lin.state_dict())                   # state_dict compatible with nn.Linear
    w = torch.rand(24, device=dev)
    (LambdaNDCGLoss2(sigma=0.7)(lin(X.to(dev)), y.to(dev), n.to(dev)) * w).sum().backward()
    out, sc = fused(X.to(dev), y.to(dev), n.to(dev), return_scores=True)
    (out * w).sum().backward()
    assert torch.allclose(sc, lin(X.to(dev)).reshape(24, 50).detach(), rtol=1e-5, atol=1e-5)
    assert torch.allclose(fused.weight.grad, lin.weight.grad, rtol=1e-4, atol=1e-5)
    assert torch.allclose(fused.bias.grad, lin.bias.grad, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("kind", ["hinge", "logistic", "ndcg2"])
def test_unchanged_user_script_reaches_the_fused_kernel(kind):
    """`loss_fn(model(xs), ys, n).mean().backward()` with model = use_linear_scorer(nn.Linear(F, 1))
    -- the reference's loop body unchanged (examples/01-basic-usage.py:66-75) -- runs scores, loss and weight gradient
    as ONE pass over the features: the layer returns LazyScores, the loss module recognises them.  Same loss and
    gradients as the plain nn.Linear composition; and anybody else who touches the scores gets real ones."""
    from pytorchltr_amd import loss as L_
    from pytorchltr_amd.evaluation import ndcg
    from pytorchltr_amd.fused import LazyScores, use_linear_scorer
    dev = _dev()
    B, L, F = 64, 100, 136
    s, y, n, X, W, b = synth(B, L, 31, F=F)
    Xd, yd, nd = X.to(dev), y.to(dev), n.to(dev)
    mk = {"hinge": L_.PairwiseHingeLoss, "logistic": lambda: L_.PairwiseLogisticLoss(0.8), "ndcg2": lambda: L_.LambdaNDCGLoss2(0.8)}[kind]
    loss_fn = mk()
    lin = torch.nn.Linear(F, 1).to(dev)
    model = use_linear_scorer(torch.nn.Linear(F, 1).to(dev))
    model.load_state_dict(lin.state_dict())
    w = torch.rand(B, device=dev)
    ref = loss_fn(lin(Xd), yd, nd)
    (ref * w).sum().backward()
    scores = model(Xd)
    assert type(scores) is LazyScores and scores._real is None
    assert tuple(scores.shape) == (B, L, 1) and scores.dtype is torch.float32 and scores.is_cuda and scores.requires_grad
    out = loss_fn(scores, yd, nd)
    assert scores._real is None                                  # the scores were never written: one fused pass
    (out * w).sum().backward()
    assert torch.allclose(out, ref, rtol=2e-5, atol=1e-5)
    tol = 1e-5 * max(1.0, float(lin.weight.grad.abs().max()))     # (the bias gradient of a pairwise loss is 0 up to rounding)
    assert torch.allclose(model.weight.grad, lin.weight.grad, rtol=1e-4, atol=tol)
    assert torch.allclose(model.bias.grad, lin.bias.grad, rtol=1e-4, atol=tol)
    # `.mean().backward()`, the literal user line, a second time (gradients accumulate like nn.Linear's)
    model.zero_grad()
    lin.zero_grad()
    loss_fn(model(Xd), yd, nd).mean().backward()
    loss_fn(lin(Xd), yd, nd).mean().backward()
    assert torch.allclose(model.weight.grad, lin.weight.grad, rtol=1e-4, atol=1e-5 * max(1.0, float(lin.weight.grad.abs().max())))
    # used by anything else, the scores are real -- and stay connected to the parameters
    sc = model(Xd)
    val = ndcg(sc, yd, nd, k=10)                                 # a metric of this package
    assert sc._real is not None and torch.allclose(sc.materialize(), lin(Xd), rtol=1e-5, atol=1e-5)
    assert torch.allclose(val, ndcg(lin(Xd), yd, nd, k=10), rtol=1e-6)
    sc2 = model(Xd)
    model.zero_grad()
    lin.zero_grad()
    (torch.tanh(sc2).sum() + loss_fn(sc2, yd, nd).sum()).backward()    # scores used twice: computed once, both paths right
    (torch.tanh(lin(Xd)).sum() + loss_fn(lin(Xd), yd, nd).sum()).backward()
    assert torch.allclose(model.weight.grad, lin.weight.grad, rtol=1e-4, atol=1e-5 * float(lin.weight.grad.abs().max()))
    with torch.no_grad():                                        # evaluation: plain scores, nothing lazy
        assert type(model(Xd)) is torch.Tensor
    # a converted layer fed something that is not an fp32 feature batch behaves like nn.Linear
    with pytest.raises(RuntimeError):
        model(Xd.double())                                       # nn.Linear's own dtype error, not a silent fp32 cast
    assert model(Xd[:, :, :F].reshape(B * L, F)).shape == (B * L, 1)   # 2-D input: the plain layer


def test_example3_training_trace():
    """BASELINE.json configs[0]: examples/01-basic-usage.py on the Example3 toy data.  Linear(5,1),
    PairwiseHingeLoss, SGD lr 0.1, batch 2; published trace: test nDCG@10 0.8617 at start."""
    from pytorchltr_amd.evaluation import ndcg
    from pytorchltr_amd.loss import PairwiseHingeLoss
    dev = _dev()
    name = "c1_example3_step"
    X, W, b = (torch.as_tensor(G.get(name, f)).to(dev) for f in ("X", "W", "b"))
    y, n = (torch.as_tensor(G.get(name, f)).to(dev) for f in ("relevance", "n"))
    model = torch.nn.Linear(5, 1).to(dev)
    with torch.no_grad():
        model.weight.copy_(W.reshape(1, 5))
        model.bias.copy_(b)
    loss = PairwiseHingeLoss()(model(X), y, n)
    assert loss.detach().cpu().numpy() == pytest.approx([4.6077, 1.0157], abs=2e-4)
    loss.mean().backward()
    assert model.weight.grad.cpu().numpy().reshape(-1) == pytest.approx(
        [-2.5, 0.0, 0.0, 0.16667, -1.0], abs=1e-4)
    assert float(model.bias.grad.cpu()) == pytest.approx(0.0, abs=1e-6)
    s, yt, nt = (torch.as_tensor(a).to(dev) for a in G.inputs("c1_example3_eval"))
    assert float(ndcg(s, yt, nt, k=10).cpu()) == pytest.approx(0.8617, abs=1e-4)


def test_sgd_learning_improves_arp():
    """The reference's only end-to-end test (tests/test_integration.py:18-63): SGD on a linear
    scorer with PairwiseHingeLoss must reduce ARP substantially.  Synthetic separable data."""
    from pytorchltr_amd.evaluation import arp
    from pytorchltr_amd.loss import PairwiseHingeLoss
    dev = _dev()
    torch.manual_seed(42)
    g = torch.Generator().manual_seed(42)
    B, L, F = 16, 30, 8
    X = torch.randn(B, L, F, generator=g)
    true_w = torch.randn(F, generator=g)
    y = ((X @ true_w) > 0.5).long() + ((X @ true_w) > 1.5).long()
    n = torch.randint(10, L + 1, (B,), generator=g)
    X, y, n = X.to(dev), y.to(dev), n.to(dev)
    model = torch.nn.Linear(F, 1).to(dev)
    opt = torch.optim.SGD(model.parameters(), lr=0.01)
    loss_fn = PairwiseHingeLoss()
    with torch.no_grad():
        start = float(arp(model(X), y, n).mean())
    for _ in range(100):
        opt.zero_grad()
        loss_fn(model(X), y, n).mean().backward()
        opt.step()
    with torch.no_grad():
        end = float(arp(model(X), y, n).mean())
    assert end - start <= -0.40


@pytest.mark.parametrize("B", [289, 300, 517, 1000, 1023, 1024])
@pytest.mark.parametrize("lists", ["ragged", "full", "empty_but_one", "two_lengths", "steps_of_16"])
def test_list_length_scheduling_visits_every_query_once(B, lists):
    """With 1.125 x #CUs < B <= 4 x #CUs the register-tile kernel maps block ids to queries through
    the in-kernel list-length order.  B workgroups write B loss entries: every entry written
    (no NaN left from the prefill) means the mapping is a bijection; the values must not depend on
    it (compared with the unscheduled general kernel, which the score output selects)."""
    from pytorchltr_amd import _C
    from pytorchltr_amd.fused import linear_loss_step
    dev = _dev()
    L, F = 128, 136
    s, y, n, X, W, b = synth(B, L, 77, F=F)
    if lists == "full":
        n = torch.full_like(n, L)
    elif lists == "empty_but_one":
        n = torch.zeros_like(n)
        n[B // 2] = 3
    elif lists == "two_lengths":
        n = torch.where(torch.arange(B) % 3 == 0, torch.full_like(n, L), torch.full_like(n, 2))
    elif lists == "steps_of_16":
        n = (n // 16) * 16
    X, W, b, y, n = X.to(dev), W.to(dev), b.to(dev), y.to(dev), n.to(dev)
    lib = _C.lib()
    loss = torch.full((B,), float("nan"), device=dev)
    part = torch.full((lib.ltr_linear_workspace_bytes(B, L, F) // 4,), float("nan"), device=dev)
    _C.check(lib.ltr_linear_partials_f32(0, 1.0, X.data_ptr(), W.data_ptr(), b.data_ptr(), y.data_ptr(),
                                         _C.label_dtype(y), n.data_ptr(), B, L, F, loss.data_ptr(), None,
                                         part.data_ptr(), _C.stream_of(X)))
    torch.cuda.synchronize()
    assert not torch.isnan(loss).any()
    assert not torch.isnan(part[:B * (F + 1)]).any()
    want, dW_want, db_want, _ = linear_loss_step(X, W, b, y, n, loss="hinge", return_scores=True)
    assert torch.allclose(loss, want, rtol=2e-5, atol=1e-5)
    got, dW, db = linear_loss_step(X, W, b, y, n, loss="hinge")
    assert torch.equal(got, loss)
    # two fp32 kernels with different summation orders: twice the 2e-5 bound each holds vs fp64
    tol = 5e-5 * max(1.0, float(dW_want.abs().max()))
    assert float((dW - dW_want).abs().max()) < tol and abs(float(db - db_want)) < tol


@pytest.mark.parametrize("kind", ["hinge", "logistic", "ndcg1"])
def test_list_length_scheduling_in_the_general_fused_kernel(kind):
    """The general (re-read) fused kernel under the list-length order: long lists with more than
    one workgroup per CU, and F % 4 != 0 at B >= 4 x #CUs."""
    _check(kind, 300, 300, 64, 9)
    _check(kind, 1100, 128, 6, 10)
    _check(kind, 310, 260, 12, 11, full=True)


@pytest.mark.parametrize("kind", ["hinge", "dcg_hinge", "logistic", "arp1", "arp2", "ndcg1", "ndcg2"])
@pytest.mark.parametrize("shape", [(5, 700, 136), (33, 1000, 220), (16, 512, 700), (64, 300, 64)])
def test_cluster_kernel_long_lists_on_small_batches(kind, shape):
    """Long lists on a small batch: a query is spread over a cluster of workgroups that keep its
    rows in registers and exchange scores / gradient slices through device memory (features read
    once).  Same step as the oracle's, including queries that are empty, have one document, end
    exactly at / just past a workgroup's row range, or are full; and bit-identical from run to run."""
    from pytorchltr_amd import _C
    from pytorchltr_amd.fused import linear_loss_step
    B, L, F = shape
    lib = _C.lib()
    kid = getattr(_C, kind.upper())
    plan = lib.ltr_linear_fused_plan(kid, B, L, F)
    assert plan == _C.PLAN_CLUSTER
    dev = _dev()
    s, y, n, X, W, b = synth(B, L, 31, F=F)
    if F // 4 <= 128:
        R = 512 // (F // 4)
        rpw = 8 * R
        if -(-L // rpw) > 16 or (-(-L // rpw) > 6 and 4 * B > 256):
            rpw = 12 * R
    else:
        rpw = 19 * (1024 // (F // 4))
    n[:5] = torch.tensor([0, 1, min(L, rpw), min(L, rpw + 1), L])[:min(5, B)]
    gout = torch.linspace(0.2, 1.7, B)
    Xd, Wd, bd, yd, nd = X.to(dev), W.to(dev), b.to(dev), y.to(dev), n.to(dev)
    loss, dW, db = linear_loss_step(Xd, Wd, bd, yd, nd, loss=kind, grad_out=gout.to(dev))
    want_l, _, want_dW, want_db = O.linear_pairwise(kind, X.numpy(), W.numpy(), float(b[0]), y.numpy(),
                                                    n.numpy(), gout.numpy().astype(np.float64), sigma=1.0)
    rtol = 5e-4 if L > 256 else 2e-5
    assert np.allclose(loss.cpu().numpy(), want_l, rtol=rtol, atol=1e-5), kind
    tol = 2e-4 * max(1.0, float(np.max(np.abs(want_dW))))
    assert np.max(np.abs(dW.cpu().numpy() - want_dW)) < tol, kind
    assert abs(float(db.cpu()[0]) - want_db) < tol, kind
    loss2, dW2, db2 = linear_loss_step(Xd, Wd, bd, yd, nd, loss=kind, grad_out=gout.to(dev))
    assert torch.equal(loss, loss2) and torch.equal(dW, dW2) and torch.equal(db, db2)
    # Round 5: the members of a query sit on block ids congruent mod 8 (one XCD) and hand each other plain stores
    # through that XCD's L2 once the launch has CHECKED the placement; a cluster found spread over several XCDs
    # re-publishes its scores write-through and keeps to the round-4 protocol.  Forced here: the same bits.
    lib.ltr_debug_cluster_mode(1)
    try:
        loss3, dW3, db3 = linear_loss_step(Xd, Wd, bd, yd, nd, loss=kind, grad_out=gout.to(dev))
    finally:
        lib.ltr_debug_cluster_mode(0)
    _C.device_status()
    assert torch.equal(loss, loss3) and torch.equal(dW, dW3) and torch.equal(db, db3)
    if kind in ("hinge", "dcg_hinge"):
        # Round 5: integer grades 0 .. 4 -> the hinge kinds by RANKS (two binary searches per document in the list
        # sorted by score, with the pair pass's own fp32 margin predicate) instead of the pair pass.  The per-document
        # gradients are the same integers: for the plain hinge dW / db are bit-identical to the pair pass's, the
        # DCG modifier scales the summed shares instead of every document (fp32 rounding), the pair sum comes from
        # #active + sum (s - c) g.
        lib.ltr_debug_cluster_mode(2)
        try:
            loss4, dW4, db4 = linear_loss_step(Xd, Wd, bd, yd, nd, loss=kind, grad_out=gout.to(dev))
        finally:
            lib.ltr_debug_cluster_mode(0)
        _C.device_status()
        assert np.allclose(loss4.cpu().numpy(), want_l, rtol=rtol, atol=1e-5)
        assert torch.allclose(loss, loss4, rtol=2e-6, atol=1e-6)
        if kind == "hinge":
            assert torch.equal(dW, dW4) and torch.equal(db, db4)
        else:
            assert torch.allclose(dW, dW4, rtol=1e-5, atol=1e-6 * float(dW4.abs().max()))


def test_out_of_range_list_lengths_under_the_scheduling():
    """n[b] < 0 or > list_len is clamped exactly as on the unscheduled paths (the scheduling pass
    orders by the clamped value and hands it to the kernel)."""
    from pytorchltr_amd.fused import linear_loss_step
    from pytorchltr_amd._autograd import pairwise_loss_and_grad
    from pytorchltr_amd import _C
    dev = _dev()
    B, L, F = 600, 128, 136
    s, y, n, X, W, b = synth(B, L, 99, F=F)
    wild = n.clone()
    wild[::7] = -3
    wild[3::11] = L + 50
    tame = wild.clamp(0, L)
    Xd, Wd, bd, yd = X.to(dev), W.to(dev), b.to(dev), y.to(dev)
    a = linear_loss_step(Xd, Wd, bd, yd, wild.to(dev), loss="logistic")
    c = linear_loss_step(Xd, Wd, bd, yd, tame.to(dev), loss="logistic")
    assert all(torch.equal(u, v) for u, v in zip(a, c))
    B2 = 1100
    s2, y2, n2 = synth(B2, L, 98)[:3]
    wild2 = n2.clone()
    wild2[::5] = -1
    wild2[2::9] = 10 * L
    l1, g1 = pairwise_loss_and_grad(s2.to(dev), y2.to(dev), wild2.to(dev), _C.NDCG2)
    l2, g2 = pairwise_loss_and_grad(s2.to(dev), y2.to(dev), wild2.clamp(0, L).to(dev), _C.NDCG2)
    assert torch.equal(l1, l2) and torch.equal(g1, g2)


def _sorted_runs_cases(count, seed):
    import random
    rnd = random.Random(seed)
    out = []
    for _ in range(count):
        F = rnd.choice([64, 136, 220])
        L = rnd.choice([257, 300, 512, 600, 777, 1000, 1024])
        B = rnd.choice([1, 3, 8, 9, 31, 33, 48])
        out.append((B, L, F, rnd.choice(["hinge", "dcg_hinge"]), rnd.choice(["ties", "margin", "mixed_labels", "int32", "plain"]),
                    rnd.randrange(1 << 20)))
    return out


@pytest.mark.parametrize("case", _sorted_runs_cases(30, 20260930), ids=lambda c: "%dx%dx%d-%s-%s-%d" % c)
def test_hinge_by_sorted_runs_against_the_pair_pass(case):
    """Round 5: seeded stress of the sorted-runs path of the cluster kernel's 8-sweep members against its own pair pass
    (`ltr_debug_cluster_mode(2)`) and the oracle -- scores with many exact ties (features and weights on a coarse grid),
    score differences EXACTLY at the margin and one ulp to either side of it, batches in which some queries carry a
    grade outside 0..4 or a negative one (those queries take the pair pass, the others the runs: every member of a query
    decides alike), int32 labels.  The per-document gradients are integers either way: for the plain hinge dW and db
    must be bit-identical between the two paths; losses to fp32 rounding."""
    from pytorchltr_amd import _C
    from pytorchltr_amd.fused import linear_loss_step
    B, L, F, kind, flavour, seed = case
    lib = _C.lib()
    if lib.ltr_linear_fused_plan(getattr(_C, kind.upper()), B, L, F) != _C.PLAN_CLUSTER:
        pytest.skip("not a cluster shape")
    dev = _dev()
    g = torch.Generator().manual_seed(seed)
    s_, y, n, X, W, b = synth(B, L, seed, F=F)
    if flavour == "ties":
        X = torch.round(X * 2) / 2
        W = torch.round(W * 8 * F ** 0.5) / 8                    # sums of a few multiples of 1/16: plenty of equal scores
        b = torch.zeros(1)
    elif flavour == "margin":
        # column 0 is the score (W = e_0, bias 0): exact differences of 1 and 1 +- one ulp between neighbours
        W = torch.zeros(F); W[0] = 1.0; b = torch.zeros(1)
        base = torch.randn(B, L, generator=g).float()
        base[:, 1::3] = base[:, 0::3][:, :base[:, 1::3].shape[1]] - 1.0
        up = torch.nextafter(base[:, 0::3] - 1.0, torch.full_like(base[:, 0::3], 10.0))
        base[:, 2::3] = up[:, :base[:, 2::3].shape[1]]
        X = X.clone(); X[:, :, 0] = base
    elif flavour == "mixed_labels":
        y = y.clone()
        y[::3, 5] = 7                                              # a grade above 4 in every third query
        if B > 1:
            y[1::3, 2] = -1                                        # a negative one
    n[0] = L
    if B > 2:
        n[1], n[2] = 1, 0
    ydev = y.to(dev).to(torch.int32) if flavour == "int32" else y.to(dev)
    Xd, Wd, bd, nd = X.to(dev), W.to(dev), b.to(dev), n.to(dev)
    loss, dW, db = linear_loss_step(Xd, Wd, bd, ydev, nd, loss=kind)
    lib.ltr_debug_cluster_mode(2)
    try:
        loss_p, dW_p, db_p = linear_loss_step(Xd, Wd, bd, ydev, nd, loss=kind)
    finally:
        lib.ltr_debug_cluster_mode(0)
    _C.device_status()
    assert torch.isfinite(loss).all()
    assert torch.allclose(loss, loss_p, rtol=3e-6, atol=2e-5)
    if kind == "hinge":
        assert torch.equal(dW, dW_p) and torch.equal(db, db_p)
    else:
        assert torch.allclose(dW, dW_p, rtol=2e-5, atol=2e-6 * float(dW_p.abs().max() + 1e-30))
    if flavour not in ("margin", "ties"):        # (at exact ties / margins the fp64 oracle's scores decide some pairs the other way)
        want_l, _, want_dW, want_db = O.linear_pairwise(kind, X.numpy(), W.numpy(), float(b[0]), y.numpy(), n.numpy(), np.full(B, 1.0 / B))
        assert np.allclose(loss.cpu().numpy(), want_l, rtol=5e-4, atol=1e-5)
        tol = 4e-4 * max(1.0, float(np.max(np.abs(want_dW))))
        assert np.max(np.abs(dW.cpu().numpy() - want_dW)) < tol
