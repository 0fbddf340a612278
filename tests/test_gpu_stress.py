"""GPU tier: limits and odd shapes -- maximum list length, huge batches, degenerate feature
widths, extreme n -- against the oracle (small B) or through invariants (huge B)."""
import numpy as np
import pytest
import torch

from oracle import ltr_oracle as O
from tests.conftest import synth
from tests.test_gpu_parity import _check_grad, _check_loss, _run_direct

pytestmark = pytest.mark.gpu
KINDS = list(O.KINDS)


@pytest.mark.parametrize("kind", KINDS)
def test_maximum_list_length(kind):
    """list_len = ltr_max_list_len() = 4096: one workgroup's LDS, both-ends pass."""
    from pytorchltr_amd import _C
    L = _C.lib().ltr_max_list_len()
    s, y, n = synth(2, L, 23)
    n[0] = L
    n[1] = L // 3
    loss, ds = _run_direct(kind, s.numpy(), y.numpy(), n.numpy())
    want_l, want_g = O.pairwise_loss(kind, s.numpy(), y.numpy(), n.numpy())
    rtol = 2e-3 if kind in ("ndcg1", "ndcg2") else 5e-4          # ~1.7e7 fp32 terms per query
    assert np.allclose(loss, want_l, rtol=rtol, atol=1e-5)
    scale = np.max(np.abs(want_g), axis=1, keepdims=True)
    assert np.all(np.abs(ds - want_g) <= 2e-4 * scale + 1e-5)
    if kind == "hinge":
        assert np.array_equal(ds, want_g)


def test_list_too_long_is_an_error():
    import pytorchltr_amd.loss as losses
    dev = torch.device("cuda:0")
    with pytest.raises(ValueError, match="exceeds"):
        losses.PairwiseHingeLoss()(torch.zeros(1, 5000, device=dev),
                                   torch.zeros(1, 5000, dtype=torch.long, device=dev),
                                   torch.tensor([5000], device=dev))


@pytest.mark.parametrize("kind", ["hinge", "ndcg2"])
def test_huge_batch_of_short_lists(kind):
    """B = 200k queries of 8 documents: grid-size and indexing limits; spot rows vs oracle."""
    B, L = 200_000, 8
    s, y, n = synth(B, L, 31)
    loss, ds = _run_direct(kind, s.numpy(), y.numpy(), n.numpy())
    rows = np.r_[0:50, B - 50:B, B // 2:B // 2 + 50]
    want_l, want_g = O.pairwise_loss(kind, s.numpy()[rows], y.numpy()[rows], n.numpy()[rows])
    _check_loss(loss[rows], want_l, L, kind)
    _check_grad(ds[rows], want_g, kind, exact=(kind == "hinge"))
    assert np.all(np.isfinite(loss)) and np.all(np.isfinite(ds))


def test_metrics_on_huge_batch_and_long_lists():
    import pytorchltr_amd.evaluation as ev
    dev = torch.device("cuda:0")
    s, y, n = synth(50_000, 20, 5)
    y = y * (torch.arange(20)[None, :] < n[:, None])
    got = ev.ndcg(s.to(dev), y.to(dev), n.to(dev), k=10).cpu().numpy()
    rows = np.r_[0:40, 49_960:50_000]
    assert np.allclose(got[rows], O.ndcg(s.numpy()[rows], y.numpy()[rows], n.numpy()[rows], k=10),
                       rtol=2e-6, atol=1e-6)
    s, y, n = synth(3, 4096, 6)
    y = y * (torch.arange(4096)[None, :] < n[:, None])
    for fn, ofn in ((ev.dcg, O.dcg), (ev.ndcg, O.ndcg)):
        curve = fn(s.to(dev), y.to(dev), n.to(dev)).cpu().numpy()
        assert np.allclose(curve, ofn(s.numpy(), y.numpy(), n.numpy()), rtol=5e-5, atol=1e-5)
    assert np.allclose(ev.arp(s.to(dev), y.to(dev), n.to(dev)).cpu().numpy(),
                       O.arp(s.numpy(), y.numpy(), n.numpy()), rtol=2e-5)


@pytest.mark.parametrize("F", [1, 2, 3, 7, 4, 8, 500, 1023, 1024, 2048, 2052, 4100])
def test_fused_step_feature_widths(F):
    """Scalar path (F % 4 != 0), vector path, one column, very wide rows."""
    from pytorchltr_amd.fused import linear_loss_step
    dev = torch.device("cuda:0")
    B, L = 5, 19
    s, y, n, X, W, b = synth(B, L, 40 + F, F=F)
    for kind in ("hinge", "arp1"):
        loss, dW, db = linear_loss_step(X.to(dev), W.to(dev), b.to(dev), y.to(dev), n.to(dev), loss=kind)
        want_l, _, want_dW, want_db = O.linear_pairwise(kind, X.numpy(), W.numpy(), float(b[0]), y.numpy(),
                                                        n.numpy(), np.full(B, 1.0 / B))
        assert np.allclose(loss.cpu().numpy(), want_l, rtol=5e-5, atol=1e-5), (kind, F)
        tol = 5e-5 * max(1.0, float(np.max(np.abs(want_dW))))
        assert np.max(np.abs(dW.cpu().numpy() - want_dW)) < tol, (kind, F)
        assert abs(float(db.cpu()[0]) - want_db) < tol


def test_fused_step_too_wide_is_an_error():
    """The weight vector must fit one workgroup's LDS next to the query block."""
    from pytorchltr_amd.fused import linear_loss_step
    dev = torch.device("cuda:0")
    F = 60000
    X = torch.zeros(1, 4, F, device=dev)
    with pytest.raises(RuntimeError, match="shape"):
        linear_loss_step(X, torch.zeros(F, device=dev), torch.zeros(1, device=dev),
                         torch.zeros(1, 4, dtype=torch.long, device=dev), torch.tensor([4], device=dev))


def test_extreme_n_values_and_nonfinite_padding():
    """n far outside [0, L], int64 extremes; inf/NaN garbage in padded slots must not leak."""
    s, y, n = synth(6, 40, 77)
    n = torch.tensor([-(2 ** 62), -1, 0, 40, 41, 2 ** 62])
    s2 = s.clone()
    s2[3:, :] = s[3:, :]
    for kind in KINDS:
        loss, ds = _run_direct(kind, s2.numpy(), y.numpy(), n.numpy())
        want_l, want_g = O.pairwise_loss(kind, s2.numpy(), y.numpy(), n.numpy())
        _check_loss(loss, want_l, 40, kind)
        _check_grad(ds, want_g, kind, exact=(kind == "hinge"))
    s3, y3, n3 = synth(4, 40, 78)
    pad = torch.arange(40)[None, :] >= n3[:, None]
    s4 = s3.clone()
    s4[pad] = float("nan")
    s5 = s3.clone()
    s5[pad] = float("inf")
    for kind in KINDS:
        base = _run_direct(kind, s3.numpy(), y3.numpy(), n3.numpy())
        for other in (s4, s5):
            got = _run_direct(kind, other.numpy(), y3.numpy(), n3.numpy())
            assert np.array_equal(base[0], got[0]) and np.array_equal(base[1], got[1]), kind


@pytest.mark.parametrize("L", [129, 200, 256, 300, 512, 1000, 1024, 1500, 2048, 2049, 4096])
def test_sorted_ranks_with_ties_on_long_lists(L):
    """Lists longer than 128 are ranked by a bitonic sort of (score, index) keys.  The tie rule
    must be the counting rank's: score descending, then index ascending, padded tail in index
    order -- bit-exact against the oracle on rows that are FULL of ties (few distinct scores,
    all-equal rows, +0.0 / -0.0, -inf scores), and the metrics built on the ranks agree."""
    from pytorchltr_amd.evaluation import arp, ndcg
    from pytorchltr_amd.utils import rank_by_score
    dev = torch.device("cuda")
    B = 7
    g = torch.Generator().manual_seed(L)
    scores = torch.randint(-3, 4, (B, L), generator=g).float() * 0.5       # 7 distinct values
    scores[1] = 1.25                                                        # one value only
    scores[2, ::2] = 0.0
    scores[2, 1::2] = -0.0                                                  # signed zeros tie
    scores[3] = torch.randn(L, generator=g)
    scores[3, ::5] = float("-inf")
    y = torch.randint(0, 5, (B, L), generator=g)
    n = torch.tensor([L, L, L - 1, L // 2 + 1, 1, 0, L + 3])
    got = rank_by_score(scores.to(dev), n.to(dev)).cpu().numpy()
    want = O.rank_by_score(scores.numpy(), n.numpy())
    assert np.array_equal(got, want)
    y0 = y * (torch.arange(L)[None, :] < n[:, None])
    for fn, ofn in ((lambda: ndcg(scores.to(dev), y0.to(dev), n.to(dev), k=10), lambda: O.ndcg(scores.numpy(), y0.numpy(), n.numpy(), k=10)),
                    (lambda: arp(scores.to(dev), y0.to(dev), n.to(dev)), lambda: O.arp(scores.numpy(), y0.numpy(), n.numpy()))):
        assert np.allclose(fn().cpu().numpy(), ofn(), rtol=2e-5, atol=1e-6)


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("shape", [(24, 1000), (256, 1000), (40, 300), (5, 1024), (64, 700)])
def test_split_query_launch_matches_single_workgroup(kind, shape):
    """ltr_pairwise_loss_ws_f32 (several workgroups share a long query, parts added in a fixed
    order by a finish kernel) against the one-workgroup-per-query kernel: same loss to fp32
    summation-order round-off, same gradient; hinge gradients (integers) bit-exact; padded slots
    exactly zero; and against the oracle on a few rows."""
    from pytorchltr_amd import _C
    from pytorchltr_amd._autograd import pairwise_loss_and_grad
    dev = torch.device("cuda")
    B, L = shape
    scores, y, n = synth(B, L, 17 + L)
    n[:4] = torch.tensor([L, 1, 0, 65])[:min(4, B)]
    kid = getattr(_C, kind.upper())
    a_l, a_g = pairwise_loss_and_grad(scores.to(dev), y.to(dev), n.to(dev), kid)
    b_l, b_g = pairwise_loss_and_grad(scores.to(dev), y.to(dev), n.to(dev), kid, cfg="split")
    assert torch.allclose(a_l, b_l, rtol=2e-5, atol=1e-5)
    scale = a_g.abs().amax(dim=1, keepdim=True).clamp_min(1e-30)
    assert bool(((a_g - b_g).abs() <= 2e-5 * scale + 1e-6).all())
    if kind == "hinge":
        assert torch.equal(a_g, b_g)
    pad = torch.arange(L)[None, :] >= n.clamp(max=L)[:, None]
    assert not b_g.cpu()[pad].any()
    rows = slice(0, min(B, 6))
    want_l, want_g = O.pairwise_loss(kind, scores[rows].numpy(), y[rows].numpy(), n[rows].numpy())
    _check_loss(b_l[rows].cpu().numpy(), want_l, L, kind)
    _check_grad(b_g[rows].cpu().numpy(), want_g, kind)


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("shape", [(3, 1025), (40, 1100), (300, 1300), (20, 2048), (600, 1251)])
def test_loss_kernels_between_1024_and_2048_documents(kind, shape):
    """Round 4: the symmetric pass (eight waves) and the split launch reach 2048 documents in the loss-only kernels --
    an untruncated MSLR-WEB30K batch has lists of up to 1251; above 1024 the both-ends pass was 4-7x slower.  Plain and
    workspace entry points against each other and against the oracle on a few rows, empty / one-document / full lists,
    padded slots exactly zero."""
    from pytorchltr_amd import _C
    from pytorchltr_amd._autograd import pairwise_loss_and_grad
    dev = torch.device("cuda")
    B, L = shape
    scores, y, n = synth(B, L, 5 + L)
    n[:3] = torch.tensor([L, 1, 0])[:min(3, B)]
    kid = getattr(_C, kind.upper())
    a_l, a_g = pairwise_loss_and_grad(scores.to(dev), y.to(dev), n.to(dev), kid)
    b_l, b_g = pairwise_loss_and_grad(scores.to(dev), y.to(dev), n.to(dev), kid, cfg="split")
    assert torch.allclose(a_l, b_l, rtol=5e-5, atol=1e-5)
    scale = a_g.abs().amax(dim=1, keepdim=True).clamp_min(1e-30)
    assert bool(((a_g - b_g).abs() <= 5e-5 * scale + 1e-6).all())
    pad = torch.arange(L)[None, :] >= n.clamp(max=L)[:, None]
    assert not a_g.cpu()[pad].any() and not b_g.cpu()[pad].any()
    rows = slice(0, min(B, 5))
    want_l, want_g = O.pairwise_loss(kind, scores[rows].numpy(), y[rows].numpy(), n[rows].numpy())
    for got_l, got_g in ((a_l, a_g), (b_l, b_g)):
        _check_loss(got_l[rows].cpu().numpy(), want_l, L, kind)
        _check_grad(got_g[rows].cpu().numpy(), want_g, kind)
    _C.device_status()


def test_split_query_launch_through_the_module_and_forward_only():
    """The loss modules take the split launch by themselves for long lists on small batches:
    forward-only (no gradient buffer) and forward+backward agree with the direct kernel."""
    from pytorchltr_amd import _C
    from pytorchltr_amd._autograd import pairwise_loss_and_grad
    from pytorchltr_amd.loss import PairwiseLogisticLoss
    dev = torch.device("cuda")
    B, L = 16, 900
    assert _C.lib().ltr_pairwise_loss_workspace_bytes(_C.LOGISTIC, B, L) > 0
    scores, y, n = synth(B, L, 3)
    ref_l, ref_g = pairwise_loss_and_grad(scores.to(dev), y.to(dev), n.to(dev), _C.LOGISTIC)
    loss_fn = PairwiseLogisticLoss()
    with torch.no_grad():
        fwd = loss_fn(scores.to(dev), y.to(dev), n.to(dev))
    assert torch.allclose(fwd, ref_l, rtol=2e-5, atol=1e-5)
    s = scores.to(dev).requires_grad_(True)
    out = loss_fn(s, y.to(dev), n.to(dev))
    out.sum().backward()
    assert torch.allclose(out.detach(), ref_l, rtol=2e-5, atol=1e-5)
    scale = ref_g.abs().amax(dim=1, keepdim=True).clamp_min(1e-30)
    assert bool(((s.grad - ref_g).abs() <= 2e-5 * scale + 1e-6).all())


@pytest.mark.parametrize("seed", range(12))
def test_random_shapes_in_the_scheduled_and_split_ranges(seed):
    """Random (B, L, kind, list-length pattern) in the ranges where block ids are mapped to queries
    by the in-kernel list-length order or a query is split over workgroups; both entry points
    against the oracle (NaN prefill: every loss / gradient row written exactly by its query)."""
    from pytorchltr_amd import _C
    rng = np.random.default_rng(1000 + seed)
    dev = torch.device("cuda")
    B = int(rng.integers(270, 1100)) if seed % 3 else int(rng.integers(20, 400))
    L = int(rng.integers(65, 330)) if seed % 3 else int(rng.integers(257, 900))
    kind = KINDS[int(rng.integers(0, len(KINDS)))]
    scores, y, n = synth(B, L, 5000 + seed)
    pattern = int(rng.integers(0, 4))
    if pattern == 1:
        n = torch.full_like(n, L)
    elif pattern == 2:
        n = torch.where(torch.rand(B, generator=torch.Generator().manual_seed(seed)) < 0.7,
                        torch.full_like(n, L), n)
    elif pattern == 3:
        n = torch.clamp(n // 8, min=0)
    lib = _C.lib()
    kid = getattr(_C, kind.upper())
    sd, yd, nd = scores.to(dev), y.to(dev), n.to(dev)
    want_l, want_g = O.pairwise_loss(kind, scores.numpy(), y.numpy(), n.numpy())
    for entry in ("plain", "workspace"):
        loss = torch.full((B,), float("nan"), device=dev)
        ds = torch.full((B, L), float("nan"), device=dev)
        if entry == "plain":
            rc = lib.ltr_pairwise_loss_f32(kid, 1.0, sd.data_ptr(), yd.data_ptr(), _C.label_dtype(yd),
                                           nd.data_ptr(), B, L, loss.data_ptr(), ds.data_ptr(), _C.stream_of(sd))
        else:
            wsb = lib.ltr_pairwise_loss_workspace_bytes(kid, B, L)
            ws = torch.empty(max(wsb, 4) // 4, device=dev)
            rc = lib.ltr_pairwise_loss_ws_f32(kid, 1.0, sd.data_ptr(), yd.data_ptr(), _C.label_dtype(yd),
                                              nd.data_ptr(), B, L, loss.data_ptr(), ds.data_ptr(),
                                              ws.data_ptr(), wsb, _C.stream_of(sd))
        _C.check(rc)
        torch.cuda.synchronize()
        what = "%s B=%d L=%d %s pattern %d" % (entry, B, L, kind, pattern)
        assert not torch.isnan(loss).any() and not torch.isnan(ds).any(), what
        _check_loss(loss.cpu().numpy(), want_l, L, what)
        _check_grad(ds.cpu().numpy(), want_g, what)


def test_cluster_kernel_soak():
    """Many launches of the cluster kernel over random eligible shapes: its spin-waits must always
    complete (a stall is bounded and shows up as a NaN loss, never as a hung GPU)."""
    import random
    from pytorchltr_amd import _C
    lib = _C.lib()
    dev = torch.device("cuda")
    rnd = random.Random(7)
    launches = 0
    for it in range(120):
        B = rnd.choice([3, 17, 64, 100, 192, 256, 384])
        L = rnd.choice([300, 400, 512, 700, 1000, 1024])
        F = rnd.choice([64, 136, 220, 700])
        kid = rnd.randrange(7)
        if lib.ltr_linear_fused_plan(kid, B, L, F) != _C.PLAN_CLUSTER or B * L * F > 60e6:
            continue
        g = torch.Generator().manual_seed(it)
        X = torch.randn(B, L, F, generator=g).to(dev)
        y = torch.randint(0, 5, (B, L), generator=g).to(dev)
        n = torch.randint(0, L + 1, (B,), generator=g).to(dev)
        W = (torch.randn(F, generator=g) * 0.1).to(dev)
        bias = torch.zeros(1, device=dev)
        loss = torch.full((B,), float("nan"), device=dev)
        part = torch.empty(lib.ltr_linear_workspace_bytes(B, L, F) // 4, device=dev)
        for _ in range(4):
            _C.check(lib.ltr_linear_partials_f32(kid, 1.0, X.data_ptr(), W.data_ptr(), bias.data_ptr(), y.data_ptr(),
                                                 _C.label_dtype(y), n.data_ptr(), B, L, F, loss.data_ptr(), None,
                                                 part.data_ptr(), _C.stream_of(X)))
            launches += 1
        torch.cuda.synchronize()
        assert not torch.isnan(loss).any(), (B, L, F, kid)
        assert not torch.isnan(part[:B * (F + 1)]).any(), (B, L, F, kid)
    assert launches >= 100
