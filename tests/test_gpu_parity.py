"""GPU parity tests (run with `-m gpu` on an MI355X): the HIP kernels, called through the C ABI,
against the CPU oracle (fp64) and the golden vectors captured from the real reference.

Stated tolerances (measured spread of the reference's own fp32 vs fp64 runs):
  loss      rtol 1e-5 for list_len <= 256, 5e-4 above, atol 2e-6
  gradient  |err| <= 1e-5 * max|grad_row| + 1e-6; hinge gradients bit-exact (integers)
  rankings  bit-exact on the first n[b] entries of tie-free rows; tail = permutation of n..L-1
  metrics   rtol 2e-6 (list_len <= 256) / 2e-5 above, atol 1e-6
"""
import numpy as np
import pytest
import torch

from oracle import ltr_oracle as O
from tests.conftest import load_golden, synth

pytestmark = pytest.mark.gpu

G = load_golden()
KINDS = list(O.KINDS)
RANK_FREE = {"hinge", "dcg_hinge", "logistic", "arp1", "arp2"}


def _dev():
    assert torch.cuda.is_available(), "GPU tier needs a ROCm device"
    return torch.device("cuda:0")


def _loss_tol(L):
    return (5e-4 if L > 256 else 1e-5), 2e-6


def _check_loss(got, want, L, what):
    rtol, atol = _loss_tol(L)
    got = np.asarray(got, dtype=np.float64)
    assert np.all(np.isfinite(got)), what
    err = np.abs(got - want) - (atol + rtol * np.abs(want))
    assert np.all(err <= 0), "%s: loss mismatch, worst excess %.3e" % (what, err.max())


def _check_grad(got, want, what, exact=False):
    got = np.asarray(got, dtype=np.float64).reshape(want.shape)
    if exact:
        assert np.array_equal(got, want), what
        return
    scale = np.max(np.abs(want), axis=1, keepdims=True)
    err = np.abs(got - want) - (1e-5 * scale + 1e-6)
    assert np.all(err <= 0), "%s: grad mismatch, worst excess %.3e" % (what, err.max())


def _run_direct(kind, s, y, n, sigma=1.0, cfg=None):
    from pytorchltr_amd import _C
    from pytorchltr_amd._autograd import pairwise_loss_and_grad
    dev = _dev()
    loss, ds = pairwise_loss_and_grad(torch.as_tensor(s).to(dev), torch.as_tensor(y).to(dev),
                                      torch.as_tensor(n).to(dev), _C.__dict__[kind.upper()],
                                      sigma, cfg=cfg)
    return loss.cpu().numpy(), ds.cpu().numpy()


@pytest.mark.parametrize("name", [c["name"] for c in G.by_op("loss")])
def test_losses_vs_oracle_and_reference_vectors(name):
    case = G.cases[name]
    s, y, n = G.inputs(name)
    L = s.shape[1]
    for kind in case["kinds"]:
        loss, ds = _run_direct(kind, s, y, n, case["sigma"])
        want_l, want_g = O.pairwise_loss(kind, s, y, n, sigma=case["sigma"])
        _check_loss(loss, want_l, L, "%s/%s vs oracle" % (name, kind))
        _check_grad(ds, want_g, "%s/%s vs oracle" % (name, kind), exact=(kind == "hinge"))
        # and directly against what the real reference returned (fp32 run)
        _check_loss(loss, G.get(name, kind + "/loss32").astype(np.float64), L,
                    "%s/%s vs reference" % (name, kind))
        if case["tie_free"] or kind in RANK_FREE:
            ref_g = G.get(name, kind + "/grad32").astype(np.float64).reshape(want_g.shape)
            _check_grad(ds, ref_g, "%s/%s vs reference" % (name, kind), exact=(kind == "hinge"))


@pytest.mark.parametrize("name", [c["name"] for c in G.by_op("loss") if "literal" in c])
def test_reference_unit_test_literals(name):
    case = G.cases[name]
    s, y, n = G.inputs(name)
    for kind, expected in case["literal"].items():
        loss, _ = _run_direct(kind, s, y, n, case["sigma"])
        assert loss == pytest.approx(np.asarray(expected), rel=1e-5, abs=2e-6), kind


@pytest.mark.parametrize("kind", KINDS)
def test_module_autograd_path_mean_backward(kind):
    """The drop-in call of the reference's training loop:
    loss_fn(scores, relevance, n).mean().backward()  (examples/01-basic-usage.py:72-75)."""
    import pytorchltr_amd.loss as losses
    cls = {"hinge": losses.PairwiseHingeLoss, "dcg_hinge": losses.PairwiseDCGHingeLoss,
           "logistic": losses.PairwiseLogisticLoss, "arp1": losses.LambdaARPLoss1,
           "arp2": losses.LambdaARPLoss2, "ndcg1": losses.LambdaNDCGLoss1,
           "ndcg2": losses.LambdaNDCGLoss2}[kind]
    dev = _dev()
    s, y, n = synth(32, 48, 5)
    for shape3d in (False, True):
        sc = (s.reshape(32, 48, 1) if shape3d else s).clone().to(dev).requires_grad_(True)
        loss_fn = cls() if kind in ("hinge", "dcg_hinge") else cls(sigma=1.5)
        sigma = 1.0 if kind in ("hinge", "dcg_hinge") else 1.5
        out = loss_fn(sc, y.to(dev), n.to(dev))
        assert out.shape == (32,) and out.dtype == torch.float32
        out.mean().backward()
        assert sc.grad.shape == sc.shape
        want_l, want_g = O.pairwise_loss(kind, s.numpy(), y.numpy(), n.numpy(), sigma=sigma)
        _check_loss(out.detach().cpu().numpy(), want_l, 48, kind)
        _check_grad(sc.grad.cpu().numpy().reshape(32, 48) * 32.0, want_g, kind, exact=False)
    # no-grad forward takes the forward-only path and agrees
    with torch.no_grad():
        out2 = loss_fn(s.to(dev), y.to(dev), n.to(dev))
    assert torch.allclose(out2, out.detach(), rtol=0, atol=0)
    # weighted reduction: arbitrary upstream gradient per query
    sc = s.clone().to(dev).requires_grad_(True)
    w = torch.linspace(-1.0, 2.0, 32, device=dev)
    (loss_fn(sc, y.to(dev), n.to(dev)) * w).sum().backward()
    _check_grad(sc.grad.cpu().numpy(), want_g * w.cpu().numpy()[:, None].astype(np.float64), kind)


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("cfg", [(64, 1, 1), (64, 2, 1), (64, 4, 1), (128, 1, 2), (128, 2, 4),
                                 (256, 4, 4), (256, 1, 1), (512, 2, 2), (1024, 1, 1),
                                 # dpt == 0: symmetric pair pass, owners*msplit threads
                                 (64, 0, 1), (64, 0, 2), (128, 0, 2), (256, 0, 2), (256, 0, 4)])
def test_launch_shapes_agree(kind, cfg):
    """Every (owners, documents-per-thread, split) launch shape computes the same thing."""
    s, y, n = synth(12, 200, 9)
    n[0] = 200
    n[1] = 1
    n[2] = 0
    loss, ds = _run_direct(kind, s.numpy(), y.numpy(), n.numpy(), 1.0, cfg=cfg)
    want_l, want_g = O.pairwise_loss(kind, s.numpy(), y.numpy(), n.numpy())
    _check_loss(loss, want_l, 200, "%s %s" % (kind, cfg))
    _check_grad(ds, want_g, "%s %s" % (kind, cfg), exact=(kind == "hinge"))


@pytest.mark.parametrize("name", [c["name"] for c in G.by_op("metrics")])
def test_metrics_vs_oracle_and_reference_vectors(name):
    import pytorchltr_amd.evaluation as ev
    from pytorchltr_amd.utils import rank_by_score
    dev = _dev()
    case = G.cases[name]
    s, y, n = G.inputs(name)
    L = s.shape[1]
    rtol = 2e-5 if L > 256 else 2e-6
    ts, ty, tn = (torch.as_tensor(a).to(dev) for a in (s, y, n))
    ranking = rank_by_score(ts, tn).cpu().numpy()
    assert ranking.dtype == np.int64
    assert np.array_equal(ranking, O.rank_by_score(s, n))                 # bit-exact vs oracle
    nn = np.clip(n, 0, L)
    ref_rank = G.get(name, "ranking")
    for b in range(s.shape[0]):
        if case["tie_free"]:
            assert np.array_equal(ranking[b, :nn[b]], ref_rank[b, :nn[b]])
        assert sorted(ranking[b, nn[b]:].tolist()) == list(range(nn[b], L))
    got = ev.arp(ts, ty, tn).cpu().numpy()
    assert np.allclose(got, O.arp(s, y, n), rtol=rtol, atol=1e-6)
    assert np.allclose(got, G.get(name, "arp"), rtol=rtol, atol=1e-6)
    for k in case["ks"]:
        for exp in (True, False):
            tag = "k%s_%s" % ("all" if k is None else k, "exp" if exp else "lin")
            d = ev.dcg(ts, ty, tn, k=k, exp=exp).cpu().numpy()
            nd = ev.ndcg(ts, ty, tn, k=k, exp=exp).cpu().numpy()
            assert d.shape == G.get(name, "dcg_" + tag).shape
            assert np.allclose(d, O.dcg(s, y, n, k=k, exp=exp), rtol=rtol, atol=1e-6), tag
            assert np.allclose(nd, O.ndcg(s, y, n, k=k, exp=exp), rtol=rtol, atol=1e-6), tag
            assert np.allclose(d, G.get(name, "dcg_" + tag), rtol=rtol, atol=1e-6), tag
            assert np.allclose(nd, G.get(name, "ndcg_" + tag), rtol=rtol, atol=1e-6), tag


def test_helpers_vs_reference_vectors():
    from pytorchltr_amd.utils import batch_pairs, mask_padded_values, tiebreak_argsort
    dev = _dev()
    s = torch.as_tensor(G.get("helpers", "scores")).to(dev)
    y = torch.as_tensor(G.get("helpers", "relevance")).to(dev)
    n = torch.as_tensor(G.get("helpers", "n")).to(dev)
    assert np.array_equal(mask_padded_values(s, n).cpu().numpy(), G.get("helpers", "mask_default"))
    assert np.array_equal(mask_padded_values(s, n, mask_value=0.0).cpu().numpy(),
                          G.get("helpers", "mask_zero"))
    s2 = s.clone()
    assert mask_padded_values(s2, n, mask_value=0.0, mutate=True) is s2
    assert np.array_equal(s2.cpu().numpy(), G.get("helpers", "mask_zero"))
    assert np.array_equal(s.cpu().numpy(), G.get("helpers", "scores"))     # input untouched
    assert np.array_equal(batch_pairs(s).cpu().numpy(), G.get("helpers", "pairs_scores"))
    p = batch_pairs(y)
    assert p.dtype == torch.int64
    assert np.array_equal(p.cpu().numpy(), G.get("helpers", "pairs_relevance"))
    a = tiebreak_argsort(s).cpu().numpy()
    assert np.array_equal(a, np.argsort(-G.get("helpers", "scores"), axis=1, kind="stable"))
    a = tiebreak_argsort(s, descending=False).cpu().numpy()
    assert np.array_equal(a, np.argsort(G.get("helpers", "scores"), axis=1, kind="stable"))


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("shape", [(5, 300, (64, 0, 8)), (3, 1000, (64, 0, 16)), (3, 1000, (64, 0, 4)),
                                   (4, 700, (256, 4, 2)), (2, 1500, (256, 4, 4))])
def test_long_lists_both_pair_passes(kind, shape):
    """Lists beyond one wave tile: symmetric pass (dpt 0, up to 1024) and both-ends pass."""
    B, L, cfg = shape
    s, y, n = synth(B, L, 17)
    n[0] = L
    loss, ds = _run_direct(kind, s.numpy(), y.numpy(), n.numpy(), 1.0, cfg=cfg)
    want_l, want_g = O.pairwise_loss(kind, s.numpy(), y.numpy(), n.numpy())
    _check_loss(loss, want_l, L, "%s %s" % (kind, cfg))
    _check_grad(ds, want_g, "%s %s" % (kind, cfg), exact=(kind == "hinge"))


@pytest.mark.parametrize("kind", ["logistic", "arp1", "arp2", "ndcg1", "ndcg2"])
@pytest.mark.parametrize("L", [37, 128, 300])
def test_real_valued_labels_and_far_apart_scores(kind, L):
    """The log-sigmoid kinds on labels that are neither integers nor non-negative (LambdaNDCG2 reads the pair's
    orientation off the signed gain difference: the gains order documents as the labels do, negative labels
    included), with ties among the labels, and on scores so far apart that exp(-sigma d) is below 1e-5 for most
    pairs (log2(1 + e) without a series branch): loss and gradients against the fp64 oracle."""
    rng = np.random.default_rng(1234 + L)
    B = 6
    n = np.array([L, L - 1, max(1, L // 2), 2, 1, 0], dtype=np.int64)
    # (the row-weight kinds raise a sigmoid to the power of the label / gain: with far-apart scores a negative
    # exponent overflows in the reference's own formulation, fp64 included -- non-negative labels there)
    lo = 0.0 if kind in ("arp1", "ndcg1") else -1.5
    y = rng.uniform(lo, 3.0, size=(B, L)).astype(np.float32)
    y[:, ::5] = y[:, 1::5][:, :y[:, ::5].shape[1]]            # tied labels
    y[1] = np.round(y[1])                                       # one row of (partly negative) integers
    s = rng.normal(size=(B, L)).astype(np.float32)
    # (e = exp(-step k) for documents k places apart: every regime down to the smallest e; the largest
    # difference stays below what exp() takes in fp64, where the oracle follows the reference's formulation)
    # (LambdaARP1 raises the sigmoid to the label: a third of that range)
    span = 200.0 if kind == "arp1" else 600.0
    s[2] = (np.arange(L, dtype=np.float32) * np.float32(min(12.0, span / L)))[rng.permutation(L)]
    s[3] = s[3] * 30.0
    if kind == "arp1":
        s[3] = s[3] / 3.0
    loss, ds = _run_direct(kind, s, y, n)
    want_l, want_g = O.pairwise_loss(kind, s, y, n)
    _check_loss(loss, want_l, L, "%s L=%d" % (kind, L))
    _check_grad(ds, want_g, "%s L=%d" % (kind, L))


def test_ties_follow_the_documented_rule():
    """Ties: score descending, then index ascending -- identical to the oracle."""
    from pytorchltr_amd.utils import rank_by_score
    dev = _dev()
    g = torch.Generator().manual_seed(3)
    s = torch.randint(0, 4, (16, 50), generator=g).float()              # many ties
    y = torch.randint(0, 3, (16, 50), generator=g)
    n = torch.randint(0, 51, (16,), generator=g)
    r = rank_by_score(s.to(dev), n.to(dev)).cpu().numpy()
    assert np.array_equal(r, O.rank_by_score(s.numpy(), n.numpy()))
    for kind in KINDS:
        loss, ds = _run_direct(kind, s.numpy(), y.numpy(), n.numpy())
        want_l, want_g = O.pairwise_loss(kind, s.numpy(), y.numpy(), n.numpy())
        _check_loss(loss, want_l, 50, kind)
        _check_grad(ds, want_g, kind, exact=(kind == "hinge"))


# ---------------------------------------------------------------------------------------
# BASELINE.json full sizes: oracle on every row (list_len 128) / every 4th row and every nearly full
# list (long lists) + size-independent properties
# ---------------------------------------------------------------------------------------
FULL = [("C2", 1024, 128, ["hinge", "logistic", "ndcg2"]),
        ("C3", 1024, 128, ["ndcg2", "arp1", "ndcg1"]),
        ("C4", 256, 1000, ["dcg_hinge", "ndcg2"]),
        ("C5", 512, 512, ["hinge", "arp2"])]


@pytest.mark.parametrize("cfg", FULL, ids=[f[0] for f in FULL])
def test_full_size_configs(cfg):
    _, B, L, kinds = cfg
    dev = _dev()
    s, y, n = synth(B, L, 0)
    # oracle rows: EVERY row at list_len 128 (C2 / C3: the C oracle takes ~20 ms per kind); for the long
    # lists every 4th row plus every nearly full list (n >= 0.9 L: the longest pair passes, all tiles,
    # every split of the launch shapes)
    if L <= 128:
        rows = np.arange(B)
    else:
        rows = np.unique(np.r_[0:B:4, np.nonzero(n.numpy() >= 0.9 * L)[0], B - 2:B])
    for kind in kinds:
        loss, ds = _run_direct(kind, s.numpy(), y.numpy(), n.numpy())
        assert np.all(np.isfinite(loss)) and np.all(np.isfinite(ds))
        want_l, want_g = O.pairwise_loss(kind, s.numpy()[rows], y.numpy()[rows], n.numpy()[rows])
        _check_loss(loss[rows], want_l, L, kind)
        _check_grad(ds[rows], want_g, kind, exact=(kind == "hinge"))
        nn = n.numpy()
        # property: gradients vanish on padded documents, and sum to 0 over a query
        # (every loss is a function of score differences only)
        mask = np.arange(L)[None, :] >= nn[:, None]
        assert np.all(ds[mask] == 0.0)
        scale = np.abs(ds).sum(axis=1) + 1e-6
        assert np.all(np.abs(ds.sum(axis=1)) <= 2e-5 * scale)
        # property: the n-mask -- garbage in padded slots changes nothing (bit-exact)
        s2, y2 = s.clone(), y.clone()
        pad = torch.as_tensor(mask)
        s2[pad] = 1e6
        y2[pad] = 7
        loss2, ds2 = _run_direct(kind, s2.numpy(), y2.numpy(), nn)
        assert np.array_equal(loss, loss2) and np.array_equal(ds, ds2)
        # property: permuting the real documents of a query permutes the gradient and keeps the
        # loss (up to summation order)
        perm = torch.stack([torch.cat([torch.randperm(int(k)), torch.arange(int(k), L)]) for k in nn])
        s3 = torch.gather(s, 1, perm)
        y3 = torch.gather(y, 1, perm)
        loss3, ds3 = _run_direct(kind, s3.numpy(), y3.numpy(), nn)
        rtol, atol = _loss_tol(L)
        assert np.allclose(loss3, loss, rtol=10 * rtol, atol=10 * atol)
        back = np.take_along_axis(ds, perm.numpy(), axis=1)
        gs = np.max(np.abs(back), axis=1, keepdims=True)
        ok = np.abs(ds3 - back) <= 2e-4 * gs + 1e-5
        if kind in ("ndcg1", "ndcg2"):
            # rank-dependent losses: the index tie-break is not permutation invariant, so
            # only rows whose real scores are tie-free are comparable
            tie_free = np.array([len(np.unique(s.numpy()[b, :nn[b]])) == nn[b] for b in range(B)])
            assert tie_free.sum() > B // 2
            ok = ok[tie_free]
        assert np.all(ok)


def test_full_size_metrics_c3():
    import pytorchltr_amd.evaluation as ev
    from pytorchltr_amd.utils import rank_by_score
    dev = _dev()
    s, y, n = synth(1024, 128, 0)
    y = y * (torch.arange(128)[None, :] < n[:, None])
    ts, ty, tn = s.to(dev), y.to(dev), n.to(dev)
    r = rank_by_score(ts, tn).cpu().numpy()
    assert np.array_equal(r, O.rank_by_score(s.numpy(), n.numpy()))
    # sortedness property on every row
    srt = np.take_along_axis(s.numpy(), r, axis=1)
    for b in range(1024):
        k = int(n[b])
        assert np.all(np.diff(srt[b, :k]) <= 0)
    got = ev.ndcg(ts, ty, tn, k=10).cpu().numpy()
    assert np.allclose(got, O.ndcg(s.numpy(), y.numpy(), n.numpy(), k=10), rtol=2e-6, atol=1e-6)
    assert np.all((got >= 0) & (got <= 1 + 1e-6))
    # ndcg of the ideal ordering is 1 wherever a relevant document exists
    ideal = ev.ndcg(ty.float(), ty, tn, k=10).cpu().numpy()
    has_rel = (y.numpy().sum(axis=1) > 0)
    assert np.allclose(ideal[has_rel], 1.0, atol=1e-6)
    curve = ev.dcg(ts, ty, tn).cpu().numpy()
    # cumulative (a parallel prefix sum is monotone only up to an ulp of the running value)
    assert np.all(np.diff(curve, axis=1) >= -2e-6 * np.abs(curve[:, 1:]) - 1e-6)
    assert np.allclose(curve[:, 9], ev.dcg(ts, ty, tn, k=10).cpu().numpy(), rtol=2e-6, atol=1e-6)
    a = ev.arp(ts, ty, tn).cpu().numpy()
    assert np.allclose(a, O.arp(s.numpy(), y.numpy(), n.numpy()), rtol=2e-6, atol=1e-6)


def test_large_scores_stay_finite():
    """Outside the reference's finite domain (|sigma d| > 88 gives inf/NaN there) the kernels
    use a stable softplus: finite loss, gradients bounded by the pair weights."""
    s = np.array([[0.0, 500.0, -500.0, 90.0, -90.0]], dtype=np.float32)
    y = np.array([[2, 0, 1, 1, 0]], dtype=np.int64)
    n = np.array([5], dtype=np.int64)
    for kind in KINDS:
        loss, ds = _run_direct(kind, s, y, n)
        assert np.all(np.isfinite(loss)) and np.all(np.isfinite(ds)), kind
        want_l, want_g = O.pairwise_loss(kind, s, y, n)
        if np.all(np.isfinite(want_l)):
            _check_loss(loss, want_l, 5, kind)


def test_argument_errors_are_loud():
    import pytorchltr_amd.loss as losses
    dev = _dev()
    s, y, n = synth(4, 8, 1)
    with pytest.raises(RuntimeError):
        losses.PairwiseHingeLoss()(s, y, n)                          # CPU tensors: no fallback
    with pytest.raises(ValueError):
        losses.PairwiseHingeLoss()(s.to(dev), y[:, :4].to(dev), n.to(dev))
    with pytest.raises(ValueError):
        losses.PairwiseHingeLoss()(s.to(dev), y.to(dev), n[:2].to(dev))
    out = losses.PairwiseHingeLoss()(s.to(dev)[:0], y.to(dev)[:0], n.to(dev)[:0])
    assert out.shape == (0,)
    # int32 n and float labels are accepted, like the reference
    a = losses.LambdaNDCGLoss2()(s.to(dev), y.to(dev), n.to(dev))
    b = losses.LambdaNDCGLoss2()(s.to(dev), y.float().to(dev), n.int().to(dev))
    assert torch.equal(a, b)


# ---------------------------------------------------------------------------------------
# fp64: the reference computes in the dtype of `scores` (fp64 in -> fp64 out)
# ---------------------------------------------------------------------------------------
_LOSS_CLS = None


def _loss_cls(kind):
    import pytorchltr_amd.loss as losses
    return {"hinge": losses.PairwiseHingeLoss, "dcg_hinge": losses.PairwiseDCGHingeLoss,
            "logistic": losses.PairwiseLogisticLoss, "arp1": losses.LambdaARPLoss1,
            "arp2": losses.LambdaARPLoss2, "ndcg1": losses.LambdaNDCGLoss1,
            "ndcg2": losses.LambdaNDCGLoss2}[kind]


@pytest.mark.parametrize("kind", KINDS)
def test_fp64_scores_give_fp64_arithmetic(kind):
    dev = _dev()
    for (B, L, seed) in ((6, 40, 3), (3, 300, 4)):
        s, y, n = synth(B, L, seed)
        sc = s.double().to(dev).requires_grad_(True)
        out = _loss_cls(kind)()(sc, y.to(dev), n.to(dev))
        assert out.dtype == torch.float64
        out.sum().backward()
        assert sc.grad.dtype == torch.float64
        want_l, want_g = O.pairwise_loss(kind, s.double().numpy(), y.numpy(), n.numpy())
        assert np.allclose(out.detach().cpu().numpy(), want_l, rtol=1e-11, atol=1e-12), kind
        scale = np.max(np.abs(want_g), axis=1, keepdims=True) + 1e-300
        assert np.all(np.abs(sc.grad.cpu().numpy() - want_g) <= 1e-11 * scale + 1e-13), kind
        # and it agrees with the reference's own fp64 run where that was captured
    name = "syn_b8_l16"
    s, y, n = G.inputs(name)
    sc = torch.as_tensor(s).double().to(dev).requires_grad_(True)
    out = _loss_cls(kind)()(sc, torch.as_tensor(y).to(dev), torch.as_tensor(n).to(dev))
    out.sum().backward()
    tol = 1e-6 if kind in ("ndcg1", "ndcg2") else 1e-11      # the reference keeps fp32 tables there
    assert np.allclose(out.detach().cpu().numpy(), G.get(name, kind + "/loss64"), rtol=tol, atol=1e-12)
    assert np.allclose(sc.grad.cpu().numpy(), G.get(name, kind + "/grad64").reshape(8, 16), rtol=10 * tol, atol=1e-9)


@pytest.mark.parametrize("kind", ["logistic", "arp1", "arp2", "ndcg1", "ndcg2"])
def test_gradcheck_in_double(kind):
    """torch.autograd.gradcheck: analytic backward vs numerical Jacobian, in fp64."""
    dev = _dev()
    s, y, n = synth(3, 9, 12)
    sc = s.double().to(dev).requires_grad_(True)
    fn = _loss_cls(kind)(sigma=1.3)
    assert torch.autograd.gradcheck(lambda t: fn(t, y.to(dev), n.to(dev)), (sc,), eps=1e-6, atol=1e-7,
                                    rtol=1e-6, nondet_tol=0.0)


def test_hipgraph_capture_and_side_stream():
    """The C ABI only enqueues kernels on the caller's stream: a training-step slice can be
    captured in a hipGraph and replayed on new data; work issued on a side stream stays there."""
    from pytorchltr_amd.evaluation import ndcg
    dev = _dev()
    s, y, n = synth(64, 100, 8)
    sc = s.clone().to(dev).requires_grad_(True)
    yd, nd = y.to(dev), n.to(dev)
    loss_fn = _loss_cls("ndcg2")()
    static_loss = torch.zeros(64, device=dev)
    static_metric = torch.zeros(64, device=dev)

    def step():
        sc.grad = None
        out = loss_fn(sc, yd, nd)
        out.mean().backward()
        static_loss.copy_(out.detach())
        static_metric.copy_(ndcg(sc.detach(), yd, nd, k=10))

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()
    for seed in (9, 10):
        s2, _, _ = synth(64, 100, seed)
        with torch.no_grad():
            sc.copy_(s2.to(dev))                  # new data into the static input
        graph.replay()
        torch.cuda.synchronize()
        want_l, want_g = O.pairwise_loss("ndcg2", s2.numpy(), y.numpy(), n.numpy())
        _check_loss(static_loss.cpu().numpy(), want_l, 100, "graph replay")
        _check_grad(sc.grad.cpu().numpy() * 64.0, want_g, "graph replay")
        assert np.allclose(static_metric.cpu().numpy(), O.ndcg(s2.numpy(), y.numpy(), n.numpy(), k=10),
                           rtol=2e-6, atol=1e-6)


@pytest.mark.parametrize("shape", [(300, 256), (1100, 128), (1030, 300), (2048, 130)])
@pytest.mark.parametrize("lists", ["ragged", "full", "two_lengths", "mostly_empty"])
def test_list_length_scheduling_in_the_loss_kernel(shape, lists):
    """Shapes where the loss kernel maps block ids to queries through the in-kernel list-length
    order (more than one workgroup per CU, lists of 128+): every query still gets its own loss and
    gradient row (NaN prefill), equal to the oracle's."""
    from pytorchltr_amd import _C
    B, L = shape
    dev = _dev()
    s, y, n = synth(B, L, 4242)[:3]
    if lists == "full":
        n = torch.full_like(n, L)
    elif lists == "two_lengths":
        n = torch.where(torch.arange(B) % 5 == 0, torch.full_like(n, L), torch.full_like(n, 3))
    elif lists == "mostly_empty":
        n = torch.where(torch.arange(B) % 7 == 0, n, torch.zeros_like(n))
    sd, yd, nd = s.to(dev), y.to(dev), n.to(dev)
    lib = _C.lib()
    for kind in ("hinge", "logistic", "ndcg2"):
        loss = torch.full((B,), float("nan"), device=dev)
        ds = torch.full((B, L), float("nan"), device=dev)
        _C.check(lib.ltr_pairwise_loss_f32(_C.__dict__[kind.upper()], 1.0, sd.data_ptr(), yd.data_ptr(),
                                           _C.label_dtype(yd), nd.data_ptr(), B, L, loss.data_ptr(),
                                           ds.data_ptr(), _C.stream_of(sd)))
        torch.cuda.synchronize()
        assert not torch.isnan(loss).any() and not torch.isnan(ds).any(), kind
        want_l, want_g = O.pairwise_loss(kind, s.numpy(), y.numpy(), n.numpy(), sigma=1.0)
        _check_loss(loss.cpu().numpy(), want_l, L, "%s %s %s" % (shape, lists, kind))
        _check_grad(ds.cpu().numpy(), want_g, "%s %s %s" % (shape, lists, kind), exact=(kind == "hinge"))


@pytest.mark.parametrize("shape", [(4096, 128), (16384, 100), (33000, 128), (66000, 90), (4100, 64), (4096, 256), (4100, 300), (16400, 258), (1024, 520)])
def test_many_queries_take_narrow_workgroups(shape):
    """Round 6: once every CU has several rounds of queries the loss-only kernels give a query fewer waves (choose_loss_shape:
    1 / 2 waves on lists up to 128, 4 on 256, 8 / 4 on longer ones) and the metric kernels one wave (lists of 65 .. 128) --
    what bounds those launches is the number of queries a CU has in flight.  Every shape the rule can pick, against the oracle,
    every row written (NaN prefill), run to run identical."""
    from pytorchltr_amd import _C
    B, L = shape
    dev = _dev()
    s, y, n = synth(B, L, 77 + L)[:3]
    sd, yd, nd = s.to(dev), y.to(dev), n.to(dev)
    lib = _C.lib()
    for kind in ("hinge", "dcg_hinge", "logistic", "arp2", "ndcg1", "ndcg2"):
        outs = []
        for rep in range(2):
            loss = torch.full((B,), float("nan"), device=dev)
            ds = torch.full((B, L), float("nan"), device=dev)
            _C.check(lib.ltr_pairwise_loss_f32(_C.__dict__[kind.upper()], 1.0, sd.data_ptr(), yd.data_ptr(), _C.label_dtype(yd), nd.data_ptr(),
                                               B, L, loss.data_ptr(), ds.data_ptr(), _C.stream_of(sd)))
            torch.cuda.synchronize()
            outs.append((loss.cpu(), ds.cpu()))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), kind
        loss, ds = outs[0]
        assert not torch.isnan(loss).any() and not torch.isnan(ds).any(), kind
        want_l, want_g = O.pairwise_loss(kind, s.numpy(), y.numpy(), n.numpy(), sigma=1.0)
        _check_loss(loss.numpy(), want_l, L, "%s %s" % (shape, kind))
        if kind in ("hinge", "dcg_hinge"):
            # a pair whose score difference sits within an fp32 ulp of the margin (tens of millions of pairs here: expect about one
            # per kind) may fall on the other side in the fp64 oracle: its two gradient entries then differ by the pair's weight (<= 1)
            diff = np.abs(ds.numpy().astype(np.float64) - want_g)
            bad = diff > 1e-5 * np.max(np.abs(want_g), axis=1, keepdims=True) + 1e-6
            assert np.count_nonzero(bad) <= 8 and np.max(diff) <= 2.0, (kind, np.count_nonzero(bad), np.max(diff))
        else:
            _check_grad(ds.numpy(), want_g, "%s %s" % (shape, kind))
    # the metric kernels: one wave per query on lists of 65 .. 128, two keys per thread on the sort path (lists beyond 256)
    import pytorchltr_amd.evaluation as ev
    from pytorchltr_amd.utils import rank_by_score
    rtol = 2e-5 if L > 256 else 2e-6
    for k in (10, None):
        assert np.allclose(ev.ndcg(sd, yd, nd, k=k).cpu().numpy(), O.ndcg(s.numpy(), y.numpy(), n.numpy(), k=k), rtol=rtol, atol=1e-6), k
    assert np.allclose(ev.dcg(sd, yd, nd, k=5, exp=False).cpu().numpy(), O.dcg(s.numpy(), y.numpy(), n.numpy(), k=5, exp=False), rtol=rtol, atol=1e-6)
    assert np.allclose(ev.arp(sd, yd, nd).cpu().numpy(), O.arp(s.numpy(), y.numpy(), n.numpy()), rtol=rtol, atol=1e-6)
    assert np.array_equal(rank_by_score(sd, nd).cpu().numpy(), O.rank_by_score(s.numpy(), n.numpy()))
