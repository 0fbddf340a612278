"""fused ReLU-MLP scorer + loss + backward (ltr_mlp_pairwise_f32) against
the fp64 oracle (oracle_mlp_pairwise, itself checked against the real reference's autograd in
tests/test_oracle_golden.py::test_mlp_oracle_matches_reference_autograd) and against the unfused
composition of torch layers + the HIP loss kernel.

Tolerances (fp32 kernel vs fp64 oracle): loss rtol 2e-5 / atol 1e-5; every gradient tensor
<= 2e-5 * max(max|that tensor|, max|any gradient| / 4) + 1e-6 (sums over up to B*L documents of fp32
products; db3 cancels to zero exactly in exact arithmetic)."""
import numpy as np
import pytest
import torch

from tests.conftest import synth

pytestmark = pytest.mark.gpu

KINDS = ("hinge", "dcg_hinge", "logistic", "arp1", "arp2", "ndcg1", "ndcg2")


@pytest.fixture(params=["tile", "wide"], autouse=True)
def _mlp_layout(request):
    """Every test of this file runs on both kernel layouts of the training step: the 4-wave tile
    kernel (csrc/ltr_mlp2.inc, the default for F <= 144) and the 8-wave kernel (csrc/ltr_mlp.inc,
    which also serves wider feature rows and the forward-only scores)."""
    from pytorchltr_amd import _C
    _C.lib().ltr_debug_mlp_layout(1 if request.param == "wide" else 2)
    yield
    _C.lib().ltr_debug_mlp_layout(0)


def _mlp_params(F, H1, H2, seed):
    g = torch.Generator().manual_seed(seed)

    def u(*shape, fan):
        bound = 1.0 / np.sqrt(fan)
        return (torch.rand(*shape, generator=g) * 2 - 1) * bound
    return [u(H1, F, fan=F), u(H1, fan=F), u(H2, H1, fan=H1), u(H2, fan=H1), u(1, H2, fan=H2), u(1, fan=H2)]


def _case(B, L, F, H1, H2, seed, full=False):
    scores, y, n = synth(B, L, seed)
    g = torch.Generator().manual_seed(seed + 77)
    X = torch.randn(B, L, F, generator=g)
    if full:
        n = torch.full_like(n, L)
    return X, y, n, _mlp_params(F, H1, H2, seed + 5)


def _check(kind, X, y, n, params, grad_out=None, sigma=1.0):
    from oracle import ltr_oracle as O
    from pytorchltr_amd import fused
    from pytorchltr_amd import loss as Lmod
    B, L, F = X.shape
    dev = torch.device("cuda")
    go = np.full(B, 1.0 / B) if grad_out is None else grad_out.numpy().astype(np.float64)
    want_l, want_s, want_g = O.mlp_pairwise(kind, X.numpy(), [p.numpy() for p in params], y.numpy(),
                                            n.numpy(), go, sigma=sigma)
    loss_obj = kind
    if sigma != 1.0:
        cls = {"logistic": Lmod.PairwiseLogisticLoss, "arp1": Lmod.LambdaARPLoss1, "arp2": Lmod.LambdaARPLoss2,
               "ndcg1": Lmod.LambdaNDCGLoss1, "ndcg2": Lmod.LambdaNDCGLoss2}[kind]
        loss_obj = cls(sigma=sigma)
    lossv, grads, scores, lsum = fused.mlp_loss_step(
        X.to(dev), [p.to(dev) for p in params], y.to(dev), n.to(dev), loss=loss_obj,
        grad_out=None if grad_out is None else grad_out.to(dev), return_scores=True, return_loss_sum=True)
    torch.cuda.synchronize()
    got_l = lossv.cpu().numpy()
    assert np.allclose(got_l, want_l, rtol=2e-5, atol=1e-5), (kind, np.abs(got_l - want_l).max())
    assert np.allclose(float(lsum), want_l.sum(), rtol=2e-5, atol=1e-4)
    valid = (np.arange(L)[None, :] < np.clip(n.numpy(), 0, L)[:, None])
    got_s = scores.cpu().numpy()
    assert np.allclose(got_s[valid], want_s[valid], rtol=1e-5, atol=2e-6)
    assert not got_s[~valid].any()
    # cancelling sums (db3 = sum of ds is exactly 0 for a pairwise loss) are judged against the
    # scale of the whole gradient, not against their own near-zero value
    scale = max(np.abs(want_g[k]).max() for k in ("W1", "b1", "W2", "b2", "W3", "b3"))
    _, ds = O.pairwise_loss(kind, want_s, y.numpy(), n.numpy(), sigma=sigma)
    scale = max(scale, float((np.abs(ds) * go[:, None]).sum()))
    for key, got in zip(("W1", "b1", "W2", "b2", "W3", "b3"), grads):
        w = want_g[key]
        tol = 2e-5 * max(np.abs(w).max(), 0.25 * scale) + 1e-6
        err = np.abs(got.cpu().numpy().reshape(w.shape) - w).max()
        assert err <= tol, (kind, key, err, tol)


@pytest.mark.parametrize("kind", KINDS)
def test_small_network(kind):
    _check(kind, *_case(8, 16, 8, 5, 3, 1234))


@pytest.mark.parametrize("kind", KINDS)
def test_guide_network_c2_rows(kind):
    # the guide's 136-50-10-1 network on C2-shaped queries (L = 128), ragged n
    _check(kind, *_case(12, 128, 136, 50, 10, 7))


@pytest.mark.parametrize("kind", ("hinge", "ndcg2"))
def test_full_lists_and_weights(kind):
    X, y, n, params = _case(9, 128, 136, 50, 10, 11, full=True)
    g = torch.Generator().manual_seed(3)
    _check(kind, X, y, n, params, grad_out=torch.rand(9, generator=g) + 0.1)


@pytest.mark.parametrize("shape", [(5, 20, 44, 50, 10), (4, 100, 220, 64, 16), (3, 64, 64, 17, 1),
                                   (2, 128, 4, 1, 1), (3, 33, 80, 33, 9), (2, 17, 144, 50, 10)])
def test_shapes(shape):
    B, L, F, H1, H2 = shape
    for kind in ("hinge", "logistic", "ndcg1"):
        _check(kind, *_case(B, L, F, H1, H2, 21 + F))


def test_more_queries_than_workgroups():
    # persistent workgroups loop over queries: B > #CUs, edge rows n = 0, 1, L, > L
    B, L, F = 700, 24, 16
    X, y, n, params = _case(B, L, F, 12, 4, 99)
    n[:6] = torch.tensor([0, 1, L, L + 5, 2, 0])
    _check("hinge", X, y, n, params)
    _check("arp2", X, y, n, params)


def test_sigma():
    for kind in ("logistic", "ndcg2"):
        _check(kind, *_case(6, 40, 24, 10, 4, 5), sigma=2.0)


def test_padded_garbage_is_ignored():
    from pytorchltr_amd import fused
    dev = torch.device("cuda")
    X, y, n, params = _case(10, 64, 32, 20, 6, 8)
    P = [p.to(dev) for p in params]
    a = fused.mlp_loss_step(X.to(dev), P, y.to(dev), n.to(dev), loss="ndcg2")
    Xg = X.clone()
    yg = y.clone()
    for b in range(10):
        Xg[b, int(n[b]):] = float("nan")
        yg[b, int(n[b]):] = 4
    bb = fused.mlp_loss_step(Xg.to(dev), P, yg.to(dev), n.to(dev), loss="ndcg2")
    assert torch.equal(a[0], bb[0])
    for u, v in zip(a[1], bb[1]):
        assert torch.equal(u, v)                       # bit-exact


def test_deterministic():
    from pytorchltr_amd import fused
    dev = torch.device("cuda")
    X, y, n, params = _case(300, 128, 136, 50, 10, 2)
    args = (X.to(dev), [p.to(dev) for p in params], y.to(dev), n.to(dev))
    a = fused.mlp_loss_step(*args, loss="logistic")
    b = fused.mlp_loss_step(*args, loss="logistic")
    assert torch.equal(a[0], b[0])
    for u, v in zip(a[1], b[1]):
        assert torch.equal(u, v)


@pytest.mark.parametrize("reduction", ("mean", "sum"))
def test_module_matches_unfused_composition(reduction):
    """FusedMLPLoss == the guide's Model + loss module + .mean()/.sum() + backward."""
    from pytorchltr_amd.fused import FusedMLPLoss
    from pytorchltr_amd.loss import PairwiseLogisticLoss
    dev = torch.device("cuda")
    X, y, n, _ = _case(16, 20, 136, 50, 10, 42)
    torch.manual_seed(42)
    fusedm = FusedMLPLoss(136, PairwiseLogisticLoss(sigma=1.5), reduction=reduction).to(dev)
    plain = FusedMLPLoss(136, PairwiseLogisticLoss(sigma=1.5)).to(dev)
    plain.load_state_dict(fusedm.state_dict())
    out = fusedm(X.to(dev), y.to(dev), n.to(dev))
    (out * 3.0).backward()                                  # upstream scalar is honoured
    per_query = PairwiseLogisticLoss(sigma=1.5)(plain.score(X.to(dev)), y.to(dev), n.to(dev))
    ref = per_query.mean() if reduction == "mean" else per_query.sum()
    (ref * 3.0).backward()
    assert torch.allclose(out, ref, rtol=1e-5, atol=1e-6)
    assert torch.allclose(fusedm.last_losses, per_query.detach(), rtol=1e-5, atol=1e-5)
    # absolute tolerance from the scale of the whole gradient: d/d(l3.bias) = sum of ds is a
    # cancelling sum whose exact value is 0
    scale = max(float(b.grad.abs().max()) for b in plain.parameters())
    for a, b in zip(fusedm.parameters(), plain.parameters()):
        assert a.grad is not None
        assert torch.allclose(a.grad, b.grad, rtol=2e-4, atol=2e-5 * max(1.0, scale))


@pytest.mark.parametrize("shape", [(7, 160, 136, 50, 10), (5, 256, 136, 50, 10), (6, 200, 24, 13, 5),
                                   (3, 129, 80, 64, 16)])
def test_lists_of_129_to_256_documents(shape, request):
    """F <= 144: lists up to 256 documents run on the tile kernel (no parking: the forward pass of
    a fill runs again in the backward pass); all seven kinds at the guide's network."""
    if "wide" in request.node.name:
        pytest.skip("the 8-wave layout stops at 128 documents")
    B, L, F, H1, H2 = shape
    case = _case(B, L, F, H1, H2, 300 + L)
    case[2][0] = L                       # a full list
    case[2][1] = 129                     # one document in the fifth fill
    kinds = KINDS if F == 136 and L == 160 else ("hinge", "logistic", "ndcg2")
    for kind in kinds:
        _check(kind, *case)
    g = torch.Generator().manual_seed(L)
    _check("arp1", *case, grad_out=torch.rand(B, generator=g) + 0.1)


def test_scores_of_lists_of_129_to_256_documents():
    from pytorchltr_amd import fused
    dev = torch.device("cuda")
    B, L, F, H1, H2 = 40, 256, 136, 50, 10
    X, y, n, params = _case(B, L, F, H1, H2, 77)
    n[0] = L
    n[1] = 0
    n[2] = 129
    P = [p.to(dev) for p in params]
    Xd = X.to(dev)
    h = torch.relu(Xd @ P[0].t() + P[1])
    h = torch.relu(h @ P[2].t() + P[3])
    want = (h @ P[4].t() + P[5]).squeeze(-1)
    got = fused.mlp_scores(Xd, P, n.to(dev))
    valid = (torch.arange(L)[None, :] < n[:, None]).to(dev)
    assert torch.allclose(got[valid], want[valid], rtol=1e-5, atol=2e-6)
    assert not got[~valid].any()
    assert torch.allclose(fused.mlp_scores(Xd, P), want, rtol=1e-5, atol=2e-6)


def test_module_long_lists_take_the_unfused_path():
    from pytorchltr_amd.fused import FusedMLPLoss
    dev = torch.device("cuda")
    X, y, n, _ = _case(3, 300, 24, 50, 10, 4)
    m = FusedMLPLoss(24, "hinge").to(dev)
    out = m(X.to(dev), y.to(dev), n.to(dev))
    out.backward()
    assert out.dim() == 0 and m.l1.weight.grad is not None and m.last_losses.shape == (3,)


def test_argument_errors():
    import ctypes
    from pytorchltr_amd import _C, fused
    dev = torch.device("cuda")
    X, y, n, params = _case(2, 16, 6, 5, 3, 1)              # F % 4 != 0
    with pytest.raises(ValueError):
        fused.mlp_loss_step(X.to(dev), [p.to(dev) for p in params], y.to(dev), n.to(dev))
    lib = _C.lib()
    z = ctypes.c_void_p(8)
    assert lib.ltr_mlp_max_list_len(136) == 256 and lib.ltr_mlp_max_list_len(200) == 128
    assert lib.ltr_mlp_pairwise_f32(0, 1.0, z, z, z, z, z, z, z, z, 0, z, None, 1, 257, 8, 4, 4, z, None,
                                    z, None, z, 1 << 30, None) == -4        # LTR_ERR_LIST_TOO_LONG
    assert lib.ltr_mlp_pairwise_f32(0, 1.0, z, z, z, z, z, z, z, z, 0, z, None, 1, 129, 200, 4, 4, z, None,
                                    z, None, z, 1 << 30, None) == -4        # (wide rows: 128)
    assert lib.ltr_mlp_pairwise_f32(0, 1.0, z, z, z, z, z, z, z, z, 0, z, None, 1, 16, 8, 65, 4, z, None,
                                    z, None, z, 1 << 30, None) == -2        # LTR_ERR_SHAPE
    assert lib.ltr_mlp_pairwise_f32(0, 1.0, z, z, z, z, z, z, z, z, 0, z, None, 1, 16, 8, 4, 4, z, None,
                                    z, None, z, 4, None) == -5              # LTR_ERR_WORKSPACE
    assert lib.ltr_mlp_pairwise_f32(9, 1.0, z, z, z, z, z, z, z, z, 0, z, None, 1, 16, 8, 4, 4, z, None,
                                    z, None, z, 1 << 30, None) == -3        # LTR_ERR_KIND


@pytest.mark.parametrize("kind", KINDS)
def test_against_reference_fp32_outputs(kind):
    """The kernel vs what the REAL reference computed in float32 for the guide's network
    (tests/golden/mlp_vectors.npz): same tolerance class as fp32-vs-fp32 round-off."""
    import os
    from pytorchltr_amd import fused
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mlp_vectors.npz"))
    tag = "f32_guide"
    dev = torch.device("cuda")
    names = ("l1.weight", "l1.bias", "l2.weight", "l2.bias", "l3.weight", "l3.bias")
    X = torch.from_numpy(z[tag + "/X"]).to(dev)
    y = torch.from_numpy(z[tag + "/y"]).to(dev)
    n = torch.from_numpy(z[tag + "/n"]).to(dev)
    params = [torch.from_numpy(z["%s/param/%s" % (tag, k)]).to(dev) for k in names]
    lossv, grads, scores = fused.mlp_loss_step(X, params, y, n, loss=kind, return_scores=True)
    B, L = scores.shape
    valid = (np.arange(L)[None, :] < z[tag + "/n"][:, None])
    assert np.allclose(scores.cpu().numpy()[valid], z["%s/%s/scores" % (tag, kind)][valid], rtol=1e-5, atol=2e-6)
    assert np.allclose(lossv.cpu().numpy(), z["%s/%s/loss" % (tag, kind)], rtol=3e-5, atol=1e-5)
    scale = max(np.abs(z["%s/%s/grad/%s" % (tag, kind, k)]).max() for k in names)
    for got, name in zip(grads, names):
        want = z["%s/%s/grad/%s" % (tag, kind, name)]
        assert np.abs(got.cpu().numpy().reshape(want.shape) - want).max() <= 5e-5 * scale + 1e-6, (kind, name)


def test_several_scheduling_blocks_guide_network():
    """B = 2500 = two full 1024-query scheduling blocks + a partial one (n-sorted snake
    assignment per block), guide-sized network, ragged n with zeros: against the oracle."""
    X, y, n, params = _case(2500, 128, 136, 50, 10, 31)
    n[::97] = 0
    n[5::211] = 128
    _check("hinge", X, y, n, params)


def test_uniform_n_blocks_skip_the_sort():
    """All queries of a block with the same n (full lists) take the round-robin shortcut."""
    X, y, n, params = _case(1100, 64, 32, 16, 4, 9, full=True)
    _check("logistic", X, y, n, params)


@pytest.mark.parametrize("shape", [(16, 20, 136, 50, 10), (300, 128, 136, 50, 10), (5, 77, 220, 64, 16), (3, 128, 8, 3, 1)])
def test_forward_only_scores(shape):
    """ltr_mlp_scores_f32 (the evaluation half of the guide's workflow) against the three torch
    layers: fp32 round-off; padded documents score exactly 0; n=None scores everything."""
    from pytorchltr_amd import fused
    dev = torch.device("cuda")
    B, L, F, H1, H2 = shape
    X, y, n, params = _case(B, L, F, H1, H2, 55 + L)
    n[0] = L
    if B > 2:
        n[1] = 0
    P = [p.to(dev) for p in params]
    Xd = X.to(dev)
    h = torch.relu(Xd @ P[0].t() + P[1])
    h = torch.relu(h @ P[2].t() + P[3])
    want = (h @ P[4].t() + P[5]).squeeze(-1)
    got = fused.mlp_scores(Xd, P, n.to(dev))
    valid = (torch.arange(L)[None, :] < n[:, None]).to(dev)
    assert torch.allclose(got[valid], want[valid], rtol=1e-5, atol=2e-6)
    assert not got[~valid].any()
    assert torch.allclose(fused.mlp_scores(Xd, P), want, rtol=1e-5, atol=2e-6)


def test_module_score_uses_the_fused_kernel_under_no_grad():
    from pytorchltr_amd.evaluation import ndcg
    from pytorchltr_amd.fused import FusedMLPLoss
    dev = torch.device("cuda")
    X, y, n, _ = _case(32, 40, 136, 50, 10, 6)
    m = FusedMLPLoss(136, "hinge").to(dev)
    with torch.no_grad():
        fast = m.score(X.to(dev), n.to(dev))
    slow = m.score(X.to(dev))                       # grad enabled: the nn.Linear layers
    assert fast.shape == slow.shape == (32, 40, 1) and slow.requires_grad and not fast.requires_grad
    valid = (torch.arange(40)[None, :] < n[:, None]).to(dev)
    assert torch.allclose(fast.squeeze(-1)[valid], slow.detach().squeeze(-1)[valid], rtol=1e-5, atol=2e-6)
    y0 = (y * (torch.arange(40)[None, :] < n[:, None])).to(dev)
    assert torch.allclose(ndcg(fast, y0, n.to(dev), k=10), ndcg(slow.detach(), y0, n.to(dev), k=10), atol=1e-6)


def test_getting_started_example_learns():
    """examples/02_mlp_getting_started.py: the guide's workflow end to end (device split, device
    collate with UniformSampler(20), FusedMLPLoss + Adagrad, fused scoring + ndcg@10): the metric
    must rise on the synthetic learnable split."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "mlp_example", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "examples",
                                    "02_mlp_getting_started.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    trace = mod.run(epochs=3, log=lambda msg: None)
    assert len(trace) == 4
    assert trace[-1] > trace[0] + 0.05, trace


@pytest.mark.parametrize("F", [5, 46, 135])
def test_module_pads_feature_counts_that_are_not_multiples_of_four(F):
    """MQ2007 has 46 features, Example3 5: the module zero-pads features and W1 for the kernel
    and returns gradients in the original shapes -- same step as the unfused composition."""
    from pytorchltr_amd.fused import FusedMLPLoss
    from pytorchltr_amd.loss import PairwiseHingeLoss
    dev = torch.device("cuda")
    X, y, n, _ = _case(12, 30, F, 50, 10, 70 + F)
    torch.manual_seed(F)
    fusedm = FusedMLPLoss(F, "hinge").to(dev)
    plain = FusedMLPLoss(F, "hinge").to(dev)
    plain.load_state_dict(fusedm.state_dict())
    out = fusedm(X.to(dev), y.to(dev), n.to(dev))
    out.backward()
    ref = PairwiseHingeLoss()(plain.score(X.to(dev)), y.to(dev), n.to(dev)).mean()
    ref.backward()
    assert torch.allclose(out, ref, rtol=1e-5, atol=1e-6)
    scale = max(float(b.grad.abs().max()) for b in plain.parameters())
    for a, b in zip(fusedm.parameters(), plain.parameters()):
        assert a.grad.shape == b.grad.shape
        assert torch.allclose(a.grad, b.grad, rtol=2e-4, atol=2e-5 * max(1.0, scale))
    with torch.no_grad():
        assert torch.allclose(fusedm.score(X.to(dev), n.to(dev))[:, :1], plain.score(X.to(dev))[:, :1].detach(),
                              rtol=1e-5, atol=2e-6)
