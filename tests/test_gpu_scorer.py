"""LinearScorer (ltr_linear_scores_f32 / ltr_linear_grad_f32): torch.nn.Linear(F, 1) semantics on
(B, L, F) batches -- forward and weight gradients against torch in fp32 round-off, padded
documents skipped, deterministic, and the reference's user code `loss_fn(model(xs), ys, n)`
end to end against the oracle."""
import numpy as np
import pytest
import torch

from tests.conftest import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(8, 16, 5), (64, 128, 136), (7, 33, 137), (3, 1000, 220), (5, 77, 700),
                                   (2, 9, 1024), (300, 20, 46)])
def test_forward_and_gradients_match_torch_linear(shape):
    from pytorchltr_amd.fused import LinearScorer
    dev = torch.device("cuda")
    B, L, F = shape
    s, y, n, X, W, b = synth(B, L, 11 + F, F=F)
    torch.manual_seed(F)
    ours = LinearScorer(F).to(dev)
    ref = torch.nn.Linear(F, 1).to(dev)
    ref.load_state_dict(ours.state_dict())                       # state_dict compatible
    Xd = X.to(dev)
    out = ours(Xd)
    want = ref(Xd)
    assert out.shape == want.shape == (B, L, 1)
    assert torch.allclose(out, want, rtol=1e-5, atol=2e-6 * F ** 0.5)
    g = torch.randn(B, L, 1, generator=torch.Generator().manual_seed(3)).to(dev)
    (out * g).sum().backward()
    (want * g).sum().backward()
    scale = float(ref.weight.grad.abs().max())
    assert torch.allclose(ours.weight.grad, ref.weight.grad, rtol=1e-4, atol=1e-5 * max(1.0, scale))
    assert torch.allclose(ours.bias.grad, ref.bias.grad, rtol=1e-4, atol=1e-4 * max(1.0, float(ref.bias.grad.abs().max())))


def test_padded_documents_are_skipped_and_results_deterministic():
    from pytorchltr_amd.fused import LinearScorer
    dev = torch.device("cuda")
    B, L, F = 50, 64, 136
    s, y, n, X, W, b = synth(B, L, 5, F=F)
    n[0], n[1] = 0, L
    m = LinearScorer(F).to(dev)
    Xg = X.clone()
    for q in range(B):
        Xg[q, int(n[q]):] = float("nan")                         # garbage in the padded slots
    out = m(Xg.to(dev), n.to(dev))
    full = m(X.to(dev))
    valid = (torch.arange(L)[None, :] < n[:, None]).to(dev)
    assert torch.equal(out.squeeze(-1)[valid], full.squeeze(-1)[valid])
    assert not out.squeeze(-1)[~valid].any()
    g = torch.randn(B, L, 1, generator=torch.Generator().manual_seed(1)).to(dev) * valid.unsqueeze(-1)
    (out * g).sum().backward()
    first = (m.weight.grad.clone(), m.bias.grad.clone())
    m.zero_grad()
    (m(Xg.to(dev), n.to(dev)) * g).sum().backward()
    assert torch.equal(m.weight.grad, first[0]) and torch.equal(m.bias.grad, first[1])
    assert bool(torch.isfinite(m.weight.grad).all())


@pytest.mark.parametrize("kind", ["hinge", "ndcg2"])
def test_reference_user_code_against_oracle(kind):
    """loss_fn(model(xs), ys, n).mean().backward() with model = LinearScorer: loss and dW, db
    against the fp64 oracle of the same composition."""
    from oracle import ltr_oracle as O
    from pytorchltr_amd.fused import LinearScorer
    from pytorchltr_amd.loss import LambdaNDCGLoss2, PairwiseHingeLoss
    dev = torch.device("cuda")
    B, L, F = 24, 100, 136
    s, y, n, X, W, b = synth(B, L, 77, F=F)
    m = LinearScorer(F).to(dev)
    with torch.no_grad():
        m.weight.copy_(W.reshape(1, F))
        m.bias.copy_(b)
    loss_fn = {"hinge": PairwiseHingeLoss, "ndcg2": LambdaNDCGLoss2}[kind]()
    loss = loss_fn(m(X.to(dev), n.to(dev)), y.to(dev), n.to(dev))
    loss.mean().backward()
    want_l, _, want_dW, want_db = O.linear_pairwise(kind, X.numpy(), W.numpy(), float(b), y.numpy(), n.numpy(),
                                                    np.full(B, 1.0 / B))
    assert np.allclose(loss.detach().cpu().numpy(), want_l, rtol=2e-5, atol=2e-6)
    assert np.allclose(m.weight.grad.cpu().numpy().reshape(-1), want_dW, rtol=1e-4, atol=1e-5 * max(1.0, np.abs(want_dW).max()))
    assert np.allclose(m.bias.grad.cpu().numpy(), want_db, atol=1e-4 * max(1.0, abs(want_db)))


def test_fused_linear_loss_takes_the_pieces_for_a_few_long_lists():
    """FusedLinearLoss on a small batch of long lists with an NDCG loss (the cluster kernel takes the
    rank-free kinds) runs as scorer + split-query loss + weight gradient kernels; same loss and gradients as the one-kernel fused path (forced via the C ABI
    step function) and as the oracle."""
    from oracle import ltr_oracle as O
    from pytorchltr_amd.fused import FusedLinearLoss, linear_loss_step
    dev = torch.device("cuda")
    B, L, F = 80, 500, 512      # NDCG kinds on rows of 128 float4: neither the cluster nor the parts kernel takes this one
    s, y, n, X, W, b = synth(B, L, 9, F=F)
    m = FusedLinearLoss(F, "ndcg2").to(dev)
    assert m._prefer_pieces(B, L)
    assert not FusedLinearLoss(F, "logistic").to(dev)._prefer_pieces(B, L)      # cluster kernel instead
    with torch.no_grad():
        m.weight.copy_(W.reshape(1, F))
        m.bias.copy_(b)
    loss, scores = m(X.to(dev), y.to(dev), n.to(dev), return_scores=True)
    loss.mean().backward()
    ref_l, ref_dW, ref_db = linear_loss_step(X.to(dev), m.weight, m.bias, y.to(dev), n.to(dev), loss="ndcg2")
    assert torch.allclose(loss.detach(), ref_l, rtol=5e-4, atol=1e-5)
    scale = float(ref_dW.abs().max())
    assert torch.allclose(m.weight.grad.reshape(-1), ref_dW, rtol=1e-4, atol=2e-5 * max(1.0, scale))
    assert torch.allclose(m.bias.grad, ref_db, atol=1e-4 * max(1.0, scale))
    want_l, want_s, _, _ = O.linear_pairwise("ndcg2", X[:3].numpy(), W.numpy(), float(b), y[:3].numpy(),
                                             n[:3].numpy(), np.full(3, 1.0 / 3))
    assert np.allclose(loss.detach().cpu().numpy()[:3], want_l, rtol=5e-4, atol=1e-5)
    valid = np.arange(L)[None, :] < n[:3].numpy()[:, None]
    assert np.allclose(scores.cpu().numpy()[:3][valid], want_s[valid], rtol=1e-5, atol=1e-5)


def test_use_linear_scorer_swaps_the_layer_and_keeps_the_parameters():
    """the one-liner for an existing script.  nn.Linear(F, 1) scorers inside a model are
    replaced by LinearScorer modules that SHARE the parameters (same state_dict, same optimiser state);
    the training step gives the same gradients as the untouched model."""
    import torch
    from pytorchltr_amd.fused import LinearScorer, use_linear_scorer
    from pytorchltr_amd.loss import PairwiseHingeLoss
    dev = torch.device("cuda:0")
    from tests.conftest import synth
    s, y, n, X, W, b = synth(12, 40, 3, F=24)
    X, y, n = X.to(dev), y.to(dev), n.to(dev)

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.body = torch.nn.Sequential(torch.nn.Linear(24, 16), torch.nn.ReLU())
            self.head = torch.nn.Linear(16, 1)

        def forward(self, xs):
            return self.head(self.body(xs))

    torch.manual_seed(0)
    ref = Net().to(dev)
    torch.manual_seed(0)
    net = Net().to(dev)
    keys = list(net.state_dict().keys())
    head_w = net.head.weight
    use_linear_scorer(net)
    assert isinstance(net.head, LinearScorer) and isinstance(net.body[0], torch.nn.Linear)
    assert net.head.weight is head_w and list(net.state_dict().keys()) == keys
    loss_fn = PairwiseHingeLoss()
    loss_fn(ref(X), y, n).mean().backward()
    loss_fn(net(X), y, n).mean().backward()                       # the hidden layer needs grad_xs from the scorer
    for (ka, pa), (kb, pb) in zip(ref.named_parameters(), net.named_parameters()):
        assert ka == kb
        assert torch.allclose(pa.grad, pb.grad, rtol=1e-4, atol=2e-5), ka      # (d/d bias is a sum of +-1s that cancels: rounding noise)
    # the plain case: the model IS the Linear(F, 1) of examples/01-basic-usage.py
    lin = torch.nn.Linear(24, 1).to(dev)
    sc = use_linear_scorer(lin)
    assert isinstance(sc, LinearScorer) and sc.weight is lin.weight and sc.bias is lin.bias
    loss_fn(sc(X), y, n).mean().backward()
    g1 = lin.weight.grad.clone()
    lin.weight.grad = None
    lin.bias.grad = None
    loss_fn(lin(X), y, n).mean().backward()
    assert torch.allclose(g1, lin.weight.grad, rtol=1e-4, atol=1e-6)


def test_use_linear_scorer_leaves_other_one_column_layers_usable():
    """use_linear_scorer swaps every Linear(*, 1); a layer that is NOT fed the (B, L, F) feature batch -- a
    value head on a 2-D input -- must keep working (it falls back to F.linear), and a predicate can keep it out of
    the swap altogether.  Reference user code: examples/01-basic-usage.py:36."""
    from pytorchltr_amd.fused import LinearScorer, use_linear_scorer
    dev = torch.device("cuda:0")

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.scorer = torch.nn.Linear(24, 1)
            self.value_head = torch.nn.Linear(8, 1)

    net = Net().to(dev)
    ref_w = net.value_head.weight.detach().clone()
    use_linear_scorer(net)
    assert isinstance(net.scorer, LinearScorer) and isinstance(net.value_head, LinearScorer)
    h = torch.randn(5, 8, device=dev, requires_grad=True)
    out = net.value_head(h)                                   # 2-D input: the plain layer
    assert out.shape == (5, 1)
    assert torch.allclose(out, h @ ref_w.t() + net.value_head.bias)
    out.sum().backward()
    assert h.grad is not None and net.value_head.weight.grad is not None
    net2 = Net().to(dev)
    use_linear_scorer(net2, predicate=lambda name, m: name == "scorer")
    assert isinstance(net2.scorer, LinearScorer) and isinstance(net2.value_head, torch.nn.Linear)
