"""GPU tier: pytorchltr_amd.optim.SGD -- the reference's loop body (examples/01-basic-usage.py:66-75) with only the optimizer's
import line changed reaches the ONE-launch lazy step: same weights as torch.optim.SGD on the same model, and as the C ABI's
lazy step bit for bit; gradients and weights are right whenever anybody looks (flush on read)."""
import copy

import numpy as np
import pytest
import torch

from tests.conftest import synth

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda")


def _data(count, B, L, F, seed, dev):
    out = []
    for i in range(count):
        s, y, n, X, W, b = synth(B, L, seed + i, F=F)
        out.append((X.to(dev), y.to(dev), n.to(dev)))
    return out


def _models(F, dev, seed=3):
    from pytorchltr_amd.fused import use_linear_scorer
    torch.manual_seed(seed)
    lin = torch.nn.Linear(F, 1).to(dev)
    a = use_linear_scorer(copy.deepcopy(lin))
    b = use_linear_scorer(copy.deepcopy(lin))
    return lin, a, b


def _loop(model, opt, loss_fn, data):
    """The reference's loop body, literally (examples/01-basic-usage.py:70-75)."""
    losses = []
    for xs, ys, n in data:
        loss = loss_fn(model(xs), ys, n).mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(loss.detach())
    return torch.stack(losses)


@pytest.mark.parametrize("kind,shape", [("hinge", (64, 128, 136)), ("ndcg2", (48, 100, 136)), ("logistic", (40, 60, 24)),
                                        ("hinge", (16, 20, 45))])
def test_same_training_as_torch_sgd(kind, shape):
    from pytorchltr_amd import loss as L_
    from pytorchltr_amd.optim import SGD, LazyGrad
    dev = _dev()
    B, L, F = shape
    loss_fn = {"hinge": L_.PairwiseHingeLoss, "ndcg2": L_.LambdaNDCGLoss2, "logistic": L_.PairwiseLogisticLoss}[kind]()
    data = _data(7, B, L, F, 500, dev)
    _, m_ref, m_lazy = _models(F, dev)
    o_ref = torch.optim.SGD(m_ref.parameters(), lr=0.03)
    o_lazy = SGD(m_lazy.parameters(), lr=0.03)
    l_ref = _loop(m_ref, o_ref, loss_fn, data)
    l_lazy = _loop(m_lazy, o_lazy, loss_fn, data)
    assert torch.allclose(l_lazy, l_ref, rtol=2e-5, atol=1e-6)
    # the last update is still pending -- and reading the parameters applies it
    st = m_lazy.weight._ltr_lazy
    if F % 4 == 0:
        assert st.pending is not None and isinstance(m_lazy.weight.grad, LazyGrad)
    w = m_lazy.weight.detach().clone()
    assert st.pending is None
    assert torch.allclose(w, m_ref.weight.detach(), rtol=1e-5, atol=1e-6)
    assert torch.allclose(m_lazy.bias.detach(), m_ref.bias.detach(), rtol=1e-5, atol=1e-6)
    # .grad after the step: the gradient that step applied
    assert torch.allclose(m_lazy.weight.grad, m_ref.weight.grad, rtol=1e-4, atol=1e-5 * max(1.0, float(m_ref.weight.grad.abs().max())))
    assert m_lazy.weight.grad.shape == m_ref.weight.grad.shape and m_lazy.bias.grad.shape == m_ref.bias.grad.shape


def test_lazy_path_is_the_c_abi_lazy_step_bit_for_bit():
    from pytorchltr_amd import loss as L_
    from pytorchltr_amd.fused import LazySGD
    from pytorchltr_amd.optim import SGD
    dev = _dev()
    B, L, F = 96, 128, 136
    data = _data(6, B, L, F, 900, dev)
    _, _, model = _models(F, dev)
    W = model.weight.detach().clone().reshape(F)
    b = model.bias.detach().clone()
    opt = SGD(model.parameters(), lr=0.02)
    # ONE launch per step: nothing in the loop body may read the parameters' values (that would apply the pending update in a
    # launch of its own) or the gradients' (a reduction launch) -- count both
    from pytorchltr_amd import optim as O_
    counts = {"flush": 0, "grad": 0}
    real_flush, real_grad = O_._LazyLinear.flush, O_._LazyLinear.gradient_of

    def flush(self):
        counts["flush"] += 1 if self.pending is not None else 0
        return real_flush(self)

    def gradient_of(self, pg):
        counts["grad"] += 1
        return real_grad(self, pg)
    O_._LazyLinear.flush, O_._LazyLinear.gradient_of = flush, gradient_of
    try:
        _loop(model, opt, L_.PairwiseHingeLoss(), data)
    finally:
        O_._LazyLinear.flush, O_._LazyLinear.gradient_of = real_flush, real_grad
    assert counts == {"flush": 0, "grad": 0}, counts
    ref = LazySGD(W, b, 0.02, loss="hinge")
    for xs, ys, n in data:
        ref.step(xs, ys, n)
    ref.flush()
    assert torch.equal(model.weight.detach().reshape(F), W) and torch.equal(model.bias.detach(), b)


def test_gradients_read_before_the_step_and_the_step_still_right():
    from pytorchltr_amd import loss as L_
    from pytorchltr_amd.optim import SGD, LazyGrad
    dev = _dev()
    B, L, F = 32, 64, 136
    data = _data(4, B, L, F, 40, dev)
    loss_fn = L_.PairwiseHingeLoss()
    _, m_ref, m_lazy = _models(F, dev)
    o_ref, o_lazy = torch.optim.SGD(m_ref.parameters(), lr=0.05), SGD(m_lazy.parameters(), lr=0.05)
    for i, (xs, ys, n) in enumerate(data):
        for m, o in ((m_ref, o_ref), (m_lazy, o_lazy)):
            o.zero_grad()
            loss_fn(m(xs), ys, n).mean().backward()
        assert isinstance(m_lazy.weight.grad, LazyGrad)
        if i % 2 == 0:
            # gradient clipping / logging before the step: the values are there (a reduction launch of its own), and the step that
            # follows is torch's own on the materialised gradient
            gn = torch.nn.utils.clip_grad_norm_(m_lazy.parameters(), 1e9)
            gr = torch.nn.utils.clip_grad_norm_(m_ref.parameters(), 1e9)
            assert torch.allclose(gn, gr, rtol=1e-4)
        o_ref.step()
        o_lazy.step()
        assert torch.allclose(m_lazy.weight.detach(), m_ref.weight.detach(), rtol=1e-5, atol=1e-6), i


def test_flush_on_read_evaluation_state_dict_and_print():
    from pytorchltr_amd import loss as L_
    from pytorchltr_amd.evaluation import ndcg
    from pytorchltr_amd.optim import SGD
    dev = _dev()
    B, L, F = 32, 64, 136
    data = _data(3, B, L, F, 70, dev)
    loss_fn = L_.PairwiseHingeLoss()
    _, m_ref, m_lazy = _models(F, dev)
    o_ref, o_lazy = torch.optim.SGD(m_ref.parameters(), lr=0.05), SGD(m_lazy.parameters(), lr=0.05)
    for xs, ys, n in data:
        _loop(m_ref, o_ref, loss_fn, [(xs, ys, n)])
        _loop(m_lazy, o_lazy, loss_fn, [(xs, ys, n)])
        assert m_lazy.weight._ltr_lazy.pending is not None
        with torch.no_grad():                                  # an evaluation pass between the steps sees the updated weights
            a = ndcg(m_lazy(xs), ys, n, k=10)
            b = ndcg(m_ref(xs), ys, n, k=10)
        assert m_lazy.weight._ltr_lazy.pending is None
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)
    _loop(m_lazy, o_lazy, loss_fn, data[:1])
    _loop(m_ref, o_ref, loss_fn, data[:1])
    sd = m_lazy.state_dict()
    assert torch.allclose(sd["weight"], m_ref.state_dict()["weight"], rtol=1e-5, atol=1e-6)
    _loop(m_lazy, o_lazy, loss_fn, data[1:2])
    _loop(m_ref, o_ref, loss_fn, data[1:2])
    assert m_lazy.weight._ltr_lazy.pending is not None
    assert "Parameter containing" in repr(m_lazy.weight) and m_lazy.weight._ltr_lazy.pending is None
    assert np.allclose(m_lazy.weight.detach().cpu().numpy(), m_ref.weight.detach().cpu().numpy(), rtol=1e-5, atol=1e-6)
    assert isinstance(m_lazy.weight, torch.nn.Parameter) and list(m_lazy.state_dict()) == ["weight", "bias"]


def test_everything_that_is_not_plain_sgd_runs_as_torch_does():
    from pytorchltr_amd import loss as L_
    from pytorchltr_amd.optim import SGD
    dev = _dev()
    B, L, F = 24, 50, 136
    data = _data(4, B, L, F, 11, dev)
    loss_fn = L_.PairwiseLogisticLoss()
    for kw in ({"momentum": 0.9}, {"weight_decay": 0.01}, {"momentum": 0.5, "nesterov": True}):
        _, m_ref, m_lazy = _models(F, dev)
        o_ref, o_lazy = torch.optim.SGD(m_ref.parameters(), lr=0.02, **kw), SGD(m_lazy.parameters(), lr=0.02, **kw)
        _loop(m_ref, o_ref, loss_fn, data)
        _loop(m_lazy, o_lazy, loss_fn, data)
        assert m_lazy.weight._ltr_lazy.pending is None
        assert torch.allclose(m_lazy.weight.detach(), m_ref.weight.detach(), rtol=1e-5, atol=1e-6), kw
    # per-query weights on the loss (an upstream gradient that is not one scalar), gradient accumulation, zero_grad(set_to_none=False)
    _, m_ref, m_lazy = _models(F, dev)
    o_ref, o_lazy = torch.optim.SGD(m_ref.parameters(), lr=0.02), SGD(m_lazy.parameters(), lr=0.02)
    wq = torch.rand(B, device=dev)
    for m, o in ((m_ref, o_ref), (m_lazy, o_lazy)):
        for i, (xs, ys, n) in enumerate(data):
            o.zero_grad(set_to_none=(i % 2 == 0))
            (loss_fn(m(xs), ys, n) * wq).sum().backward()
            loss_fn(m(data[0][0]), data[0][1], data[0][2]).mean().backward()          # accumulates
            o.step()
    assert torch.allclose(m_lazy.weight.detach(), m_ref.weight.detach(), rtol=2e-5, atol=2e-6)
    # a learning-rate schedule: the lr of the step that recorded the update is the one applied
    _, m_ref, m_lazy = _models(F, dev)
    o_ref, o_lazy = torch.optim.SGD(m_ref.parameters(), lr=0.05), SGD(m_lazy.parameters(), lr=0.05)
    s_ref = torch.optim.lr_scheduler.StepLR(o_ref, 1, gamma=0.5)
    s_lazy = torch.optim.lr_scheduler.StepLR(o_lazy, 1, gamma=0.5)
    for xs, ys, n in data:
        _loop(m_ref, o_ref, loss_fn, [(xs, ys, n)])
        _loop(m_lazy, o_lazy, loss_fn, [(xs, ys, n)])
        s_ref.step()
        s_lazy.step()
    assert torch.allclose(m_lazy.weight.detach(), m_ref.weight.detach(), rtol=1e-5, atol=1e-6)


def test_example3_trace_with_the_lazy_optimizer():
    """BASELINE.json configs[0] end to end (examples/01_basic_usage.py with `model = use_linear_scorer(model)` and this
    optimizer): the same trace as with torch.optim.SGD on nn.Linear -- test nDCG@10 0.8617 at the start, 1.0 at the end."""
    import importlib.util
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("basic_usage", os.path.join(here, "..", "examples", "01_basic_usage.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    plain = mod.run(log=lambda *_: None)
    fast = mod.run(log=lambda *_: None, fast=True)
    assert fast == pytest.approx(plain, abs=1e-6)
    assert fast[0] == pytest.approx(0.8617, abs=1e-4) and fast[-1] == pytest.approx(1.0, abs=1e-6)


def test_graphed_step_with_the_lazy_optimizer_matches_eager():
    from pytorchltr_amd import loss as L_
    from pytorchltr_amd.graphed import GraphedStep
    from pytorchltr_amd.optim import SGD
    dev = _dev()
    B, L, F = 32, 64, 136
    data = _data(6, B, L, F, 300, dev)
    loss_fn = L_.PairwiseHingeLoss()
    _, m_ref, m_g = _models(F, dev)
    o_ref = torch.optim.SGD(m_ref.parameters(), lr=0.05)
    _loop(m_ref, o_ref, loss_fn, [data[0]] * 3 + data)
    o_g = SGD(m_g.parameters(), lr=0.05)
    step = GraphedStep(m_g, o_g, lambda xs, ys, n: loss_fn(m_g(xs), ys, n).mean(), example_batch=data[0], warmup=3)
    for batch in data:
        step(*batch)
    assert torch.allclose(m_g.weight.detach(), m_ref.weight.detach(), rtol=1e-5, atol=1e-6)
    assert torch.allclose(m_g.bias.detach(), m_ref.bias.detach(), rtol=1e-5, atol=1e-6)
