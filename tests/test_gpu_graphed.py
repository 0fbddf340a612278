"""GraphedStep: a hipGraph replay of forward + backward + optimizer step must train exactly like
the eager loop (same kernels, same order)."""
import pytest
import torch

from tests.conftest import synth

pytestmark = pytest.mark.gpu


def _batches(count, B, L, F, seed):
    out = []
    for i in range(count):
        s, y, n, X, W, b = synth(B, L, seed + i, F=F)
        out.append((X, y, n))
    return out


@pytest.mark.parametrize("which", ["dropin", "fused_linear", "fused_mlp"])
def test_graphed_training_matches_eager(which):
    from pytorchltr_amd.fused import FusedLinearLoss, FusedMLPLoss
    from pytorchltr_amd.graphed import GraphedStep
    from pytorchltr_amd.loss import PairwiseLogisticLoss
    dev = torch.device("cuda")
    B, L, F = 32, 40, 24
    data = [tuple(t.to(dev) for t in b) for b in _batches(8, B, L, F, 100)]

    def make():
        torch.manual_seed(7)
        if which == "dropin":
            model = torch.nn.Linear(F, 1).to(dev)
            loss_fn = PairwiseLogisticLoss()
            return model, (lambda xs, ys, n: loss_fn(model(xs), ys, n).mean())
        if which == "fused_linear":
            model = FusedLinearLoss(F, PairwiseLogisticLoss()).to(dev)
            return model, (lambda xs, ys, n: model(xs, ys, n).mean())
        model = FusedMLPLoss(F, PairwiseLogisticLoss(), hidden=(16, 4)).to(dev)
        return model, (lambda xs, ys, n: model(xs, ys, n))

    # eager reference: 3 warm-up steps on batch 0 (GraphedStep does the same; the capture itself
    # records the step without running it), then all batches
    model_e, closure_e = make()
    opt_e = torch.optim.SGD(model_e.parameters(), lr=0.05)
    losses_e = []
    for batch in [data[0]] * 3 + data:
        opt_e.zero_grad(set_to_none=True)
        loss = closure_e(*batch)
        loss.backward()
        opt_e.step()
        losses_e.append(float(loss))

    model_g, closure_g = make()
    opt_g = torch.optim.SGD(model_g.parameters(), lr=0.05)
    step = GraphedStep(model_g, opt_g, closure_g, example_batch=data[0], warmup=3)   # 3 warm-ups + capture
    losses_g = [float(step(*batch)) for batch in data]
    assert losses_g == pytest.approx(losses_e[3:], rel=1e-5, abs=1e-6)
    for a, b in zip(model_g.parameters(), model_e.parameters()):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)
    with pytest.raises(ValueError):
        step(data[0][0][:, :-1], data[0][1][:, :-1], data[0][2])
