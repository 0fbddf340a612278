"""CPU tier: the materialising torch port timed by bench.py's cpu_baseline leg computes the
same thing as the C oracle and as the real reference (golden vectors)."""
import numpy as np
import pytest
import torch

from oracle import ltr_oracle as O
from oracle import materialized_torch as M
from tests.conftest import load_golden, synth

G = load_golden()


@pytest.mark.parametrize("kind", M.KINDS)
def test_port_matches_oracle(kind):
    s, y, n = synth(12, 33, 4)
    n[0] = 0
    n[1] = 33
    sigma = 1.0 if kind in ("hinge", "dcg_hinge") else 1.7
    loss, grad = M.loss_step(kind, s, y, n, sigma)
    want_l, want_g = O.pairwise_loss(kind, s.numpy(), y.numpy(), n.numpy(), sigma=sigma)
    assert np.allclose(loss.numpy(), want_l, rtol=2e-5, atol=2e-6)
    assert np.allclose(grad.numpy() * 12, want_g, rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("name", ["syn_b8_l16", "ut_doc_hinge", "edge_n_rows"])
def test_port_matches_reference_vectors(name):
    case = G.cases[name]
    s, y, n = (torch.as_tensor(a) for a in G.inputs(name))
    for kind in case["kinds"]:
        loss, _ = M.loss_step(kind, s, y, n, case["sigma"])
        assert np.allclose(loss.numpy(), G.get(name, kind + "/loss32"), rtol=1e-5, atol=2e-6), kind


def test_port_linear_step_and_metrics():
    s, y, n, X, W, b = synth(8, 16, 1234, F=5)
    loss, dW, db = M.linear_step("hinge", X, W, b, y, n)
    name = "syn_linear_b8_l16_f5"
    assert np.allclose(loss.numpy(), G.get(name, "hinge/loss"), rtol=1e-5, atol=2e-6)
    assert np.allclose(dW.numpy(), G.get(name, "hinge/dW"), rtol=1e-4, atol=1e-5)
    y0 = y * (torch.arange(16)[None, :] < n[:, None])
    assert np.allclose(M.ndcg(s, y0, n, k=10).numpy(), O.ndcg(s.numpy(), y0.numpy(), n.numpy(), k=10),
                       rtol=1e-5, atol=1e-6)
    assert np.allclose(M.arp(s, y0, n).numpy(), O.arp(s.numpy(), y0.numpy(), n.numpy()), rtol=1e-5)
