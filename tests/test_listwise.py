"""Listwise softmax cross-entropy.  PARITY UNPINNED: the reference has no such
loss (pytorchltr/loss/__init__.py:1-7), so there is nothing to capture golden vectors from.  The
fp64 oracle (oracle/ltr_oracle.c: oracle_listwise_softmax) is checked here against an independent
pure-Python evaluation of the definition and against central finite differences; the HIP kernel is
checked against the oracle."""
import math

import numpy as np
import pytest
import torch

from oracle import ltr_oracle as O
from tests.conftest import synth


def _definition(s, y, n):
    """-sum_j softmax(y)_j ln softmax(s)_j over j < n, term by term in python floats."""
    out = []
    for b in range(s.shape[0]):
        nb = max(0, min(int(n[b]), s.shape[1]))
        if nb == 0:
            out.append(0.0)
            continue
        zs = sum(math.exp(v) for v in s[b, :nb])
        zy = sum(math.exp(v) for v in y[b, :nb])
        out.append(-sum(math.exp(y[b, j]) / zy * (s[b, j] - math.log(zs)) for j in range(nb)))
    return np.array(out)


def test_oracle_matches_definition_and_finite_differences():
    s, y, n = synth(6, 9, 21)
    s, y, n = s.double().numpy(), y.double().numpy(), n.numpy()
    n[0], n[1] = 0, 1
    loss, ds = O.listwise_softmax(s, y, n)
    assert np.allclose(loss, _definition(s, y, n), rtol=1e-12, atol=1e-12)
    eps = 1e-6
    for b in range(s.shape[0]):
        for j in range(s.shape[1]):
            sp, sm = s.copy(), s.copy()
            sp[b, j] += eps
            sm[b, j] -= eps
            fd = (O.listwise_softmax(sp, y, n)[0][b] - O.listwise_softmax(sm, y, n)[0][b]) / (2 * eps)
            assert abs(fd - ds[b, j]) < 1e-7
    assert loss[0] == 0.0 and np.all(ds[0] == 0.0)
    assert loss[1] == pytest.approx(0.0, abs=1e-15)          # one document: P = 1, ln 1 = 0
    # padded slots never matter
    s2 = s.copy()
    for b in range(s.shape[0]):
        s2[b, n[b]:] = 1e6
    assert np.array_equal(O.listwise_softmax(s2, y, n)[0], loss)


def test_oracle_minimum_is_where_the_distributions_agree():
    y = np.array([[0.0, 1.0, 3.0, 2.0]])
    n = np.array([4])
    loss, ds = O.listwise_softmax(y + 7.0, y, n)             # softmax is shift invariant
    assert np.allclose(ds, 0.0, atol=1e-15)
    p = np.exp(y[0]) / np.exp(y[0]).sum()
    assert loss[0] == pytest.approx(-(p * np.log(p)).sum(), rel=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(1, 1), (3, 2), (8, 16), (33, 100), (1024, 128), (5, 1000), (2, 4096)])
def test_kernel_vs_oracle(shape):
    from pytorchltr_amd._autograd import LISTWISE_SOFTMAX, pairwise_loss_and_grad
    B, L = shape
    dev = torch.device("cuda:0")
    s, y, n = synth(B, L, 5)
    if B > 2:
        n[0], n[1] = 0, 1
    for labels in (y, y.int(), y.float() * 0.5):
        loss, ds = pairwise_loss_and_grad(s.to(dev), labels.to(dev), n.to(dev), LISTWISE_SOFTMAX)
        want_l, want_d = O.listwise_softmax(s.numpy(), labels.numpy(), n.numpy())
        assert np.allclose(loss.cpu().numpy(), want_l, rtol=1e-5, atol=2e-6)
        assert np.allclose(ds.cpu().numpy(), want_d, rtol=1e-5, atol=1e-6)
        k = torch.arange(L)[None, :] >= n[:, None]
        assert torch.all(ds.cpu()[k] == 0)


@pytest.mark.gpu
def test_module_autograd_mean_and_sum():
    from pytorchltr_amd.loss import ListNetLoss, ListwiseSoftmaxLoss
    assert ListNetLoss is ListwiseSoftmaxLoss
    dev = torch.device("cuda:0")
    B, L = 64, 40
    s, y, n = synth(B, L, 6)
    want_l, want_d = O.listwise_softmax(s.numpy(), y.numpy(), n.numpy())
    for shape3 in (False, True):
        sc = s.to(dev).reshape(B, L, 1) if shape3 else s.to(dev)
        sc = sc.clone().requires_grad_(True)
        loss = ListwiseSoftmaxLoss()(sc, y.to(dev), n.to(dev))
        assert loss.shape == (B,)
        loss.mean().backward()
        assert sc.grad.shape == sc.shape
        assert np.allclose(sc.grad.cpu().numpy().reshape(B, L), want_d / B, rtol=1e-5, atol=1e-7)
        sc.grad = None
        ListwiseSoftmaxLoss()(sc, y.to(dev), n.to(dev)).sum().backward()      # expanded-scalar gradient
        assert np.allclose(sc.grad.cpu().numpy().reshape(B, L), want_d, rtol=1e-5, atol=1e-6)
    assert np.allclose(loss.detach().cpu().numpy(), want_l, rtol=1e-5, atol=2e-6)


@pytest.mark.gpu
def test_listwise_loss_takes_fp64_scores_and_long_lists():
    """its own autograd Function -- fp64 scores are computed in fp32 and cast back (the seven
    reference losses accept fp64 too), and lists beyond the pairwise kernels' limit are fine (one wave
    walks the list)."""
    from pytorchltr_amd import _C
    from pytorchltr_amd.loss import ListwiseSoftmaxLoss
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    L = _C.max_list_len() + 500
    s = torch.randn(3, L, generator=g, dtype=torch.float64)
    y = torch.randint(0, 5, (3, L), generator=g)
    n = torch.tensor([L, 17, 0])
    sd = s.to(dev).requires_grad_(True)
    out = ListwiseSoftmaxLoss()(sd, y.to(dev), n.to(dev))
    assert out.dtype == torch.float64
    out.sum().backward()
    assert sd.grad.dtype == torch.float64 and sd.grad.shape == (3, L)
    want_l, want_g = O.listwise_softmax(s.numpy(), y.numpy(), n.numpy())
    assert np.allclose(out.detach().cpu().numpy(), want_l, rtol=1e-5, atol=1e-6)
    assert np.allclose(sd.grad.cpu().numpy(), want_g, rtol=1e-4, atol=1e-7)
