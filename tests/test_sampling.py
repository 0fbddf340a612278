"""Sampling row: Plackett-Luce ranking sampler and PBM click simulator.

CPU tier: the numpy oracle reproduces the reference's rankings (given the same uniform draw) and
its propensities / deterministic clicks.  GPU tier: the HIP kernels equal the oracle given the
same draw, reproduce the reference vectors, and pass the reference's own Monte-Carlo tests
(tests/utils/test_tensor_operations.py, tests/click_simulation/test_pbm.py: 100 runs, abs 0.1).
"""
import os

import numpy as np
import pytest
import torch

from oracle import sampling_oracle as S

HERE = os.path.dirname(os.path.abspath(__file__))
V = np.load(os.path.join(HERE, "golden", "sampling_vectors.npz"))
PBM_TAGS = {"perfect": ([0.0, 0.2, 0.4, 0.8, 1.0], None, 0.0),
            "perfect_cut3": ([0.0, 0.2, 0.4, 0.8, 1.0], 3, 0.0),
            "position": ([0.1, 0.1, 0.1, 1.0, 1.0], None, 1.0),
            "position_eta2_cut5": ([0.1, 0.1, 0.1, 1.0, 1.0], 5, 2.0),
            "nearrandom_eta0": ([0.4, 0.45, 0.5, 0.55, 0.6], None, 0.0)}


@pytest.mark.parametrize("name", ["pl_small", "pl_c2"])
def test_oracle_plackettluce_matches_reference(name):
    s, n, u, ref = (V["%s/%s" % (name, k)] for k in ("scores", "n", "u", "ranking"))
    got = S.plackettluce_ranking(s, n, u)
    for b in range(s.shape[0]):
        assert np.array_equal(got[b, :n[b]], ref[b, :n[b]])
        assert sorted(got[b, n[b]:].tolist()) == list(range(n[b], s.shape[1]))


@pytest.mark.parametrize("name", ["pbm_small", "pbm_c2"])
def test_oracle_pbm_matches_reference(name):
    rk, ys, n = V[name + "/rankings"], V[name + "/ys"], V[name + "/n"]
    for tag, (probs, cutoff, eta) in PBM_TAGS.items():
        _, props = S.pbm(rk, ys, n, probs, cutoff, eta)
        assert np.allclose(props, V["%s/%s/props" % (name, tag)], rtol=1e-6, atol=1e-7), tag
    _, props = S.pbm(rk, ys, n, V[name + "/custom_probs"], None, 0.5)
    assert np.allclose(props, V[name + "/custom_eta05/props"], rtol=1e-6, atol=1e-7)
    u = np.full(rk.shape, 0.5)
    clicks, props = S.pbm_clicks(rk, V[name + "/ys04"], n, [0.0, 0.2, 0.4, 0.8, 1.0], u, 4, 0.0)
    assert np.array_equal(clicks, V[name + "/perfect04_cut4/clicks"])
    assert np.allclose(props, V[name + "/perfect04_cut4/props"])


# ------------------------------------------------------------------------------- GPU tier
def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["pl_small", "pl_c2"])
def test_device_plackettluce_given_uniform(name):
    from pytorchltr_amd.utils.tensor_operations import _plackettluce_from_uniform
    dev = _dev()
    s, n, u, ref = (V["%s/%s" % (name, k)] for k in ("scores", "n", "u", "ranking"))
    got = _plackettluce_from_uniform(torch.as_tensor(s).to(dev), torch.as_tensor(n).to(dev),
                                     torch.as_tensor(u).to(dev)).cpu().numpy()
    want = S.plackettluce_ranking(s, n, u)
    for b in range(s.shape[0]):
        assert np.array_equal(got[b, :n[b]], ref[b, :n[b]])            # == the reference
        assert np.array_equal(got[b], want[b])                          # == oracle incl. the tail


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["pbm_small", "pbm_c2"])
def test_device_pbm_given_uniform(name):
    from pytorchltr_amd.click_simulation.pbm import _simulate_pbm_from_uniform
    dev = _dev()
    rk, ys, n = (torch.as_tensor(V[name + "/" + k]).to(dev) for k in ("rankings", "ys", "n"))
    g = np.random.RandomState(3)
    u = g.rand(*V[name + "/rankings"].shape).astype(np.float32)
    for tag, (probs, cutoff, eta) in PBM_TAGS.items():
        clicks, props = _simulate_pbm_from_uniform(rk, ys, n, torch.tensor(probs), torch.as_tensor(u),
                                                   cutoff, eta)
        assert clicks.dtype == torch.int64 and props.dtype == torch.float32
        assert np.allclose(props.cpu().numpy(), V["%s/%s/props" % (name, tag)], rtol=1e-6, atol=1e-7), tag
        want_c, _ = S.pbm_clicks(V[name + "/rankings"], V[name + "/ys"], V[name + "/n"], probs, u, cutoff, eta)
        p_rank, _ = S.pbm(V[name + "/rankings"], V[name + "/ys"], V[name + "/n"], probs, cutoff, eta)
        safe = np.abs(u - p_rank) > 1e-6                # away from the fp32/fp64 decision boundary
        got_rank = np.take_along_axis(clicks.cpu().numpy(), V[name + "/rankings"], axis=1)
        want_rank = np.take_along_axis(want_c, V[name + "/rankings"], axis=1)
        assert np.array_equal(got_rank[safe], want_rank[safe]), tag
    ys04 = torch.as_tensor(V[name + "/ys04"]).to(dev)
    clicks, props = _simulate_pbm_from_uniform(rk, ys04, n, torch.tensor([0.0, 0.2, 0.4, 0.8, 1.0]),
                                               torch.full(rk.shape, 0.5), 4, 0.0)
    assert np.array_equal(clicks.cpu().numpy(), V[name + "/perfect04_cut4/clicks"])   # == reference


def _rank_stats(fn, runs=100):
    shape = fn().shape
    out = torch.zeros((shape[0], shape[1], shape[1]))
    for _ in range(runs):
        ranking = fn().cpu()
        for b in range(shape[0]):
            for j, r_j in enumerate(ranking[b, :]):
                out[b, j, r_j] += 1
    return out / runs


@pytest.mark.gpu
def test_plackettluce_statistics_like_the_reference_tests():
    """tests/utils/test_tensor_operations.py: rank-1 frequencies follow softmax(scores); padded
    documents always come last; 3-d scores are accepted."""
    from pytorchltr_amd.utils import rank_by_plackettluce
    dev = _dev()
    torch.manual_seed(42)
    scores = torch.tensor([[5.0, 3.0, 2.0, 1.0], [10.0, 3.0, 10.0, 100.0]], device=dev)
    n = torch.tensor([4, 4], dtype=torch.int32, device=dev)
    out = _rank_stats(lambda: rank_by_plackettluce(scores, n), runs=200)
    expected = torch.softmax(scores.cpu(), dim=1)
    assert out[0, 0, :].numpy() == pytest.approx(expected[0].numpy(), abs=0.1)
    assert out[1, 0, :].numpy() == pytest.approx(expected[1].numpy(), abs=0.1)
    scores = torch.tensor([[5.0, 3.0, 2.0, 1.0, 10.0]], device=dev)
    out = _rank_stats(lambda: rank_by_plackettluce(scores.reshape(1, 5, 1), torch.tensor([4], device=dev)), runs=50)
    assert out[0, 4, :].tolist() == [0.0, 0.0, 0.0, 0.0, 1.0]


@pytest.mark.gpu
def test_click_simulators_monte_carlo_like_the_reference_tests():
    """tests/click_simulation/test_pbm.py:8-57 and :112-131 (perfect / position models)."""
    from pytorchltr_amd.click_simulation import simulate_perfect, simulate_position
    dev = _dev()
    rankings = torch.tensor([[3, 4, 0, 2, 1], [1, 0, 2, 4, 3]], device=dev)
    ys = torch.tensor([[1, 0, 4, 0, 2], [4, 3, 0, 0, 0]], device=dev)
    n = torch.tensor([5, 3], device=dev)
    torch.manual_seed(4200)

    def monte_carlo(fn, nr=300):
        c = torch.zeros(2, 5)
        p = torch.zeros(2, 5)
        for _ in range(nr):
            clicks, props = fn(rankings, ys, n)
            c += clicks.float().cpu()
            p += props.cpu()
        return c / nr, p / nr

    rel = torch.tensor([[0.2, 0.0, 1.0, 0.0, 0.4], [1.0, 0.8, 0.0, 0.0, 0.0]])
    clicks, props = monte_carlo(simulate_perfect)
    want_p = torch.tensor([[1.0] * 5, [1.0, 1.0, 1.0, 0.0, 0.0]])
    assert props.numpy() == pytest.approx(want_p.numpy(), abs=1e-6)
    assert clicks.numpy() == pytest.approx((rel * want_p).numpy(), abs=0.1)
    clicks, props = monte_carlo(lambda r, y, m: simulate_perfect(r, y, m, cutoff=2))
    want_p = torch.tensor([[0.0, 0.0, 0.0, 1.0, 1.0], [1.0, 1.0, 0.0, 0.0, 0.0]])
    assert props.numpy() == pytest.approx(want_p.numpy(), abs=1e-6)
    rel = torch.tensor([[0.1, 0.1, 1.0, 0.1, 0.1], [1.0, 1.0, 0.1, 0.1, 0.1]])
    clicks, props = monte_carlo(simulate_position)
    want_p = torch.tensor([[1 / 4.0, 1 / 6.0, 1 / 5.0, 1 / 2.0, 1 / 3.0], [1 / 3.0, 1 / 2.0, 1 / 4.0, 0.0, 0.0]])
    assert props.numpy() == pytest.approx(want_p.numpy(), abs=1e-6)
    assert clicks.numpy() == pytest.approx((rel * want_p).numpy(), abs=0.1)


@pytest.mark.gpu
def test_pbm_rejects_labels_and_rankings_out_of_range():
    """the reference's gathers raise on bad indices; no silent clamping here either."""
    import torch
    from pytorchltr_amd.click_simulation import simulate_pbm
    dev = torch.device("cuda:0")
    rk = torch.tensor([[0, 1, 2]], device=dev)
    n = torch.tensor([3], device=dev)
    probs = torch.tensor([0.1, 0.5], device=dev)
    simulate_pbm(rk, torch.tensor([[0, 1, 1]], device=dev), n, probs)
    with pytest.raises(IndexError):
        simulate_pbm(rk, torch.tensor([[0, 2, 1]], device=dev), n, probs)          # label 2: no probability
    with pytest.raises(IndexError):
        simulate_pbm(torch.tensor([[0, 1, 3]], device=dev), torch.tensor([[0, 1, 1]], device=dev), n, probs)


@pytest.mark.gpu
def test_mask_padded_values_mutates_any_dtype():
    import torch
    from pytorchltr_amd.utils import mask_padded_values
    dev = torch.device("cuda:0")
    n = torch.tensor([1, 3], device=dev)
    for dtype in (torch.float64, torch.float16, torch.float32):
        x = torch.ones(2, 3, dtype=dtype, device=dev)
        out = mask_padded_values(x, n, mask_value=0.0, mutate=True)
        assert out is x and x.cpu().tolist() == [[1.0, 0.0, 0.0], [1.0, 1.0, 1.0]]
    xt = torch.ones(3, 2, device=dev).t()                                            # non-contiguous view
    mask_padded_values(xt, n, mask_value=-1.0, mutate=True)
    assert xt.cpu().tolist() == [[1.0, -1.0, -1.0], [1.0, 1.0, 1.0]]
