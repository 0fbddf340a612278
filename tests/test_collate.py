"""Collate / list-sampler row.

CPU tier: the list samplers (host code in the product, as in the reference) reproduce the
reference's index vectors draw for draw; the numpy collate oracle reproduces the reference's
padded batches; the host-side plan of RaggedQueries consumes the RNG exactly like the reference.
GPU tier: ltr_collate_pad_f32 through RaggedQueries.collate == reference batch, bit-exact.
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle.collate_oracle import collate_dense
from pytorchltr_amd.datasets.list_sampler import (BalancedRelevanceSampler, ListSampler,
                                                  UniformSampler)

HERE = os.path.dirname(os.path.abspath(__file__))
V = np.load(os.path.join(HERE, "golden", "collate_vectors.npz"))
with open(os.path.join(HERE, "golden", "collate_vectors.json")) as fh:
    CASES = json.load(fh)["cases"]
SAMPLERS = {"list": ListSampler, "uniform": UniformSampler, "balanced": BalancedRelevanceSampler}


def _sampler(case):
    kw = {}
    if case["sampler"] != "list" and case["seed"] is not None:
        kw["generator"] = torch.Generator().manual_seed(case["seed"])
    return SAMPLERS[case["sampler"]](case["max_list_size"], **kw)


def _split(case):
    s = case["split"]
    return V[s + "/xs"], V[s + "/ys"], V[s + "/offsets"], V[s + "/qids"]


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_samplers_and_oracle_match_reference(case):
    xs, ys, offsets, qids = _split(case)
    name = case["name"]
    indices = V[name + "/indices"].tolist()
    calls = [V[name + "/call%d" % k] for k in range(case["n_calls"])]
    # (1) oracle collate == reference batch, given the reference's sampler outputs
    ox, oy, on = collate_dense(xs, ys, offsets, indices, calls, case["max_list_size"])
    assert np.array_equal(ox, V[name + "/features"])
    assert np.array_equal(oy, V[name + "/relevance"])
    assert np.array_equal(on, V[name + "/n"])
    # (2) our samplers, seeded like the reference's, draw the same index vectors in the same order
    sampler = _sampler(case)
    sizes = [sampler.max_list_size(torch.as_tensor(ys[offsets[q]:offsets[q + 1]])) for q in indices]
    list_size = max(sizes)
    assert list_size == V[name + "/features"].shape[1]
    k = 0
    for q in indices:
        rel = torch.as_tensor(ys[offsets[q]:offsets[q + 1]])
        if rel.shape[0] > list_size:
            got = sampler(rel).numpy()
            assert np.array_equal(got, calls[k]), (name, k)
            k += 1
    assert k == case["n_calls"]


def test_sampler_contracts():
    rel = torch.tensor([0, 0, 0, 0, 1, 1, 2, 0, 0, 0])
    assert ListSampler().max_list_size(rel) == 10 and ListSampler(4).max_list_size(rel) == 4
    assert ListSampler(4)(rel).tolist() == [0, 1, 2, 3]
    g = torch.Generator().manual_seed(0)
    u = UniformSampler(6, generator=g)(rel)
    assert len(set(u.tolist())) == 6 and all(0 <= i < 10 for i in u.tolist())
    b = BalancedRelevanceSampler(3, generator=torch.Generator().manual_seed(1))(rel)
    assert sorted(rel[b].tolist()) == [0, 1, 2]                   # one document per grade
    b = BalancedRelevanceSampler(100, generator=torch.Generator().manual_seed(1))(rel)
    assert sorted(b.tolist()) == list(range(10))                  # everything survives a huge limit


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_device_collate_matches_reference(case):
    from pytorchltr_amd.datasets import RaggedQueries
    assert torch.cuda.is_available()
    xs, ys, offsets, qids = _split(case)
    name = case["name"]
    ragged = RaggedQueries(xs, ys, offsets, qids, device="cuda:0")
    batch = ragged.collate(V[name + "/indices"].tolist(), _sampler(case))
    assert batch.sparse is False
    assert batch.features.dtype == torch.float32 and batch.relevance.dtype == torch.int64
    assert np.array_equal(batch.features.cpu().numpy(), V[name + "/features"])      # bit-exact
    assert np.array_equal(batch.relevance.cpu().numpy(), V[name + "/relevance"])
    assert np.array_equal(batch.n.cpu().numpy(), V[name + "/n"])
    assert np.array_equal(batch.qid.cpu().numpy(), V[name + "/qid"])


@pytest.mark.gpu
def test_dataloader_training_loop_on_device_collate():
    """The reference's loop shape (examples/01-basic-usage.py:66-75) with the batch assembled on
    the device: DataLoader over query indices, collate_fn from RaggedQueries, loss + ndcg."""
    from pytorchltr_amd.datasets import RaggedQueries, UniformSampler as US
    from pytorchltr_amd.evaluation import ndcg
    from pytorchltr_amd.loss import PairwiseHingeLoss
    torch.manual_seed(0)
    g = torch.Generator().manual_seed(3)
    counts = torch.randint(5, 60, (40,), generator=g)
    offsets = torch.cat([torch.zeros(1, dtype=torch.long), counts.cumsum(0)])
    xs = torch.randn(int(offsets[-1]), 16, generator=g)
    w = torch.randn(16, generator=g)
    ys = ((xs @ w) > 0.3).long() + ((xs @ w) > 1.2).long()
    ragged = RaggedQueries(xs, ys, offsets, device="cuda:0")
    loader = torch.utils.data.DataLoader(ragged, batch_size=8, shuffle=True,
                                         collate_fn=ragged.collate_fn(US(max_list_size=32)))
    model = torch.nn.Linear(16, 1).cuda()
    opt = torch.optim.SGD(model.parameters(), lr=0.05)
    loss_fn = PairwiseHingeLoss()

    def evaluate():
        tot = 0.0
        with torch.no_grad():
            for b in torch.utils.data.DataLoader(ragged, batch_size=8, collate_fn=ragged.collate_fn()):
                tot += float(ndcg(model(b.features), b.relevance, b.n, k=10).sum())
        return tot / len(ragged)

    before = evaluate()
    for _ in range(5):
        for batch in loader:
            assert batch.features.shape[1] <= 32 and int(batch.n.max()) <= 32
            opt.zero_grad()
            loss_fn(model(batch.features), batch.relevance, batch.n).mean().backward()
            opt.step()
    assert evaluate() > before + 0.05


def _grade_histogram(sampler, relevance, position, draws=1000):
    hist = torch.zeros(3)
    for _ in range(draws):
        hist[int(relevance[sampler(relevance)][position])] += 1
    return (hist / draws).tolist()


def test_sampler_distributions_match_the_reference_expectations():
    """The distributions the reference's own sampler tests pin
    (tests/datasets/test_list_sampler.py:84-111, 158-198): a uniform sampler picks grades in
    proportion to their frequency at every output position; the balanced sampler cycles through
    the grades while they last (1/3 each, then 1/2 each once grade 2 is exhausted, then only 0)."""
    seed = 1608637542
    rel = torch.tensor([0, 0, 1, 0, 0, 0, 2, 1])
    for size, positions in ((1, [0]), (5, range(5))):
        sampler = UniformSampler(max_list_size=size, generator=torch.Generator().manual_seed(seed))
        for pos in positions:
            assert _grade_histogram(sampler, rel, pos) == pytest.approx([5 / 8, 2 / 8, 1 / 8], abs=0.05)
    sampler = BalancedRelevanceSampler(max_list_size=1, generator=torch.Generator().manual_seed(seed))
    assert _grade_histogram(sampler, rel, 0) == pytest.approx([1 / 3, 1 / 3, 1 / 3], abs=0.05)
    rel11 = torch.tensor([0, 0, 1, 0, 0, 0, 2, 1, 0, 0, 0])
    sampler = BalancedRelevanceSampler(max_list_size=7, generator=torch.Generator().manual_seed(seed))
    expected = [[1 / 3] * 3] * 3 + [[0.5, 0.5, 0.0]] * 2 + [[1.0, 0.0, 0.0]] * 2
    for pos in range(7):
        assert _grade_histogram(sampler, rel11, pos) == pytest.approx(expected[pos], abs=0.05)


def test_sampler_sizes_and_global_rng():
    rel = torch.tensor([0, 1, 0, 0, 2, 0, 1, 0, 0, 0])
    torch.manual_seed(1608637542)                                   # no generator: global RNG
    assert UniformSampler(max_list_size=5)(rel).shape == (5,)
    assert BalancedRelevanceSampler(max_list_size=5)(rel).shape == (5,)
    g = torch.Generator().manual_seed(3)
    assert UniformSampler(max_list_size=9, generator=g)(torch.tensor([0, 1])).shape == (2,)
    assert UniformSampler(max_list_size=None, generator=g)(rel).shape == (10,)
    assert BalancedRelevanceSampler(max_list_size=None, generator=g)(rel).shape == (10,)
    assert ListSampler(max_list_size=1)(torch.tensor([0])).tolist() == [0]


@pytest.mark.gpu
def test_sort_by_length_reorders_whole_queries():
    """collate(sort_by_length=True): same queries, ordered by decreasing document count (stable),
    every row still the right query."""
    from pytorchltr_amd.datasets import RaggedQueries
    g = torch.Generator().manual_seed(5)
    counts = torch.randint(1, 30, (12,), generator=g)
    offsets = torch.cat([torch.zeros(1, dtype=torch.long), torch.cumsum(counts, 0)])
    xs = torch.randn(int(offsets[-1]), 4, generator=g)
    ys = torch.randint(0, 5, (int(offsets[-1]),), generator=g)
    ds = RaggedQueries(xs, ys, offsets, qids=torch.arange(100, 112), device="cuda")
    picked = [3, 7, 0, 11, 5, 2]
    plain = ds.collate(picked)
    srt = ds.collate(picked, sort_by_length=True)
    n_sorted = srt.n.cpu()
    assert bool((n_sorted[:-1] >= n_sorted[1:]).all())
    assert sorted(srt.qid.cpu().tolist()) == sorted(plain.qid.cpu().tolist())
    for row, qid in enumerate(srt.qid.cpu().tolist()):
        src = plain.qid.cpu().tolist().index(qid)
        k = int(srt.n[row])
        assert torch.equal(srt.features[row, :k], plain.features[src, :k])
        assert torch.equal(srt.relevance[row, :k], plain.relevance[src, :k])


@pytest.mark.gpu
def test_query_indices_are_validated():
    """out-of-range indices raise IndexError (negative ones count from the end) instead
    of being clamped by the kernel."""
    import torch
    from pytorchltr_amd.datasets import RaggedQueries
    X = torch.arange(12, dtype=torch.float32).reshape(6, 2)
    y = torch.tensor([1, 0, 2, 0, 1, 1])
    ds = RaggedQueries(X, y, [0, 2, 3, 6], device="cuda:0")
    assert ds[-1] == 2 and ds[0] == 0
    b = ds.collate([-1, 0])
    assert b.n.cpu().tolist() == [3, 2]
    for bad in (3, -4, 100):
        with pytest.raises(IndexError):
            ds[bad]
        with pytest.raises(IndexError):
            ds.collate([0, bad])


# ---- the sparse branch (svmrank.py:162-176,197-202): CSR split -> dense padded batch ----
VS = np.load(os.path.join(HERE, "golden", "collate_sparse_vectors.npz"))
with open(os.path.join(HERE, "golden", "collate_sparse_vectors.json")) as fh:
    SPARSE_CASES = json.load(fh)["cases"]


def _sparse_sampler(case):
    kw = {"generator": torch.Generator().manual_seed(case["seed"])} if case["seed"] is not None else {}
    return (ListSampler if case["sampler"] == "list" else UniformSampler)(case["max_list_size"], **kw)


@pytest.mark.parametrize("case", SPARSE_CASES, ids=[c["name"] for c in SPARSE_CASES])
def test_sparse_collate_oracle_matches_reference(case):
    """The numpy restatement on the CSR split == the dense form of the reference's sparse batch (for
    truncated samples: == the reference's dense-branch batch, see oracle/collate_oracle.py:collate_csr)."""
    from oracle.collate_oracle import collate_csr
    sp, name = case["split"], case["name"]
    calls = [VS[name + "/call%d" % k] for k in range(case["n_calls"])]
    ox, oy, on = collate_csr(VS[sp + "/indptr"], VS[sp + "/indices"], VS[sp + "/values"], case["features"],
                             VS[sp + "/ys"], VS[sp + "/offsets"], VS[name + "/indices"].tolist(), calls,
                             case["max_list_size"])
    assert tuple(ox.shape) == tuple(VS[name + "/shape"])
    assert np.array_equal(ox, VS[name + "/features"])
    assert np.array_equal(oy, VS[name + "/relevance"]) and np.array_equal(on, VS[name + "/n"])


@pytest.mark.gpu
@pytest.mark.parametrize("case", SPARSE_CASES, ids=[c["name"] for c in SPARSE_CASES])
def test_device_sparse_collate_matches_reference(case):
    """ltr_collate_pad_csr_f32 through RaggedQueries(csr=...) == the reference's sparse batch, densified,
    bit for bit; relevance, n and the list size as the reference computes them."""
    from pytorchltr_amd.datasets import RaggedQueries
    sp, name = case["split"], case["name"]
    ragged = RaggedQueries(None, VS[sp + "/ys"], VS[sp + "/offsets"], device="cuda:0",
                           csr=(VS[sp + "/indptr"], VS[sp + "/indices"], VS[sp + "/values"]),
                           num_features=case["features"])
    batch = ragged.collate(VS[name + "/indices"].tolist(), _sparse_sampler(case))
    assert tuple(batch.features.shape) == tuple(VS[name + "/shape"])
    assert np.array_equal(batch.features.cpu().numpy(), VS[name + "/features"])           # bit-exact
    assert np.array_equal(batch.relevance.cpu().numpy(), VS[name + "/relevance"])
    assert np.array_equal(batch.n.cpu().numpy(), VS[name + "/n"])


@pytest.mark.gpu
def test_sparse_and_dense_storage_collate_to_the_same_batch():
    """A dense split stored as CSR (from_dense_as_csr) collates to the same batch as the dense storage,
    samplers included; duplicate entries of a (row, column) add up like torch's coalesce()."""
    from pytorchltr_amd.datasets import RaggedQueries, UniformSampler as US
    g = torch.Generator().manual_seed(8)
    counts = torch.randint(1, 50, (30,), generator=g)
    offsets = torch.cat([torch.zeros(1, dtype=torch.long), counts.cumsum(0)])
    N = int(offsets[-1])
    xs = torch.randn(N, 136, generator=g) * (torch.rand(N, 136, generator=g) < 0.2)
    ys = torch.randint(0, 5, (N,), generator=g)
    dense = RaggedQueries(xs, ys, offsets, device="cuda:0")
    sparse = RaggedQueries.from_dense_as_csr(xs, ys, offsets, device="cuda:0")
    idx = list(range(0, 30, 2))
    for limit in (None, 20):
        a = dense.collate(idx, US(limit, generator=torch.Generator().manual_seed(1)))
        b = sparse.collate(idx, US(limit, generator=torch.Generator().manual_seed(1)))
        assert torch.equal(a.features, b.features) and torch.equal(a.relevance, b.relevance) and torch.equal(a.n, b.n)
    dup = RaggedQueries(None, torch.tensor([1, 0]), torch.tensor([0, 2]), device="cuda:0",
                        csr=(torch.tensor([0, 3, 3]), torch.tensor([2, 0, 2]), torch.tensor([1.5, 4.0, 2.0])),
                        num_features=4)
    out = dup.collate([0]).features.cpu()
    assert out.tolist() == [[[4.0, 0.0, 3.5, 0.0], [0.0, 0.0, 0.0, 0.0]]]


@pytest.mark.gpu
@pytest.mark.parametrize("F", [5, 46, 137])
def test_rows_padded_to_whole_float4_take_the_vector_kernels(F):
    """feature counts that are not a multiple of 4 (Example3 = BASELINE config C1: 5, the reference's
    test file: 45, MQ2007 / MQ2008: 46) ran the scalar kernels -- no register tile.  RaggedQueries(pad_features_to=4)
    collates batches whose rows sit F4 floats apart; `features` is still the reference's (B, L, F) tensor (a view), the
    fused modules and the streaming scorer take the padded row width (the register tile where the shape fits) and
    give the same losses and gradients as the packed batch, torch.nn.Linear takes the view as it is."""
    import numpy as np
    from oracle import ltr_oracle as O
    from pytorchltr_amd import _C
    from pytorchltr_amd import loss as L_
    from pytorchltr_amd.datasets.ragged import RaggedQueries
    from pytorchltr_amd.fused import FusedLinearLoss, linear_loss_step, use_linear_scorer
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(7 + F)
    counts = torch.randint(1, 40, (24,), generator=g)
    offsets = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(counts, 0)])
    N = int(offsets[-1])
    X = torch.randn(N, F, generator=g)
    y = torch.randint(0, 5, (N,), generator=g)
    packed = RaggedQueries(X, y, offsets, device=dev)
    padded = RaggedQueries(X, y, offsets, device=dev, pad_features_to=4)
    idx = list(range(24))
    bp, bq = packed.collate(idx), padded.collate(idx)
    F4 = (F + 3) & ~3
    assert tuple(bq.features.shape) == tuple(bp.features.shape) and bq.features.stride(1) == F4 and not bq.features.is_contiguous()
    assert torch.equal(bq.features, bp.features) and torch.equal(bq.relevance, bp.relevance) and torch.equal(bq.n, bp.n)
    B, L = bq.features.shape[:2]
    assert _C.lib().ltr_linear_fused_plan(_C.HINGE, B, L, F4) == _C.PLAN_REGISTER_TILE
    lin = torch.nn.Linear(F, 1).to(dev)
    W, b = lin.weight.detach().reshape(-1), lin.bias.detach()
    want_l, want_s, want_dW, want_db = O.linear_pairwise("hinge", bp.features.cpu().numpy(), W.cpu().numpy(), float(b[0]),
                                                         bp.relevance.cpu().numpy(), bp.n.cpu().numpy(), np.full(B, 1.0 / B))
    for batch in (bp, bq):
        loss, dW, db = linear_loss_step(batch.features, W, b, batch.relevance, batch.n, loss="hinge")
        assert dW.shape == (F,)
        assert np.allclose(loss.cpu().numpy(), want_l, rtol=2e-5, atol=1e-5)
        assert np.max(np.abs(dW.cpu().numpy() - want_dW)) < 2e-5 * max(1.0, float(np.max(np.abs(want_dW))))
    # the unchanged script on the padded batch: nn.Linear, the converted layer (lazy, fused) and FusedLinearLoss agree
    loss_fn = L_.PairwiseHingeLoss()
    loss_fn(lin(bq.features), bq.relevance, bq.n).mean().backward()
    model = use_linear_scorer(torch.nn.Linear(F, 1).to(dev))
    model.load_state_dict(lin.state_dict())
    loss_fn(model(bq.features), bq.relevance, bq.n).mean().backward()
    fused = FusedLinearLoss(F, "hinge").to(dev)
    fused.load_state_dict(lin.state_dict())
    fused(bq.features, bq.relevance, bq.n).mean().backward()
    tol = 2e-5 * max(1.0, float(lin.weight.grad.abs().max()))
    for m in (model, fused):
        assert m.weight.grad.shape == lin.weight.grad.shape
        assert torch.allclose(m.weight.grad, lin.weight.grad, rtol=1e-4, atol=tol)
        assert torch.allclose(m.bias.grad, lin.bias.grad, rtol=1e-4, atol=tol)
    with torch.no_grad():
        assert torch.allclose(model(bq.features), lin(bq.features), rtol=1e-5, atol=1e-5)
    # a user's OWN view with the same strides hides whatever its tensor holds there -- NaN markers here -- and nn.Linear never
    # sees those columns: such a view is copied, not taken for a zero-padded batch
    big = torch.full((B, L, F4), float("nan"), device=dev)
    big[:, :, :F] = bq.features
    view = big[:, :, :F]
    assert not getattr(view, "_ltr_zero_padded_rows", False)
    loss2, dW2, db2 = linear_loss_step(view, W, b, bq.relevance, bq.n, loss="hinge")
    assert torch.isfinite(loss2).all() and np.allclose(loss2.cpu().numpy(), want_l, rtol=2e-5, atol=1e-5)
    with torch.no_grad():
        assert torch.allclose(model(view), lin(view), rtol=1e-5, atol=1e-5)
