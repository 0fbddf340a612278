"""Pin the CPU oracle (oracle/ltr_oracle.c) to the reference.

Three independent anchors:
  1. the literal known answers asserted by the reference's own unit tests and docs
     (stored as `literal` in tests/golden/reference_vectors.json),
  2. the reference's outputs (fp32 and fp64 runs, incl. autograd gradients) captured by
     tests/golden/generate_golden.py,
  3. a pure-Python pair-by-pair re-derivation of every loss (small cases only), written
     from the formulas in the reference's docstrings/tests, with central finite
     differences as an independent check of the analytic gradients.
"""
import math

import numpy as np
import pytest

from oracle import ltr_oracle as O
from tests.conftest import load_golden

G = load_golden()
LOSS_CASES = [c["name"] for c in G.by_op("loss")]
METRIC_CASES = [c["name"] for c in G.by_op("metrics")]
RANK_FREE = {"hinge", "dcg_hinge", "logistic", "arp1", "arp2"}   # do not depend on the ranking


def _rel_err(a, b, floor=1e-12):
    return float(np.max(np.abs(a - b) / (np.abs(b) + floor))) if a.size else 0.0


def _grad_err(g, ref):
    ref = ref.reshape(g.shape)
    scale = np.max(np.abs(ref), axis=1, keepdims=True) + 1e-12
    return float(np.max(np.abs(g - ref) / scale)) if g.size else 0.0


@pytest.mark.parametrize("name", LOSS_CASES)
def test_losses_match_reference_vectors(name):
    case = G.cases[name]
    s, y, n = G.inputs(name)
    for kind in case["kinds"]:
        loss, grad = O.pairwise_loss(kind, s, y, n, sigma=case["sigma"])
        l64 = G.get(name, kind + "/loss64")
        l32 = G.get(name, kind + "/loss32")
        # fp64 run of the reference: additive/ARP agree to rounding; the NDCG losses keep
        # fp32 discount/gain tables inside the reference even for fp64 scores -> ~1e-7.
        tol64 = 1e-6 if kind in ("ndcg1", "ndcg2") else 1e-11
        assert np.allclose(loss, l64, rtol=tol64, atol=1e-12), (kind, _rel_err(loss, l64))
        # fp32 run: stated tolerance rtol 1e-5 (L<=128) / 5e-4 (L=1000), atol 1e-6;
        # the "perfect" orderings carry e^-10..e^-20 terms the fp32 reference rounds away
        # (its own test uses abs=1e-6 there).
        L = s.shape[1]
        rtol32 = 5e-4 if L > 256 else 1e-5
        assert np.allclose(loss, l32, rtol=rtol32, atol=2e-6), (kind, _rel_err(loss, l32))
        if case["tie_free"] or kind in RANK_FREE:
            assert _grad_err(grad, G.get(name, kind + "/grad64")) < 1e-6, kind
            assert _grad_err(grad, G.get(name, kind + "/grad32")) < 1e-5, kind
        if kind == "hinge":
            # hinge gradients are exact integers: bit-exact
            assert np.array_equal(grad, G.get(name, "hinge/grad32").reshape(grad.shape))


@pytest.mark.parametrize("name", [c["name"] for c in G.by_op("loss") if "literal" in c])
def test_losses_match_reference_unit_test_literals(name):
    case = G.cases[name]
    s, y, n = G.inputs(name)
    for kind, expected in case["literal"].items():
        loss, _ = O.pairwise_loss(kind, s, y, n, sigma=case["sigma"], need_grad=False)
        assert loss == pytest.approx(np.asarray(expected), rel=1e-6, abs=1e-9), kind


@pytest.mark.parametrize("name", METRIC_CASES)
def test_metrics_match_reference_vectors(name):
    case = G.cases[name]
    s, y, n = G.inputs(name)
    L = s.shape[1]
    rtol = 2e-5 if L > 256 else 2e-6
    nn = np.clip(n, 0, L)
    ranking = O.rank_by_score(s, n)
    ref_rank = G.get(name, "ranking")
    for b in range(s.shape[0]):
        if case["tie_free"]:
            assert np.array_equal(ranking[b, :nn[b]], ref_rank[b, :nn[b]])     # bit-exact
        assert sorted(ranking[b, nn[b]:].tolist()) == list(range(nn[b], L))    # tail: a permutation
    assert np.allclose(O.arp(s, y, n), G.get(name, "arp"), rtol=rtol, atol=1e-7)
    for k in case["ks"]:
        for exp in (True, False):
            tag = "k%s_%s" % ("all" if k is None else k, "exp" if exp else "lin")
            assert np.allclose(O.dcg(s, y, n, k=k, exp=exp), G.get(name, "dcg_" + tag),
                               rtol=rtol, atol=1e-7), tag
            assert np.allclose(O.ndcg(s, y, n, k=k, exp=exp), G.get(name, "ndcg_" + tag),
                               rtol=rtol, atol=1e-7), tag
    for key, expected in case.get("literal", {}).items():
        if key == "arp":
            got = O.arp(s, y, n)
        else:
            fn, ktag, etag = key.split("_")
            got = O.dcg(s, y, n, k=int(ktag[1:]), exp=(etag == "exp"), normalize=(fn == "ndcg"))
        assert got == pytest.approx(np.asarray(expected), rel=1e-6, abs=1e-9), key


def test_helpers_match_reference_vectors():
    s = G.get("helpers", "scores")
    y = G.get("helpers", "relevance")
    n = G.get("helpers", "n")
    assert np.array_equal(O.mask_padded_values(s, n).astype(np.float32), G.get("helpers", "mask_default"))
    assert np.array_equal(O.mask_padded_values(s, n, 0.0).astype(np.float32), G.get("helpers", "mask_zero"))
    assert np.array_equal(O.batch_pairs(s).astype(np.float32), G.get("helpers", "pairs_scores"))
    assert np.array_equal(O.batch_pairs(y).astype(np.int64), G.get("helpers", "pairs_relevance"))


@pytest.mark.parametrize("name", [c["name"] for c in G.by_op("linear_step")])
def test_linear_scorer_step_matches_reference_vectors(name):
    case = G.cases[name]
    X, W, b = G.get(name, "X"), G.get(name, "W"), G.get(name, "b")
    y, n = G.get(name, "relevance"), G.get(name, "n")
    B = X.shape[0]
    gout = np.full(B, 1.0 / B)                  # d mean / d loss[b]
    for kind in case["kinds"]:
        loss, scores, dW, db = O.linear_pairwise(kind, X, W, float(b[0]), y, n, gout,
                                                 sigma=case["sigma"])
        assert np.allclose(scores, G.get(name, kind + "/scores"), rtol=1e-5, atol=1e-6)
        assert np.allclose(loss, G.get(name, kind + "/loss"), rtol=2e-5, atol=2e-6), kind
        ref_dW = G.get(name, kind + "/dW")
        tol = 2e-5 * max(1.0, float(np.max(np.abs(ref_dW))))
        assert np.max(np.abs(dW - ref_dW)) < tol, kind
        assert abs(db - float(G.get(name, kind + "/db")[0])) < tol, kind


def test_example3_known_answers():
    """BASELINE.json configs[0] / SURVEY.md 8(c)(2): hinge loss [4.6077, 1.0157],
    dmean/dW = [-2.5, 0, 0, 0.1667, -1.0], db = 0, test nDCG@10 = 0.8617."""
    name = "c1_example3_step"
    loss = G.get(name, "hinge/loss")
    assert loss == pytest.approx([4.6077, 1.0157], abs=2e-4)
    assert G.get(name, "hinge/dW") == pytest.approx([-2.5, 0.0, 0.0, 0.16667, -1.0], abs=1e-4)
    assert G.get(name, "hinge/db") == pytest.approx([0.0], abs=1e-6)
    s, y, n = G.inputs("c1_example3_eval")
    assert O.ndcg(s, y, n, k=10) == pytest.approx([0.8617], abs=1e-4)


# ----------------------------------------------------------------------------
# anchor 3: pair-by-pair Python re-derivation + finite differences (small cases)
# ----------------------------------------------------------------------------
def _python_loss(kind, sigma, s, y, n):
    """Direct transcription of the loss *definitions* (docstring formulas as corrected by
    the reference's tests: discounts log2(2+i), delta on log2(2+|i-j|))."""
    L = len(s)
    nb = max(0, min(int(n), L))
    if kind in ("hinge", "dcg_hinge", "logistic"):
        tot = 0.0
        for i in range(nb):
            for j in range(nb):
                if y[i] > y[j]:
                    d = s[i] - s[j]
                    tot += math.log2(1.0 + math.exp(-sigma * d)) if kind == "logistic" else max(0.0, 1.0 - d)
        return -1.0 / math.log(2.0 + tot) if kind == "dcg_hinge" else tot
    order = sorted(range(L), key=lambda j: (-(s[j] if j < nb else -math.inf), j))
    ideal = sorted((y[j] for j in range(nb)), reverse=True)
    maxdcg = sum((2.0 ** g - 1.0) / math.log2(2.0 + r) for r, g in enumerate(ideal)) or 1.0
    tot = 0.0
    for i in range(nb):
        for j in range(nb):
            a, b = order[i], order[j]
            softplus = math.log2(1.0 + math.exp(-sigma * (s[a] - s[b])))
            if kind == "arp1":
                w = y[a]
            elif kind == "arp2":
                w = (y[a] - y[b]) if y[a] > y[b] else 0.0
            elif kind == "ndcg1":
                w = ((2.0 ** y[a] - 1.0) / maxdcg) / math.log2(2.0 + i)
            else:
                if y[a] > y[b]:
                    dlt = abs(1.0 / math.log2(2.0 + abs(i - j)) - 1.0 / math.log2(3.0 + abs(i - j)))
                    w = dlt * abs((2.0 ** y[a] - 2.0 ** y[b]) / maxdcg)
                else:
                    w = 0.0
            tot += w * softplus
    return tot


@pytest.mark.parametrize("kind", list(O.KINDS))
@pytest.mark.parametrize("name", ["syn_b8_l16", "edge_n_rows", "ut_doc_hinge", "syn_b16_l37_sigma2"])
def test_oracle_vs_python_rederivation_and_finite_differences(kind, name):
    case = G.cases[name]
    s, y, n = G.inputs(name)
    sigma = case["sigma"]
    loss, grad = O.pairwise_loss(kind, s, y, n, sigma=sigma)
    s64 = s.astype(np.float64)
    for b in range(min(s.shape[0], 6)):
        row = s64[b].tolist()
        lab = [float(v) for v in y[b]]
        assert loss[b] == pytest.approx(_python_loss(kind, sigma, row, lab, n[b]), rel=1e-9, abs=1e-12)
        if kind in ("hinge", "dcg_hinge"):
            continue                               # kinks: finite differences not meaningful
        h = 1e-6
        for j in range(len(row)):
            up = list(row); up[j] += h
            dn = list(row); dn[j] -= h
            fd = (_python_loss(kind, sigma, up, lab, n[b]) - _python_loss(kind, sigma, dn, lab, n[b])) / (2 * h)
            assert grad[b, j] == pytest.approx(fd, rel=2e-5, abs=2e-6)


def test_oracle_rejects_bad_kind():
    lib = O._load()
    assert lib.oracle_pairwise_loss(99, 0, None, None, None, 0, 0, None, None) == -1


# ---- MLP scorer + loss step ---------------------------------------------------

_MLP_KINDS = ("hinge", "dcg_hinge", "logistic", "arp1", "arp2", "ndcg1", "ndcg2")


def _mlp_golden():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mlp_vectors.npz"))


@pytest.mark.parametrize("kind", _MLP_KINDS)
def test_mlp_oracle_matches_reference_autograd(kind):
    """oracle_mlp_pairwise vs the real reference: torch layers + the reference's loss modules +
    .mean().backward(), float64 (tests/golden/generate_mlp_golden.py)."""
    from oracle import ltr_oracle as O
    z = _mlp_golden()
    tag = "f64_small"
    X, y, n = z[tag + "/X"], z[tag + "/y"], z[tag + "/n"]
    names = ("l1.weight", "l1.bias", "l2.weight", "l2.bias", "l3.weight", "l3.bias")
    params = [z["%s/param/%s" % (tag, k)] for k in names]
    B = X.shape[0]
    loss, scores, grads = O.mlp_pairwise(kind, X, params, y, n, np.full(B, 1.0 / B))
    # the NDCG losses keep fp32 discount tables in the reference even for fp64 inputs
    rel = 3e-7 if kind in ("ndcg1", "ndcg2") else 1e-12
    assert np.allclose(scores, z["%s/%s/scores" % (tag, kind)], rtol=1e-13, atol=1e-13)
    assert np.allclose(loss, z["%s/%s/loss" % (tag, kind)], rtol=rel, atol=1e-13)
    scale = max(np.abs(z["%s/%s/grad/%s" % (tag, kind, k)]).max() for k in names)
    for key, name in zip(("W1", "b1", "W2", "b2", "W3", "b3"), names):
        want = z["%s/%s/grad/%s" % (tag, kind, name)]
        assert np.abs(grads[key].reshape(want.shape) - want).max() <= rel * scale + 1e-15, (kind, name)


@pytest.mark.parametrize("kind", _MLP_KINDS)
def test_mlp_oracle_matches_reference_fp32_guide_network(kind):
    """The guide's 136-50-10-1 network run by the reference in float32: the fp64 oracle agrees
    to fp32 round-off."""
    from oracle import ltr_oracle as O
    z = _mlp_golden()
    tag = "f32_guide"
    X, y, n = z[tag + "/X"], z[tag + "/y"], z[tag + "/n"]
    names = ("l1.weight", "l1.bias", "l2.weight", "l2.bias", "l3.weight", "l3.bias")
    params = [z["%s/param/%s" % (tag, k)] for k in names]
    B = X.shape[0]
    loss, scores, grads = O.mlp_pairwise(kind, X, params, y, n, np.full(B, 1.0 / B))
    assert np.allclose(scores, z["%s/%s/scores" % (tag, kind)], rtol=1e-5, atol=2e-6)
    assert np.allclose(loss, z["%s/%s/loss" % (tag, kind)], rtol=2e-5, atol=1e-5)
    scale = max(np.abs(z["%s/%s/grad/%s" % (tag, kind, k)]).max() for k in names)
    for key, name in zip(("W1", "b1", "W2", "b2", "W3", "b3"), names):
        want = z["%s/%s/grad/%s" % (tag, kind, name)]
        assert np.abs(grads[key].reshape(want.shape) - want).max() <= 3e-5 * scale + 1e-6, (kind, name)
