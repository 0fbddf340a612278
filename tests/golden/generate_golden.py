#!/usr/bin/env python
"""Generate golden vectors by importing the REAL reference (rjagerman/pytorchltr).

Run in the build container only (the reference is mounted read-only at
/root/reference and does not exist on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/generate_golden.py

Writes tests/golden/reference_vectors.npz (+ reference_vectors.json manifest).
The .npz holds DATA only: inputs and the reference's outputs (loss, autograd
gradients, rankings, metrics).  No reference source or bytecode is copied.

Case families
  ut_*    literal inputs of the reference's own unit tests / doc examples
          (tests/loss/test_pairwise_additive.py, tests/loss/test_pairwise_lambda.py,
          tests/evaluation/test_dcg.py, tests/evaluation/test_arp.py,
          docs/source/loss.rst:30-37, docs/source/evaluation.rst:17-23), with the
          closed-form expectation the reference test asserts stored as `literal`.
  c1_*    Example3 toy data (BASELINE.json configs[0]): Linear(5,1) + hinge step.
  syn_*   synthetic recipe of SURVEY.md section 8(d) at several shapes.
  edge_*  n=0 / n=1 / n>L rows, all-zero labels, (B,L,1) inputs, sigma=2, float labels.
"""
import json
import os
import sys

import numpy as np
import torch

REFERENCE = os.environ.get("PYTORCHLTR_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, REFERENCE)

from pytorchltr.loss import (  # noqa: E402
    PairwiseHingeLoss, PairwiseDCGHingeLoss, PairwiseLogisticLoss,
    LambdaARPLoss1, LambdaARPLoss2, LambdaNDCGLoss1, LambdaNDCGLoss2)
from pytorchltr.evaluation import arp, dcg, ndcg  # noqa: E402
from pytorchltr.utils import (  # noqa: E402
    batch_pairs, mask_padded_values, rank_by_score)

HERE = os.path.dirname(os.path.abspath(__file__))

LOSSES = {
    "hinge": lambda sigma: PairwiseHingeLoss(),
    "dcg_hinge": lambda sigma: PairwiseDCGHingeLoss(),
    "logistic": lambda sigma: PairwiseLogisticLoss(sigma),
    "arp1": lambda sigma: LambdaARPLoss1(sigma),
    "arp2": lambda sigma: LambdaARPLoss2(sigma),
    "ndcg1": lambda sigma: LambdaNDCGLoss1(sigma),
    "ndcg2": lambda sigma: LambdaNDCGLoss2(sigma),
}
ALL_KINDS = list(LOSSES)

arrays = {}
manifest = []


def synth(B, L, seed, F=None):
    """SURVEY.md section 8(d) recipe."""
    g = torch.Generator().manual_seed(seed)
    scores = torch.randn(B, L, generator=g)
    relevance = torch.randint(0, 5, (B, L), generator=g)
    n = torch.randint(1, L + 1, (B,), generator=g)
    out = [scores, relevance, n]
    if F is not None:
        X = torch.randn(B, L, F, generator=g)
        bound = 1.0 / (F ** 0.5)
        W = (torch.rand(F, generator=g) * 2 - 1) * bound
        b = (torch.rand(1, generator=g) * 2 - 1) * bound
        out += [X, W, b]
    return out


def tie_free(scores, n):
    s = scores.reshape(scores.shape[0], scores.shape[1]).double().numpy()
    nn = np.clip(np.asarray(n.numpy(), dtype=np.int64), 0, s.shape[1])
    return bool(all(len(np.unique(s[b, :nn[b]])) == nn[b] for b in range(s.shape[0])))


def put(name, field, value):
    if torch.is_tensor(value):
        value = value.detach().cpu().numpy()
    arrays["%s/%s" % (name, field)] = np.asarray(value)


def run_loss(kind, sigma, scores, relevance, n, dtype):
    torch.manual_seed(0)  # the Lambda losses draw a randperm from the global RNG
    s = scores.detach().clone().to(dtype).requires_grad_(True)
    loss = LOSSES[kind](sigma)(s, relevance, n)
    loss.sum().backward()
    return loss.detach(), s.grad.detach()


def add_loss_case(name, scores, relevance, n, kinds=ALL_KINDS, sigma=1.0, literal=None,
                  note=""):
    put(name, "scores", scores.float())
    put(name, "relevance", relevance)
    put(name, "n", n)
    entry = {"name": name, "op": "loss", "kinds": list(kinds), "sigma": sigma,
             "tie_free": tie_free(scores, n), "note": note}
    for kind in kinds:
        l32, g32 = run_loss(kind, sigma, scores, relevance, n, torch.float32)
        l64, g64 = run_loss(kind, sigma, scores, relevance, n, torch.float64)
        put(name, kind + "/loss32", l32)
        put(name, kind + "/grad32", g32)
        put(name, kind + "/loss64", l64)
        put(name, kind + "/grad64", g64)
    if literal is not None:
        entry["literal"] = literal
    manifest.append(entry)


def add_metric_case(name, scores, relevance, n, ks=(None, 1, 3, 5, 10), literal=None,
                    note="", zero_pad=True):
    if zero_pad:
        # collate zero-pads labels (datasets/svmrank/svmrank.py:149-150); with non-zero
        # padded labels the reference's dcg depends on its random tail order.
        L = relevance.shape[1]
        relevance = relevance * (torch.arange(L)[None, :] < n.reshape(-1, 1).clamp(max=L))
    put(name, "scores", scores.float())
    put(name, "relevance", relevance)
    put(name, "n", n)
    entry = {"name": name, "op": "metrics", "ks": [k for k in ks],
             "tie_free": tie_free(scores, n), "note": note}
    torch.manual_seed(0)
    put(name, "ranking", rank_by_score(scores, n))
    torch.manual_seed(0)
    put(name, "arp", arp(scores, relevance, n))
    for k in ks:
        for exp in (True, False):
            tag = "k%s_%s" % ("all" if k is None else k, "exp" if exp else "lin")
            torch.manual_seed(0)
            put(name, "dcg_" + tag, dcg(scores, relevance, n, k=k, exp=exp))
            torch.manual_seed(0)
            put(name, "ndcg_" + tag, ndcg(scores, relevance, n, k=k, exp=exp))
    if literal is not None:
        entry["literal"] = literal
    manifest.append(entry)


# --------------------------------------------------------------------------
# ut_*: the reference's own unit-test inputs, with the value its test asserts
# --------------------------------------------------------------------------
F32 = torch.FloatTensor
I64 = torch.LongTensor
from math import exp as mexp, log as mlog, log2 as mlog2  # noqa: E402

ys5 = I64([[0, 0, 1, 2, 1]])
n5 = I64([5])
# tests/loss/test_pairwise_additive.py
add_loss_case("ut_hinge_perfect", F32([[0.0, 0.0, 1.0, 2.0, 1.0]]), ys5, n5,
              kinds=["hinge"], literal={"hinge": [0.0]})
add_loss_case("ut_hinge_2", F32([[0.0, 0.0, 1.0, 1.0, 1.0]]), ys5, n5,
              kinds=["hinge"], literal={"hinge": [2.0]})
add_loss_case("ut_hinge_3", F32([[0.0, 0.0, 1.0, -5.0, 1.0]]), ys5, n5,
              kinds=["hinge"], literal={"hinge": [26.0]})
add_loss_case("ut_hinge_batch",
              F32([[0.0, 10.0, 1.0, 0.5, 1.0], [1.0, 3.5, 6.0, 4.3, 10.0]]),
              I64([[0, 2, 1, 2, 1], [1, 2, 2, 1, 0]]), I64([5, 4]),
              kinds=["hinge"], literal={"hinge": [3.5, 5.3 - 3.5]})
cut_scores = F32([[1.0, 3.5, 6.0, 4.3, 8.0]])
cut_ys = I64([[1, 2, 2, 1, 0]])
add_loss_case("ut_hinge_cutoff_n3", cut_scores, cut_ys, I64([3]), kinds=["hinge"],
              literal={"hinge": [0.0]})
add_loss_case("ut_hinge_cutoff_n4", cut_scores, cut_ys, I64([4]), kinds=["hinge"],
              literal={"hinge": [5.3 - 3.5]})
add_loss_case("ut_hinge_cutoff_n5", cut_scores, cut_ys, I64([5]), kinds=["hinge"],
              literal={"hinge": [(5.3 - 3.5) + (9.0 - 4.3) + (9.0 - 6.0) + (9.0 - 3.5) + (9.0 - 1.0)]})
add_loss_case("ut_dcghinge_perfect", F32([[0.0, 0.0, 1.0, 2.0, 1.0]]), ys5, n5,
              kinds=["dcg_hinge"], literal={"dcg_hinge": [-1.0 / mlog(2.0 + 0.0)]})
add_loss_case("ut_dcghinge_worst", F32([[3.0, 3.0, 1.0, 0.0, 1.0]]), ys5, n5,
              kinds=["dcg_hinge"], literal={"dcg_hinge": [-1.0 / mlog(2.0 + 24.0)]})


def _logistic_lit(d1, d2, d3):
    return [mlog2(1.0 + mexp(-d1)) * 2 + mlog2(1.0 + mexp(-d2)) * 2 + mlog2(1.0 + mexp(-d3)) * 4]


add_loss_case("ut_logistic_perfect", F32([[0.0, 0.0, 1.0, 2.0, 1.0]]), ys5, n5,
              kinds=["logistic"], literal={"logistic": _logistic_lit(1.0, 2.0, 1.0)})
add_loss_case("ut_logistic_worst", F32([[3.0, 3.0, 1.0, 0.0, 1.0]]), ys5, n5,
              kinds=["logistic"], literal={"logistic": _logistic_lit(-3.0, -1.0, -2.0)})

# tests/loss/test_pairwise_lambda.py:11-40 (frozen goldens)
add_loss_case("ut_lambda_batch",
              torch.tensor([[0.5, 2.0, 1.0], [0.9, -1.2, 0.0]]),
              torch.tensor([[2, 0, 1], [0, 1, 0]]), torch.tensor([3, 2]),
              kinds=["arp1", "arp2", "ndcg1", "ndcg2"],
              literal={"arp1": [13.298417091369629, 4.196318626403809],
                       "arp2": [8.209173202514648, 3.1963188648223877],
                       "ndcg1": [2.629549503326416, 2.647582530975342],
                       "ndcg2": [0.3102627396583557, 0.4184933304786682]})
# tests/loss/test_pairwise_lambda.py:43-366 perfect / worst / mid orderings.  The
# iterative re-derivations in those tests are re-derived independently in
# tests/test_oracle_golden.py; here we freeze what the reference returns.
add_loss_case("ut_lambda_perfect", F32([[0.0, 0.0, 10.0, 20.0, 10.0]]), ys5, n5,
              kinds=["arp1", "arp2", "ndcg1", "ndcg2"],
              note="score ties between equal-label docs: loss is tie-order invariant, grads are not")
add_loss_case("ut_lambda_worst", F32([[4.0, 4.0, 2.0, 0.0, 2.0]]), ys5, n5,
              kinds=["arp1", "arp2", "ndcg1", "ndcg2"],
              note="score ties between equal-label docs")
add_loss_case("ut_lambda_arp_mid", F32([[0.0, 1.0, 1.0, -2.0, 0.0]]), ys5, I64([4]),
              kinds=["arp1", "arp2"], note="tie at 1.0; ARP losses are rank-independent")
add_loss_case("ut_lambda_ndcg_mid", F32([[0.0, 1.0, 1.5, -2.0, 0.0]]), ys5, I64([4]),
              kinds=["ndcg1", "ndcg2"])
# docs/source/loss.rst:30-37
add_loss_case("ut_doc_hinge", torch.tensor([[0.5, 2.0, 1.0], [0.9, -1.2, 0.0]]),
              torch.tensor([[2, 0, 1], [0, 1, 0]]), torch.tensor([3, 2]),
              kinds=ALL_KINDS, literal={"hinge": [6.0, 3.1]})

# tests/evaluation/test_dcg.py, test_arp.py
ev_scores = F32([[10.0, 5.0, 2.0, 3.0, 4.0], [5.0, 6.0, 4.0, 2.0, 5.5]])
add_metric_case("ut_dcg_data", ev_scores, I64([[0, 1, 1, 0, 1], [3, 1, 0, 1, 0]]), I64([5, 4]),
                literal={"dcg_k3_exp": [1.1309297535714575, 5.4165082750002025],
                         "ndcg_k3_exp": [1.1309297535714575 / 2.1309297535714578,
                                         5.4165082750002025 / 8.130929753571458],
                         "dcg_k5_exp": [1.5177825608059992, 5.847184833073595],
                         "ndcg_k5_exp": [1.5177825608059992 / 2.1309297535714578,
                                         5.847184833073595 / 8.130929753571458],
                         "dcg_k5_lin": [1.5177825608059992, 3.3234658187877653],
                         "ndcg_k5_lin": [1.5177825608059992 / 2.1309297535714578,
                                         3.3234658187877653 / 4.130929753571458]})
add_metric_case("ut_arp_data", ev_scores, I64([[0, 1, 1, 0, 1], [1, 1, 0, 0, 0]]), I64([5, 4]),
                literal={"arp": [3.333333333, 1.5]})
add_metric_case("ut_all_relevant", ev_scores, I64([[1, 1, 1, 1, 1], [1, 1, 1, 1, 1]]), I64([5, 4]),
                literal={"arp": [3.0, 2.5], "ndcg_k5_lin": [1.0, 1.0]}, zero_pad=False,
                note="padded doc of row 1 carries label 1: dcg includes it at rank 4 (reference quirk)")
add_metric_case("ut_no_relevant", ev_scores, I64([[0, 0, 0, 0, 0], [0, 0, 0, 0, 0]]), I64([5, 4]),
                literal={"arp": [0.0, 0.0], "dcg_k5_lin": [0.0, 0.0], "ndcg_k5_lin": [0.0, 0.0]})
# docs/source/evaluation.rst:17-23
add_metric_case("ut_doc_ndcg", torch.tensor([[1.0, 0.0, 1.5], [1.5, 0.2, 0.5]]),
                torch.tensor([[0, 1, 0], [0, 1, 1]]), torch.tensor([3, 3]),
                literal={"ndcg_k10_exp": [0.5, 0.6934264036172708]})

# --------------------------------------------------------------------------
# c1_*: Example3 (BASELINE.json configs[0]); bytes pinned by sha256 at
# pytorchltr/datasets/svmrank/example3.py:29-30, content listed in SURVEY.md 8(c).
# Features are per-query min-max normalised as SVMRankDataset does
# (datasets/svmrank/svmrank.py:107-113).
# --------------------------------------------------------------------------
EX3_TRAIN = {
    1: ([3, 2, 1, 1], [[1, 1, 0, 0.2, 0], [0, 0, 1, 0.1, 1], [0, 1, 0, 0.4, 0], [0, 0, 1, 0.3, 0]]),
    2: ([1, 2, 1, 1], [[0, 0, 1, 0.2, 0], [1, 0, 1, 0.4, 0], [0, 0, 1, 0.1, 0], [0, 0, 1, 0.2, 0]]),
    3: ([2, 3, 4, 1], [[0, 0, 1, 0.1, 1], [1, 1, 0, 0.3, 0], [1, 0, 0, 0.4, 1], [0, 1, 1, 0.5, 0]]),
}
EX3_TEST = {
    4: ([4, 3, 2, 1], [[1, 0, 0, 0.2, 1], [1, 1, 0, 0.3, 0], [0, 0, 0, 0.2, 1], [0, 0, 1, 0.2, 0]]),
}


def _minmax(x):
    x = np.asarray(x, dtype=np.float64)
    lo, hi = x.min(axis=0, keepdims=True), x.max(axis=0, keepdims=True)
    rng = hi - lo
    rng[rng == 0.0] = 1.0
    return (x - lo) / rng


def _ex3_batch(qs, table):
    xs = torch.tensor(np.stack([_minmax(table[q][1]) for q in qs]), dtype=torch.float32)
    ys = torch.tensor([table[q][0] for q in qs], dtype=torch.int64)
    n = torch.tensor([len(table[q][0]) for q in qs], dtype=torch.int64)
    return xs, ys, n


def add_linear_case(name, X, W, b, relevance, n, kinds, sigma=1.0, note=""):
    put(name, "X", X)
    put(name, "W", W)
    put(name, "b", b)
    put(name, "relevance", relevance)
    put(name, "n", n)
    for kind in kinds:
        torch.manual_seed(0)
        model = torch.nn.Linear(X.shape[2], 1)
        with torch.no_grad():
            model.weight.copy_(W.reshape(1, -1))
            model.bias.copy_(b.reshape(1))
        scores = model(X)
        loss = LOSSES[kind](sigma)(scores, relevance, n)
        loss.mean().backward()
        put(name, kind + "/scores", scores.detach().reshape(X.shape[0], X.shape[1]))
        put(name, kind + "/loss", loss.detach())
        put(name, kind + "/dW", model.weight.grad.reshape(-1))
        put(name, kind + "/db", model.bias.grad.reshape(-1))
    manifest.append({"name": name, "op": "linear_step", "kinds": list(kinds), "sigma": sigma,
                     "note": note})


xs, ys, n = _ex3_batch([1, 2], EX3_TRAIN)
W_c1 = torch.tensor([0.3419, 0.3712, -0.1048, 0.4108, -0.0980])   # seed-42 Linear(5,1) init
b_c1 = torch.tensor([0.0902])
add_linear_case("c1_example3_step", xs, W_c1, b_c1, ys, n, kinds=ALL_KINDS,
                note="Example3 train queries 1,2; Linear(5,1) at its seed-42 init (4 decimals)")
xs_t, ys_t, n_t = _ex3_batch([4], EX3_TEST)
with torch.no_grad():
    sc_t = (xs_t @ W_c1.reshape(-1, 1)).reshape(1, -1) + b_c1
add_metric_case("c1_example3_eval", sc_t, ys_t, n_t, ks=(None, 10),
                note="Example3 test query scored by the seed-42 Linear(5,1); ndcg@10 = 0.8617")

# --------------------------------------------------------------------------
# syn_*: SURVEY.md 8(d) recipe
# --------------------------------------------------------------------------
s, y, n = synth(8, 16, 1234)
assert abs(float(s[0, 0]) - (-0.11171857)) < 1e-6 and y[0, :8].tolist() == [2, 1, 1, 3, 1, 3, 1, 1]
add_loss_case("syn_b8_l16", s, y, n, note="known answers of SURVEY.md 8(c)(3)")
add_metric_case("syn_b8_l16_metrics", s, y, n)
s, y, n = synth(64, 128, 0)
add_loss_case("syn_b64_l128", s, y, n, note="C2/C3 shape (MSLR-WEB30K) at B=64")
add_metric_case("syn_b64_l128_metrics", s, y, n, ks=(None, 10))
s, y, n = synth(4, 1000, 0)
add_loss_case("syn_b4_l1000", s, y, n, note="C4 shape (Istella-X) at B=4")
add_metric_case("syn_b4_l1000_metrics", s, y, n, ks=(None, 10, 2000))
s, y, n = synth(4, 512, 0)
add_loss_case("syn_b4_l512", s, y, n, kinds=["hinge", "ndcg2"], note="C5 shape (Yahoo) at B=4")
s, y, n = synth(6, 200, 7)
add_loss_case("syn_b6_l200_full", s, y, torch.full((6,), 200, dtype=torch.int64),
              note="n == L everywhere (worst case for work), non-power-of-two L")
s, y, n = synth(16, 37, 11)
add_loss_case("syn_b16_l37_sigma2", s, y, n, sigma=2.0, note="odd L, sigma=2")
add_metric_case("syn_b16_l37_metrics", s, y, n, ks=(None, 5, 37, 100))
s, y, n, X, W, b = synth(8, 16, 1234, F=5)
add_linear_case("syn_linear_b8_l16_f5", X, W, b, y, n, kinds=ALL_KINDS)
s, y, n, X, W, b = synth(16, 128, 3, F=136)
add_linear_case("syn_linear_b16_l128_f136", X, W, b, y, n, kinds=["hinge", "logistic", "ndcg2"],
                note="C2 shape with scorer at B=16")

# --------------------------------------------------------------------------
# edge_*
# --------------------------------------------------------------------------
s, y, n = synth(6, 12, 21)
n_edge = torch.tensor([0, 1, 2, 12, 13, 40], dtype=torch.int64)
add_loss_case("edge_n_rows", s, y, n_edge, note="n = 0, 1, 2, L, L+1, >>L")
add_metric_case("edge_n_rows_metrics", s, y, n_edge, ks=(None, 3),
                note="padded labels zeroed (as collate does)")
add_loss_case("edge_zero_labels", s, torch.zeros_like(y), n, note="all-zero labels: NDCG losses 0, zero grad")
add_loss_case("edge_3d_scores", s.reshape(6, 12, 1), y, n, note="(B,L,1) scores as nn.Linear emits them")
add_loss_case("edge_float_labels", s, y.float() * 0.5, n, note="float labels are accepted")
add_loss_case("edge_equal_labels_nonzero", s, torch.full_like(y, 3), n,
              note="no y_i > y_j pair exists; ARP1/NDCG1 still sum over all pairs")
s1, y1, n1 = synth(3, 1, 5)
add_loss_case("edge_L1", s1, y1, n1, note="list_len 1")
add_metric_case("edge_L1_metrics", s1, y1, n1, ks=(None, 1))
# non-zero padded labels with k <= min n: dcg@k unaffected by the (random) tail order
s, y, n = synth(8, 20, 33)
n = n.clamp(min=6)
add_metric_case("edge_padded_labels_k5", s, y, n, ks=(5,), zero_pad=False,
                note="padded labels left non-zero; only k <= min(n) is order independent")

# helpers (no direct reference tests exist: pinned here)
s, y, n = synth(4, 9, 2)
put("helpers", "scores", s)
put("helpers", "relevance", y)
put("helpers", "n", n)
put("helpers", "mask_default", mask_padded_values(s, n))
put("helpers", "mask_zero", mask_padded_values(s, n, mask_value=0.0))
put("helpers", "pairs_scores", batch_pairs(s))
put("helpers", "pairs_relevance", batch_pairs(y))
manifest.append({"name": "helpers", "op": "helpers"})

np.savez_compressed(os.path.join(HERE, "reference_vectors.npz"), **arrays)
with open(os.path.join(HERE, "reference_vectors.json"), "w") as fh:
    json.dump({"reference": "rjagerman/pytorchltr v0.2.1 (read-only import)",
               "torch": torch.__version__, "cases": manifest}, fh, indent=1, sort_keys=True)
print("wrote %d arrays, %d cases, %.1f KiB" % (
    len(arrays), len(manifest),
    os.path.getsize(os.path.join(HERE, "reference_vectors.npz")) / 1024.0))
