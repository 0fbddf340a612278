"""CPU baseline port -- TEST/BENCH INFRASTRUCTURE ONLY, never imported by ``pytorchltr_amd``.

The reference cannot travel to the GPU box, so ``bench.py``'s ``cpu_baseline`` leg times this
module on the box's host cores instead.  It is a *materialising* restatement of the same
op-sequence class as the reference's CPU path (SURVEY.md section 3.1 / BASELINE.md section 3):
expand scores and labels to (B, L, L, 2) pair tensors, evaluate the per-pair term with stock
ATen ops, zero pairs touching padded documents through (B, L, L) index grids, reduce, and let
autograd replay the graph backwards.  Memory and op counts therefore match the reference's
(two materialised broadcasts + one stack per pair tensor, two int64 mask grids), which is what
makes it a fair stand-in for "the reference's own CPU path".

It is written from the behaviour documented in SURVEY.md section 8(a) (reference:
utils/tensor_operations.py:94-119, loss/pairwise_additive.py:51-90,
loss/pairwise_lambda.py:50-241); tests/test_cpu_port.py checks it against the C oracle and
the golden vectors.  Ties: stable index order (torch.sort(stable=True)).
"""
import math

import torch

KINDS = ("hinge", "dcg_hinge", "logistic", "arp1", "arp2", "ndcg1", "ndcg2")


def materialize_pairs(x):
    """(B, L) -> (B, L, L, 2) with [..., 0] = x_i and [..., 1] = x_j, fully materialised."""
    B, L = x.shape
    left = x.reshape(B, L, 1).expand(B, L, L).contiguous()      # B*L^2 elements
    right = x.reshape(B, 1, L).expand(B, L, L).contiguous()     # B*L^2 elements
    return torch.stack((left, right), dim=3)                    # copies 2*B*L^2 elements


def _zero_padded_pairs(pair_values, n):
    """Zero every (i, j) with max(i, j) >= n[b], via materialised int64 grids."""
    B, L, _ = pair_values.shape
    idx = torch.arange(L, device=pair_values.device)
    upper = torch.maximum(idx.reshape(L, 1), idx.reshape(1, L))
    upper = upper.reshape(1, L, L).repeat(B, 1, 1)
    limit = n.reshape(B, 1, 1).repeat(1, L, L)
    pair_values[limit <= upper] = 0.0
    return pair_values


def _stable_rank(keys, n):
    """argsort descending with padded documents (index >= n) last; index tie-break."""
    B, L = keys.shape
    idx = torch.arange(L, device=keys.device).reshape(1, L)
    masked = keys.clone()
    masked[idx >= n.reshape(B, 1)] = -math.inf
    return torch.sort(masked, dim=1, descending=True, stable=True).indices


def _max_dcg(sorted_rel, n):
    B, L = sorted_rel.shape
    order = _stable_rank(sorted_rel.double(), n)
    pos = torch.arange(L, device=sorted_rel.device)
    ideal = torch.gather(sorted_rel, 1, order)
    ideal[n.reshape(B, 1) <= pos.reshape(1, L)] = 0.0
    ideal = (2 ** ideal) - 1.0
    return torch.sum(ideal / torch.log2(2.0 + pos).reshape(1, L), dim=1)


def pairwise_loss(kind, scores, relevance, n, sigma=1.0):
    """Per-query loss (B,), differentiable w.r.t. `scores` through autograd."""
    B, L = scores.shape[0], scores.shape[1]
    s = scores.reshape(B, L)
    y = relevance.reshape(B, L)
    lambda_kind = kind in ("arp1", "arp2", "ndcg1", "ndcg2")
    if lambda_kind:
        order = _stable_rank(s.detach(), n)
        s = torch.gather(s, 1, order)
        y = torch.gather(y, 1, order)
    sp = materialize_pairs(s)
    yp = materialize_pairs(y)
    ds = sp[..., 0] - sp[..., 1]
    dy = yp[..., 0] - yp[..., 1]

    if kind in ("hinge", "dcg_hinge"):
        val = 1.0 - ds
        val[dy <= 0] = 0.0
        val[val < 0.0] = 0.0
    elif kind == "logistic":
        val = torch.log2(1.0 + torch.exp(-sigma * ds))
        val[dy <= 0] = 0.0
    elif kind == "arp1":
        prob = 1.0 / (1.0 + torch.exp(-sigma * ds))
        val = -torch.log2(prob ** yp[..., 0])
    elif kind == "arp2":
        val = torch.log2(1.0 + torch.exp(-sigma * ds))
        val[dy <= 0] = 0.0
        val = dy * val
    else:
        max_dcg = _max_dcg(y, n)
        max_dcg[max_dcg == 0.0] = 1.0
        gains = ((2 ** yp) - 1.0) / max_dcg.reshape(B, 1, 1, 1)
        prob = 1.0 / (1.0 + torch.exp(-sigma * ds))
        if kind == "ndcg1":
            disc = torch.log2(2.0 + torch.arange(L, device=s.device))
            val = -torch.log2(prob ** (gains[..., 0] / disc.reshape(1, L, 1)))
        else:
            pos = torch.arange(L + 1, device=s.device)
            disc = torch.log2(2.0 + pos)
            gap = (pos[:-1].reshape(L, 1) - pos[:-1].reshape(1, L)).abs()
            delta = (1.0 / disc[gap] - 1.0 / disc[gap + 1]).abs()
            weight = delta.reshape(1, L, L) * (gains[..., 0] - gains[..., 1]).abs()
            val = torch.log2(prob ** weight)
            val[dy <= 0] = 0.0
            val = -val

    val = _zero_padded_pairs(val, n)
    total = val.reshape(B, L * L).sum(dim=1)
    if kind == "dcg_hinge":
        total = -1.0 / torch.log(2.0 + total)
    return total


def loss_step(kind, scores, relevance, n, sigma=1.0):
    """`loss_fn(scores, relevance, n).mean().backward()`; returns (loss[B], dmean/dscores)."""
    s = scores.detach().clone().requires_grad_(True)
    loss = pairwise_loss(kind, s, relevance, n, sigma)
    loss.mean().backward()
    return loss.detach(), s.grad


def linear_step(kind, X, weight, bias, relevance, n, sigma=1.0):
    """`loss_fn(Linear(F,1)(X), relevance, n).mean().backward()`; returns (loss, dW, db)."""
    w = weight.detach().clone().reshape(1, -1).requires_grad_(True)
    b = bias.detach().clone().reshape(1).requires_grad_(True)
    scores = torch.nn.functional.linear(X, w, b)
    loss = pairwise_loss(kind, scores, relevance, n, sigma)
    loss.mean().backward()
    return loss.detach(), w.grad.reshape(-1), b.grad


def dcg(scores, relevance, n, k=None, exp=True):
    B, L = scores.shape[0], scores.shape[1]
    order = _stable_rank(scores.reshape(B, L), n)
    rel = torch.gather(relevance.reshape(B, L), 1, order).float()
    if exp:
        rel = 2.0 ** rel - 1.0
    pos = torch.arange(L, dtype=torch.float, device=scores.device).reshape(1, L)
    curve = torch.cumsum(rel / torch.log2(pos + 2.0), dim=1)
    if k is not None:
        curve = curve[:, :k][:, -1]
    return curve


def ndcg(scores, relevance, n, k=None, exp=True):
    ideal = dcg(relevance.float(), relevance, n, k, exp)
    ideal[ideal == 0.0] = 1.0
    return dcg(scores, relevance, n, k, exp) / ideal


def arp(scores, relevance, n):
    B, L = scores.shape[0], scores.shape[1]
    order = _stable_rank(scores.reshape(B, L), n)
    rel = torch.gather(relevance.reshape(B, L), 1, order).float()
    pos = torch.arange(L, device=scores.device).reshape(1, L)
    rel[pos >= n.reshape(B, 1)] = 0.0
    num = torch.sum((pos + 1.0) * rel, dim=1)
    den = torch.sum(rel, dim=1)
    den[den == 0.0] = 1.0
    return num / den
