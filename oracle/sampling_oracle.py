"""numpy restatement of the Plackett-Luce sampler and the PBM click model, with the random
draw injected as an explicit uniform(0,1) array -- TEST INFRASTRUCTURE ONLY.

rank_by_plackettluce (utils/tensor_operations.py:67-91): r = log(-log u) - log_softmax(masked
scores) (:87-90), ascending argsort (:91); ties by index, padded documents last.
simulate_pbm (click_simulation/pbm.py:12-63): obs = 1/(1+rank)^eta with rank = 1..L, zero at
ranks >= min(n, cutoff) (:34-43); click probability = relevance_probs[label at rank] * obs
(:46-54); results gathered back to document order (:57-63).
Pinned by tests/test_sampling.py against tests/golden/sampling_vectors.npz (from the reference).
"""
import numpy as np


def plackettluce_ranking(scores, n, u):
    s = np.asarray(scores, dtype=np.float64)
    u = np.asarray(u, dtype=np.float64)
    B, L = s.shape
    out = np.zeros((B, L), dtype=np.int64)
    for b in range(B):
        nb = int(np.clip(n[b], 0, L))
        r = np.full(L, np.inf)
        if nb > 0:
            m = s[b, :nb].max()
            logp = s[b, :nb] - (m + np.log(np.exp(s[b, :nb] - m).sum()))
            with np.errstate(divide="ignore"):
                r[:nb] = np.log(-np.log(u[b, :nb])) - logp
        out[b] = np.lexsort((np.arange(L), r))        # ascending r, then index
    return out


def pbm(rankings, ys, n, relevance_probs, cutoff=None, eta=1.0):
    """Returns (click probability per RANK (B,L), propensity per DOCUMENT (B,L))."""
    rk = np.asarray(rankings, dtype=np.int64)
    ys = np.asarray(ys, dtype=np.int64)
    probs = np.asarray(relevance_probs, dtype=np.float64)
    B, L = rk.shape
    p_rank = np.zeros((B, L))
    props = np.zeros((B, L))
    for b in range(B):
        lim = int(n[b]) if cutoff is None else min(int(cutoff), int(n[b]))
        for r in range(L):
            obs = (1.0 / (2.0 + r)) ** eta if r < lim else 0.0
            doc = rk[b, r]
            p_rank[b, r] = probs[ys[b, doc]] * obs
            props[b, doc] = obs
    return p_rank, props


def pbm_clicks(rankings, ys, n, relevance_probs, u, cutoff=None, eta=1.0):
    p_rank, props = pbm(rankings, ys, n, relevance_probs, cutoff, eta)
    rk = np.asarray(rankings, dtype=np.int64)
    clicks = np.zeros(rk.shape, dtype=np.int64)
    hit = np.asarray(u, dtype=np.float64) < p_rank
    for b in range(rk.shape[0]):
        clicks[b, rk[b]] = hit[b]
    return clicks, props
