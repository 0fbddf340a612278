"""CPU tier: the algorithm of the cluster kernel's sorted runs (csrc/ltr_cluster.inc: cluster_local_ranks,
cluster_hinge_from_runs) restated in numpy, step for step, against the oracle's pair pass.

The device path replaces the hinge kinds' O(n^2) pair pass (loss/pairwise_additive.py:107-113) on integer grades 0..4 by:
local ranks inside runs of `rpw` rows, per-run inclusive prefix counts per grade threshold, and for every document two
fixed-step boundary searches per run with the pair pass's own fp32 predicate fl(1 - fl(s_hi - s_lo)) >= 0.  The claim
tested here: the per-document gradient counts are the pair pass's integers (ties, scores exactly at the margin, runs of
uneven length, empty grades), and #active + sum (s - c) g is the pair sum."""
import numpy as np
import pytest

from oracle import ltr_oracle as O

f32 = np.float32


def _rank_key(s, idx):
    """csrc/ltr_common.inc rank_key: ascending keys = score descending, index ascending (-0.0 folded into +0.0)."""
    bits = int(np.array([s + f32(0.0)], dtype=np.float32).view(np.uint32)[0])
    asc = (~bits & 0xFFFFFFFF) if (bits & 0x80000000) else (bits | 0x80000000)
    return ((~asc & 0xFFFFFFFF) << 32) | idx


def _by_runs(s, y, rpw):
    n = len(s)
    P = (n + rpw - 1) // rpw
    ss = np.zeros(n, dtype=np.float32)
    below_g = np.zeros((n, 4), dtype=np.int64)           # grade word per slot: [y < 1, y < 2, y < 3, y < 4]
    for m in range(P):
        base, ln = m * rpw, min(rpw, n - m * rpw)
        keys = [_rank_key(s[base + i], i) for i in range(ln)]
        for i in range(ln):                               # local rank = keys of my run below mine
            rk = sum(1 for k in keys if k < keys[i])
            ss[base + rk] = s[base + i]
            below_g[base + rk] = [y[base + i] < g for g in (1, 2, 3, 4)]
    tab = np.zeros((n, 4), dtype=np.int64)
    for m in range(P):                                    # inclusive prefix inside the run
        base, ln = m * rpw, min(rpw, n - m * rpw)
        tab[base:base + ln] = np.cumsum(below_g[base:base + ln], axis=0)
    top = 1 << (max(rpw, 1).bit_length() - 1)
    g = np.zeros(n, dtype=np.int64)
    active = 0
    for i in range(n):
        si, yi = f32(s[i]), int(y[i])
        below = above = 0
        for m in range(P):
            base, ln = m * rpw, min(rpw, n - m * rpw)
            for side in (0, 1):
                if (side == 0 and yi == 0) or (side == 1 and yi == 4):
                    continue
                sgn = f32(1.0) if side else f32(-1.0)
                pos, step = 0, top
                while step >= 1:                           # fixed-step search: leading slots with active == (side == 0)
                    cand = pos + step
                    o = ss[base + min(cand, ln) - 1]
                    uu = f32(sgn * f32(si - o)) + f32(1.0)         # fmaf(sgn, si - o, 1): sgn = +-1, the product is exact
                    take = cand <= ln and ((uu >= 0) != bool(side))
                    pos = cand if take else pos
                    step >>= 1
                sh = yi if side else yi - 1                # counter of "grade <= y" / "grade < y"
                tp = tab[base + pos - 1][sh] if pos > 0 else 0
                tl = tab[base + ln - 1][sh]
                if side == 0:
                    below += tp
                else:
                    above += (ln - pos) - (tl - tp)
        g[i] = above - below
        active += below
    c = f32(s[0])
    raw = float(active) + float(np.sum((s.astype(np.float32) - c).astype(np.float64) * g))
    return g, raw


@pytest.mark.parametrize("n,rpw,seed", [(1, 8, 0), (7, 8, 1), (64, 16, 2), (100, 24, 3), (257, 72, 4), (300, 120, 5)])
def test_sorted_runs_give_the_pair_pass_gradients(n, rpw, seed):
    rng = np.random.default_rng(seed)
    s = rng.normal(0, 1.2, n).astype(np.float32)
    y = rng.integers(0, 5, n)
    if n >= 7:
        s[3] = s[1]                                        # a tie
        s[5] = f32(s[2] - f32(1.0))                        # exactly at the margin against document 2
        s[6] = np.nextafter(f32(s[2] - f32(1.0)), f32(-np.inf))
        y[2], y[5], y[6] = 4, 0, 1
    if seed == 3:
        y[:] = np.where(y == 2, 3, y)                     # an empty grade
    g, raw = _by_runs(s, y, rpw)
    # the oracle on the SAME fp32 scores, fp32 arithmetic of the predicate
    sc, rel, nn = s.reshape(1, -1).astype(np.float64), y.reshape(1, -1), np.array([n])
    want_l, want_g = O.pairwise_loss("hinge", sc, rel, nn)
    # pair pass in fp32, by brute force (the predicate the kernels evaluate)
    gb = np.zeros(n, dtype=np.int64)
    lb = 0.0
    for i in range(n):
        for j in range(n):
            if y[i] > y[j]:
                u = f32(1.0) - f32(f32(s[i]) - f32(s[j]))
                if u >= 0:
                    gb[i] -= 1
                    gb[j] += 1
                    lb += float(u)
    assert np.array_equal(g, gb)
    assert raw == pytest.approx(lb, rel=2e-6, abs=1e-4)
    # and the fp64 oracle agrees wherever no pair sits within fp32 rounding of the margin
    assert want_l[0] == pytest.approx(lb, rel=1e-5, abs=1e-4)
