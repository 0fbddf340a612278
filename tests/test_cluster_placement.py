"""CPU tier: the arithmetic of the cluster kernel's placement check (csrc/ltr_cluster.inc: cluster_wait_placed).

A query's workgroups are laid out on block ids congruent mod 8 so that they share one XCD and may hand each other plain
stores through its L2.  Every launch checks that: the arrival on the first counter adds  1 | x << 8 | x * x << 16  (x = the
member's HW_REG_XCC_ID) to ONE 32-bit word, and a member concludes "all P arrivals came from my XCD" exactly when the
count is P, the sum P * x and the sum of squares P * x * x.  This file pins the two facts the device code rests on: the
fields cannot overflow into each other for P <= 16, and the test has no false positive (zero variance <=> all equal)."""
import itertools
import random


def _word(xs):
    w = 0
    for x in xs:
        w += 1 | (x << 8) | ((x * x) << 16)
    return w & 0xFFFFFFFF


def _local(word, want, x):
    return (word & 0xFF) >= want and ((word >> 8) & 0xFF) == want * x and (word >> 16) == want * x * x


def test_fields_do_not_overflow_for_sixteen_members():
    worst = _word([7] * 16)
    assert worst & 0xFF == 16 and (worst >> 8) & 0xFF == 7 * 16 and worst >> 16 == 49 * 16
    assert 7 * 16 < 256 and 49 * 16 < 65536


def test_no_false_positive_small_clusters_exhaustive():
    for P in range(1, 6):
        for xs in itertools.product(range(8), repeat=P):
            w = _word(xs)
            for x in set(xs):
                assert _local(w, P, x) == (len(set(xs)) == 1), (xs, x)


def test_no_false_positive_random_large_clusters():
    rnd = random.Random(5)
    for _ in range(20000):
        P = rnd.randint(2, 16)
        xs = [rnd.randrange(8) for _ in range(P)]
        if rnd.random() < 0.3:                       # near misses: one member elsewhere
            xs = [xs[0]] * P
            xs[rnd.randrange(P)] = rnd.randrange(8)
        w = _word(xs)
        for x in set(xs):
            assert _local(w, P, x) == (len(set(xs)) == 1), (xs, x)


def test_block_id_layout_is_a_bijection_with_one_residue_per_query():
    """block id -> (query position, part): members of a query on block ids congruent mod 8, queries dealt boustrophedon."""
    for B, Pmax in ((32, 14), (5, 10), (13, 3), (256, 10)):
        grid = (B + 7) // 8 * 8 * Pmax
        seen = {}
        for blk in range(grid):
            xslot, xidx = blk & 7, blk >> 3
            ql, part = divmod(xidx, Pmax)
            qpos = ql * 8 + ((7 - xslot) if (ql & 1) else xslot)
            if qpos >= B:
                continue
            seen.setdefault(qpos, []).append((blk, part))
        assert sorted(seen) == list(range(B))
        for qpos, members in seen.items():
            assert sorted(p for _, p in members) == list(range(Pmax))
            assert len({blk % 8 for blk, _ in members}) == 1
            blks = sorted(blk for blk, _ in members)
            assert blks == list(range(blks[0], blks[0] + 8 * Pmax, 8))      # consecutive in their XCD's dispatch order
