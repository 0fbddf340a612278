"""numpy restatement of the reference's dense collate -- TEST INFRASTRUCTURE ONLY.

Follows SVMRankDataset.collate_fn (pytorchltr/datasets/svmrank/svmrank.py:126-207, dense
branch): list_size = max_i sampler.max_list_size(rel_i) (:139-141); features/relevance
zero-initialised (:147-150); a sample longer than list_size is gathered through the sampler's
indices (:158-160, :181-182, :186-189), otherwise copied to the front (:183-184, :191-193);
n = min(n_i, list_size) (:197).  Pinned by tests/test_collate.py against batches produced by the
real reference (tests/golden/collate_vectors.npz).
"""
import numpy as np


def collate_dense(xs, ys, offsets, indices, sampler_calls, max_list_size=None):
    """sampler_calls: iterator of index vectors, consumed once per truncated sample, in order."""
    xs = np.asarray(xs, dtype=np.float32)
    ys = np.asarray(ys, dtype=np.int64)
    offsets = np.asarray(offsets, dtype=np.int64)
    counts = [int(offsets[i + 1] - offsets[i]) for i in indices]
    sizes = [c if max_list_size is None else min(max_list_size, c) for c in counts]
    list_size = max(sizes) if sizes else 0
    B, F = len(indices), xs.shape[1]
    out_x = np.zeros((B, list_size, F), dtype=np.float32)
    out_y = np.zeros((B, list_size), dtype=np.int64)
    out_n = np.zeros(B, dtype=np.int64)
    calls = iter(sampler_calls)
    for b, q in enumerate(indices):
        lo, hi = int(offsets[q]), int(offsets[q + 1])
        if hi - lo > list_size:
            picked = np.asarray(next(calls), dtype=np.int64)
            out_x[b, :, :] = xs[lo:hi][picked]
            out_y[b, :len(picked)] = ys[lo:hi][picked]
        else:
            out_x[b, :hi - lo] = xs[lo:hi]
            out_y[b, :hi - lo] = ys[lo:hi]
        out_n[b] = min(hi - lo, list_size)
    return out_x, out_y, out_n


def collate_csr(indptr, indices, values, num_features, ys, offsets, batch, sampler_calls, max_list_size=None):
    """The sparse branch (svmrank.py:162-176,197-202) restated on a CSR split: the dense form of the batch.
    For samples that fit the list this is exactly the reference's sparse batch `.to_dense()`.  For truncated
    samples the reference's sparse branch indexes its COO columns with `sum(mask)` -- an integer tensor
    where a boolean mask was meant -- and returns arbitrary entries (its own tests check the SHAPE only,
    tests/datasets/svmrank/test_svmrank.py:196-268); the restatement gathers the sampled rows like the
    dense branch does, which is what the code evidently intends."""
    indptr = np.asarray(indptr, dtype=np.int64)
    N = len(indptr) - 1
    dense = np.zeros((N, num_features), dtype=np.float32)
    for r in range(N):
        for e in range(indptr[r], indptr[r + 1]):
            dense[r, indices[e]] += np.float32(values[e])
    return collate_dense(dense, ys, offsets, batch, sampler_calls, max_list_size)
