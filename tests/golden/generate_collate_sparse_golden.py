#!/usr/bin/env python
"""Golden vectors for the SPARSE branch of the collate (reference
datasets/svmrank/svmrank.py:162-176,197-202) from the REAL reference.

    rm -rf /tmp/ref_build && cp -r /root/reference /tmp/ref_build
    (cd /tmp/ref_build && python setup.py build_ext --inplace)
    PYTORCHLTR_REFERENCE=/tmp/ref_build python tests/golden/generate_collate_sparse_golden.py

Writes tests/golden/collate_sparse_vectors.npz: random sparse splits as CSR, and per case the batch
indices and the dense form (`.to_dense()`) of the reference's sparse batch -- for samples that fit the
list.  For truncated samples the reference's sparse branch is broken (it indexes COO columns with an
integer tensor where a boolean mask was meant; its own tests check the shape only), so those cases store
the reference's DENSE-branch batch for the same data and the same sampler draws, next to the shape,
relevance and n of the sparse batch (which the reference computes correctly).  Data only.
"""
import json
import os
import sys

import numpy as np
import torch

REFERENCE = os.environ.get("PYTORCHLTR_REFERENCE", "/tmp/ref_build")
sys.dont_write_bytecode = True
sys.path.insert(0, REFERENCE)

from pytorchltr.datasets.list_sampler import ListSampler, UniformSampler  # noqa: E402
from pytorchltr.datasets.svmrank.svmrank import SVMRankDataset, SVMRankItem  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
arrays, cases = {}, []


def make_split(seed, Q, F, max_n, density):
    g = torch.Generator().manual_seed(seed)
    counts = torch.randint(1, max_n + 1, (Q,), generator=g)
    offsets = torch.cat([torch.zeros(1, dtype=torch.long), torch.cumsum(counts, 0)])
    N = int(offsets[-1])
    dense = torch.randn(N, F, generator=g) * (torch.rand(N, F, generator=g) < density)
    dense[3 % N] = 0.0                                            # an all-zero document
    ys = torch.randint(0, 5, (N,), generator=g)
    return dense, ys, offsets


class Recording:
    def __init__(self, inner):
        self.inner, self.calls = inner, []

    def max_list_size(self, relevance):
        return self.inner.max_list_size(relevance)

    def __call__(self, relevance):
        out = self.inner(relevance)
        self.calls.append(out.clone())
        return out


def items(dense, ys, offsets, indices, sparse):
    out = []
    for q in indices:
        lo, hi = int(offsets[q]), int(offsets[q + 1])
        x = dense[lo:hi].clone()
        out.append(SVMRankItem(x.to_sparse() if sparse else x, ys[lo:hi].clone(), hi - lo, 100 + q, sparse))
    return out


for split_name, (seed, Q, F, max_n, density) in {"p45": (11, 10, 45, 14, 0.3), "p700": (12, 5, 700, 30, 0.05)}.items():
    dense, ys, offsets = make_split(seed, Q, F, max_n, density)
    nz = dense != 0
    indptr = torch.cat([torch.zeros(1, dtype=torch.long), torch.cumsum(nz.sum(1), 0)])
    rows, cols = nz.nonzero(as_tuple=True)
    arrays[split_name + "/indptr"] = indptr.numpy()
    arrays[split_name + "/indices"] = cols.numpy().astype(np.int32)
    arrays[split_name + "/values"] = dense[rows, cols].numpy()
    arrays[split_name + "/ys"] = ys.numpy()
    arrays[split_name + "/offsets"] = offsets.numpy()
    rng = np.random.RandomState(seed)
    for ci, (sampler_name, limit, sseed) in enumerate([("list", None, None), ("uniform", None, 3), ("uniform", 6, 4),
                                                       ("list", 5, None)]):
        idx = rng.choice(Q, size=min(Q, 4 + ci % 2), replace=False).tolist()
        name = "%s_%s_%s_%d" % (split_name, sampler_name, limit, ci)

        def sampler():
            kw = {"generator": torch.Generator().manual_seed(sseed)} if sseed is not None else {}
            return Recording((ListSampler if sampler_name == "list" else UniformSampler)(limit, **kw))
        sp = sampler()
        sbatch = SVMRankDataset.collate_fn(sp)(items(dense, ys, offsets, idx, True))
        dn = sampler()
        dbatch = SVMRankDataset.collate_fn(dn)(items(dense, ys, offsets, idx, False))
        truncated = len(sp.calls) > 0
        assert sbatch.sparse and tuple(sbatch.features.shape) == tuple(dbatch.features.shape)
        assert torch.equal(sbatch.relevance, dbatch.relevance) and torch.equal(sbatch.n, dbatch.n)
        arrays[name + "/indices"] = np.asarray(idx, dtype=np.int64)
        arrays[name + "/relevance"] = sbatch.relevance.numpy()
        arrays[name + "/n"] = sbatch.n.numpy()
        arrays[name + "/shape"] = np.asarray(sbatch.features.shape, dtype=np.int64)
        if truncated:
            arrays[name + "/features"] = dbatch.features.numpy()          # see the module docstring
        else:
            arrays[name + "/features"] = sbatch.features.to_dense().numpy()
            assert torch.equal(sbatch.features.to_dense(), dbatch.features)
        for k, c in enumerate(dn.calls):
            arrays[name + "/call%d" % k] = c.numpy()
        cases.append({"name": name, "split": split_name, "sampler": sampler_name, "max_list_size": limit,
                      "seed": sseed, "n_calls": len(dn.calls), "truncated": truncated, "features": F})

np.savez_compressed(os.path.join(HERE, "collate_sparse_vectors.npz"), **arrays)
with open(os.path.join(HERE, "collate_sparse_vectors.json"), "w") as fh:
    json.dump({"cases": cases}, fh, indent=1, sort_keys=True)
print("wrote %d arrays, %d cases" % (len(arrays), len(cases)), [c["name"] + (" T" if c["truncated"] else "") for c in cases])
