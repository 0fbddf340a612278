#!/usr/bin/env python
"""Golden vectors for the collate/padding row from the REAL reference.

The reference's datasets package imports its Cython parser at import time, so it must be built
first -- in a scratch copy, never in /root/reference:

    rm -rf /tmp/ref_build && cp -r /root/reference /tmp/ref_build
    (cd /tmp/ref_build && python setup.py build_ext --inplace)
    PYTORCHLTR_REFERENCE=/tmp/ref_build python tests/golden/generate_collate_golden.py

Writes tests/golden/collate_vectors.npz: a random ragged split, and for each case the batch
indices, the sampler spec, every index vector the sampler returned (in call order) and the
reference's padded batch.  Data only.
"""
import json
import os
import sys

import numpy as np
import torch

REFERENCE = os.environ.get("PYTORCHLTR_REFERENCE", "/tmp/ref_build")
sys.dont_write_bytecode = True
sys.path.insert(0, REFERENCE)

from pytorchltr.datasets.list_sampler import (  # noqa: E402
    BalancedRelevanceSampler, ListSampler, UniformSampler)
from pytorchltr.datasets.svmrank.svmrank import SVMRankDataset, SVMRankItem  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
arrays = {}
cases = []

SAMPLERS = {"list": ListSampler, "uniform": UniformSampler, "balanced": BalancedRelevanceSampler}


def make_split(seed, Q, F, max_n):
    g = torch.Generator().manual_seed(seed)
    counts = torch.randint(1, max_n + 1, (Q,), generator=g)
    offsets = torch.cat([torch.zeros(1, dtype=torch.long), torch.cumsum(counts, 0)])
    N = int(offsets[-1])
    xs = torch.randn(N, F, generator=g)
    ys = torch.randint(0, 5, (N,), generator=g)
    qids = torch.arange(100, 100 + Q)
    return xs, ys, offsets, qids


class Recording:
    """Wraps a reference sampler and records every index vector it hands out."""
    def __init__(self, inner):
        self.inner = inner
        self.calls = []

    def max_list_size(self, relevance):
        return self.inner.max_list_size(relevance)

    def __call__(self, relevance):
        out = self.inner(relevance)
        self.calls.append(out.clone())
        return out


def add_case(split_name, split, name, indices, sampler_name, max_list_size, seed):
    xs, ys, offsets, qids = split
    kw = {}
    if sampler_name != "list" and seed is not None:
        kw["generator"] = torch.Generator().manual_seed(seed)
    sampler = Recording(SAMPLERS[sampler_name](max_list_size, **kw))
    items = []
    for q in indices:
        lo, hi = int(offsets[q]), int(offsets[q + 1])
        items.append(SVMRankItem(xs[lo:hi].clone(), ys[lo:hi].clone(), hi - lo, int(qids[q]), False))
    batch = SVMRankDataset.collate_fn(sampler)(items)
    arrays[name + "/indices"] = np.asarray(indices, dtype=np.int64)
    arrays[name + "/features"] = batch.features.numpy()
    arrays[name + "/relevance"] = batch.relevance.numpy()
    arrays[name + "/n"] = batch.n.numpy()
    arrays[name + "/qid"] = batch.qid.numpy()
    for k, c in enumerate(sampler.calls):
        arrays[name + "/call%d" % k] = c.numpy()
    cases.append({"name": name, "split": split_name, "sampler": sampler_name,
                  "max_list_size": max_list_size, "seed": seed, "n_calls": len(sampler.calls)})


for split_name, (seed, Q, F, max_n) in {"s8": (5, 14, 8, 40), "s5": (6, 9, 5, 25), "s136": (7, 6, 136, 150)}.items():
    split = make_split(seed, Q, F, max_n)
    for k, a in zip(("xs", "ys", "offsets", "qids"), split):
        arrays[split_name + "/" + k] = a.numpy()
    rng = np.random.RandomState(seed)
    for ci in range(1 if split_name == "s136" else 3):    # the wide split: one round (fixture size)
        idx = rng.choice(Q, size=min(Q, 5 + ci), replace=False).tolist()
        add_case(split_name, split, "%s_none_%d" % (split_name, ci), idx, "list", None, None)
        add_case(split_name, split, "%s_list10_%d" % (split_name, ci), idx, "list", 10, None)
        add_case(split_name, split, "%s_uniform12_%d" % (split_name, ci), idx, "uniform", 12, 40 + ci)
        add_case(split_name, split, "%s_balanced9_%d" % (split_name, ci), idx, "balanced", 9, 50 + ci)
    add_case(split_name, split, "%s_single" % split_name, [0], "uniform", 3, 1)
    if split_name != "s136":
        add_case(split_name, split, "%s_huge_limit" % split_name, list(range(Q)), "balanced", 10000, 2)

np.savez_compressed(os.path.join(HERE, "collate_vectors.npz"), **arrays)
with open(os.path.join(HERE, "collate_vectors.json"), "w") as fh:
    json.dump({"cases": cases}, fh, indent=1, sort_keys=True)
print("wrote %d arrays, %d cases" % (len(arrays), len(cases)))
