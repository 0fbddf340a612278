#!/usr/bin/env python
"""Golden vectors for the sampling row from the REAL reference.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/generate_sampling_golden.py

rank_by_plackettluce draws its uniform numbers internally, so the vectors are produced by
running the reference's own lines 79-91 (mask_padded_values, LogSoftmax, log(-log u) - log_p,
tiebreak_argsort) on an explicit `u`; simulate_pbm's propensities are deterministic and come
from the reference function itself, as do clicks for labels whose click probability is 0 or 1.
"""
import os
import sys

import numpy as np
import torch

REFERENCE = os.environ.get("PYTORCHLTR_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, REFERENCE)
from pytorchltr.click_simulation.pbm import (  # noqa: E402
    simulate_nearrandom, simulate_pbm, simulate_perfect, simulate_position)
from pytorchltr.utils.tensor_operations import mask_padded_values, tiebreak_argsort  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
arrays = {}
g = torch.Generator().manual_seed(99)
for name, (B, L) in {"pl_small": (6, 9), "pl_c2": (16, 128)}.items():
    scores = torch.randn(B, L, generator=g) * 2
    n = torch.randint(0, L + 1, (B,), generator=g)
    n[0] = L
    u = torch.rand(B, L, generator=g)
    masked = mask_padded_values(scores, n)
    log_p = torch.nn.LogSoftmax(dim=1)(masked)
    r = torch.log(-torch.log(u)) - log_p
    torch.manual_seed(0)
    ranking = tiebreak_argsort(r, descending=False)
    for k, v in (("scores", scores), ("n", n), ("u", u), ("ranking", ranking)):
        arrays["%s/%s" % (name, k)] = v.numpy()

for name, (B, L) in {"pbm_small": (5, 7), "pbm_c2": (12, 128)}.items():
    rankings = torch.stack([torch.randperm(L, generator=g) for _ in range(B)])
    ys = torch.randint(0, 5, (B, L), generator=g)
    n = torch.randint(0, L + 1, (B,), generator=g)
    n[0] = L
    arrays[name + "/rankings"] = rankings.numpy()
    arrays[name + "/ys"] = ys.numpy()
    arrays[name + "/n"] = n.numpy()
    for tag, fn in (("perfect", lambda: simulate_perfect(rankings, ys, n)),
                    ("perfect_cut3", lambda: simulate_perfect(rankings, ys, n, cutoff=3)),
                    ("position", lambda: simulate_position(rankings, ys, n)),
                    ("position_eta2_cut5", lambda: simulate_position(rankings, ys, n, cutoff=5, eta=2.0)),
                    ("nearrandom_eta0", lambda: simulate_nearrandom(rankings, ys, n, eta=0.0))):
        torch.manual_seed(1)
        clicks, props = fn()
        arrays["%s/%s/props" % (name, tag)] = props.numpy()
    # deterministic clicks: labels restricted to {0, 4} under the perfect model (p in {0, 1})
    ys04 = (ys >= 2).long() * 4
    torch.manual_seed(1)
    clicks, props = simulate_perfect(rankings, ys04, n, cutoff=4)
    arrays[name + "/ys04"] = ys04.numpy()
    arrays[name + "/perfect04_cut4/clicks"] = clicks.numpy()
    arrays[name + "/perfect04_cut4/props"] = props.numpy()
    # a custom relevance_probs vector through simulate_pbm
    probs = torch.tensor([0.05, 0.3, 0.5, 0.7, 0.95])
    torch.manual_seed(1)
    _, props = simulate_pbm(rankings, ys, n, probs, cutoff=None, eta=0.5)
    arrays[name + "/custom_probs"] = probs.numpy()
    arrays[name + "/custom_eta05/props"] = props.numpy()

np.savez_compressed(os.path.join(HERE, "sampling_vectors.npz"), **arrays)
print("wrote %d arrays" % len(arrays))
