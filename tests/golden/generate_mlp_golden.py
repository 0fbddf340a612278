#!/usr/bin/env python
"""Golden vectors for the MLP scorer + loss step from the REAL reference:
the guide's Model (docs/source/getting-started.rst:40-50) built from torch.nn.Linear layers, the
reference's own loss modules, `.mean().backward()` through autograd.  Run in the build container:

    python tests/golden/generate_mlp_golden.py        # reads /root/reference (pure Python path)

Writes tests/golden/mlp_vectors.npz: inputs, parameters, per-query losses, scores and the six
parameter gradients, in float64 (all seven losses) and float32 (guide-sized network).  Data only.
"""
import os
import sys

import numpy as np
import torch

REFERENCE = os.environ.get("PYTORCHLTR_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, REFERENCE)
from pytorchltr.loss import (LambdaARPLoss1, LambdaARPLoss2, LambdaNDCGLoss1, LambdaNDCGLoss2,  # noqa: E402
                             PairwiseDCGHingeLoss, PairwiseHingeLoss, PairwiseLogisticLoss)

HERE = os.path.dirname(os.path.abspath(__file__))
LOSSES = {"hinge": PairwiseHingeLoss, "dcg_hinge": PairwiseDCGHingeLoss, "logistic": PairwiseLogisticLoss,
          "arp1": LambdaARPLoss1, "arp2": LambdaARPLoss2, "ndcg1": LambdaNDCGLoss1, "ndcg2": LambdaNDCGLoss2}


class Model(torch.nn.Module):
    def __init__(self, in_features, h1, h2):
        super().__init__()
        self.l1 = torch.nn.Linear(in_features, h1)
        self.l2 = torch.nn.Linear(h1, h2)
        self.l3 = torch.nn.Linear(h2, 1)

    def forward(self, x):
        o1 = torch.nn.functional.relu(self.l1(x))
        o2 = torch.nn.functional.relu(self.l2(o1))
        return self.l3(o2)


def run(tag, B, L, F, h1, h2, dtype, seed, arrays):
    g = torch.Generator().manual_seed(seed)
    X = torch.randn(B, L, F, generator=g, dtype=dtype)
    y = torch.randint(0, 5, (B, L), generator=g)
    n = torch.randint(1, L + 1, (B,), generator=g)
    n[0] = L
    torch.manual_seed(seed)
    model = Model(F, h1, h2).to(dtype)
    arrays["%s/X" % tag] = X.numpy()
    arrays["%s/y" % tag] = y.numpy()
    arrays["%s/n" % tag] = n.numpy()
    for name, prm in model.named_parameters():
        arrays["%s/param/%s" % (tag, name)] = prm.detach().numpy().copy()
    for kind, cls in LOSSES.items():
        model.zero_grad()
        scores = model(X)
        loss = cls()(scores, y, n)
        loss.mean().backward()
        arrays["%s/%s/loss" % (tag, kind)] = loss.detach().numpy().copy()
        arrays["%s/%s/scores" % (tag, kind)] = scores.detach().numpy().reshape(B, L).copy()
        for name, prm in model.named_parameters():
            arrays["%s/%s/grad/%s" % (tag, kind, name)] = prm.grad.numpy().copy()


def main():
    arrays = {}
    run("f64_small", 6, 20, 12, 7, 3, torch.float64, 11, arrays)
    run("f32_guide", 4, 16, 136, 50, 10, torch.float32, 42, arrays)
    np.savez_compressed(os.path.join(HERE, "mlp_vectors.npz"), **arrays)
    print("wrote", len(arrays), "arrays")


if __name__ == "__main__":
    main()
