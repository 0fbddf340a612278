#!/usr/bin/env python
"""The reference's getting-started guide (docs/source/getting-started.rst) on pytorchltr_amd:
a 136-50-10-1 ReLU network trained with PairwiseHingeLoss + Adagrad on lists truncated to 20
documents by UniformSampler, evaluated with ndcg@10 -- with the guide's `Model` + `loss_fn(...)
.mean()` written as ONE module (FusedMLPLoss: one MFMA kernel per step) and `model(xs)` in the
evaluation loop as its `score(xs, n)`.

The guide trains on MSLR-WEB10K, which is not available offline; this script makes a synthetic
split of the same shape (136 features, ragged queries, 5 relevance grades that depend on the
features through a fixed random network), so the printed ndcg@10 must RISE from its untrained
value.  Pass an SVMrank file pair to train on real data instead:

    python examples/02_mlp_getting_started.py [train.txt test.txt]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pytorchltr_amd.datasets import RaggedQueries, UniformSampler, load_svmrank  # noqa: E402
from pytorchltr_amd.evaluation import ndcg  # noqa: E402
from pytorchltr_amd.fused import FusedMLPLoss  # noqa: E402
from pytorchltr_amd.loss import PairwiseHingeLoss  # noqa: E402


def synthetic_split(queries, features, seed, device):
    g = torch.Generator().manual_seed(seed)
    counts = torch.randint(5, 121, (queries,), generator=g)
    offsets = torch.cat([torch.zeros(1, dtype=torch.long), torch.cumsum(counts, 0)])
    xs = torch.randn(int(offsets[-1]), features, generator=g)
    teacher = torch.Generator().manual_seed(1234)                 # the same "truth" for every split
    w1 = torch.randn(features, 8, generator=teacher) / features ** 0.5
    w2 = torch.randn(8, generator=teacher)
    signal = torch.tanh(xs @ w1) @ w2 + 0.3 * torch.randn(xs.shape[0], generator=g)
    ys = torch.bucketize(signal, torch.tensor([0.5, 1.2, 1.9, 2.6])).long()   # grades 0..4, skewed to 0
    return RaggedQueries(xs, ys, offsets, device=device)


def run(train_path=None, test_path=None, epochs=5, device="cuda", log=print):
    torch.manual_seed(42)
    if train_path:
        train = load_svmrank(train_path, normalize=True, filter_queries=True, device=device)
        test = load_svmrank(test_path, normalize=True, device=device)
    else:
        train = synthetic_split(600, 136, 1, device)
        test = synthetic_split(200, 136, 2, device)
    model = FusedMLPLoss(train.features.shape[1], PairwiseHingeLoss()).to(device)
    optimizer = torch.optim.Adagrad(model.parameters(), lr=0.1)

    def evaluate():
        loader = torch.utils.data.DataLoader(test, batch_size=16, collate_fn=test.collate_fn())
        total = 0.0
        with torch.no_grad():
            for batch in loader:
                scores = model.score(batch.features, batch.n)
                total += float(torch.sum(ndcg(scores, batch.relevance, batch.n, k=10)))
        return total / len(test)

    trace = [evaluate()]
    log("ndcg@10 on test set before training: %f" % trace[0])
    for epoch in range(1, epochs + 1):
        loader = torch.utils.data.DataLoader(
            train, batch_size=16, shuffle=True,
            collate_fn=train.collate_fn(UniformSampler(max_list_size=20)))
        for batch in loader:
            loss = model(batch.features, batch.relevance, batch.n)      # = loss_fn(model(xs), ys, n).mean()
            optimizer.zero_grad()
            loss.backward()
            optimizer.step()
        trace.append(evaluate())
        log("Finished epoch %d: ndcg@10 on test set %f" % (epoch, trace[-1]))
    return trace


if __name__ == "__main__":
    args = sys.argv[1:]
    run(*(args[:2] if len(args) >= 2 else ()))
