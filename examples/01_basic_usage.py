#!/usr/bin/env python
"""The reference's examples/01-basic-usage.py on pytorchltr_amd, end to end on the GPU:
SVMrank text -> parser (libltr_io) -> device-resident split -> device collate -> Linear scorer ->
PairwiseHingeLoss (HIP) -> SGD, evaluated with ndcg@10 (HIP).

    python examples/01_basic_usage.py [train.dat test.dat]

Without arguments it uses the Example3 toy dataset bundled as test data
(tests/golden/example3_{train,test}.dat: the public files of
http://download.joachims.org/svm_light/examples/example3.tar.gz, sha256-checked against the
digests the reference pins in pytorchltr/datasets/svmrank/example3.py:29-30).  The reference
prints for this data: start 0.8617, epoch 1 0.8617, epoch 2 1.0000, epoch 3 1.0000 (its batch
order comes from the global CPU RNG, which its tie-breaking also consumes; ours does not, so the
middle of the trace may differ -- start and end do not).
"""
import hashlib
import logging
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pytorchltr_amd.datasets import load_svmrank  # noqa: E402
from pytorchltr_amd.evaluation import ndcg  # noqa: E402
from pytorchltr_amd.loss import PairwiseHingeLoss  # noqa: E402

EXAMPLE3 = {
    "train": ("example3_train.dat", "503aa66c6a1b1bb8a86b14e52163dcdb5bcffc017981afdff4cf026eacc592cf"),
    "test": ("example3_test.dat", "81aaac13dfc5180edce38a588cec80ee00b5d85662e00d1b7ac1d3f98242698e"),
}


def example3_path(split):
    name, digest = EXAMPLE3[split]
    path = os.path.join(ROOT, "tests", "golden", name)
    with open(path, "rb") as f:
        if hashlib.sha256(f.read()).hexdigest() != digest:
            raise RuntimeError("%s does not match the pinned sha256" % path)
    return path


def run(train_path=None, test_path=None, epochs=3, device="cuda", log=logging.info, fast=False):
    """fast=True: the two lines that put the loop on the one-launch path -- `model = use_linear_scorer(model)` and
    `pytorchltr_amd.optim.SGD` in place of `torch.optim.SGD`; the loop body is the reference's either way."""
    torch.manual_seed(42)
    # Example3(normalize=True) in the reference (example3.py:39)
    train = load_svmrank(train_path or example3_path("train"), normalize=True, device=device)
    test = load_svmrank(test_path or example3_path("test"), normalize=True, device=device)
    model = torch.nn.Linear(train.features.shape[1], 1).to(device)
    if fast:
        from pytorchltr_amd.fused import use_linear_scorer
        from pytorchltr_amd.optim import SGD
        model = use_linear_scorer(model)
        optimizer = SGD(model.parameters(), lr=0.1)
    else:
        optimizer = torch.optim.SGD(model.parameters(), lr=0.1)
    loss_fn = PairwiseHingeLoss()

    def evaluate():
        model.eval()
        loader = torch.utils.data.DataLoader(test, batch_size=2, shuffle=True, collate_fn=test.collate_fn())
        total = 0.0
        with torch.no_grad():
            for batch in loader:
                total += float(torch.sum(ndcg(model(batch.features), batch.relevance, batch.n, k=10)))
        model.train()
        return total / len(test)

    trace = [evaluate()]
    log("Test nDCG at start: %.4f" % trace[-1])
    for epoch in range(epochs):
        loader = torch.utils.data.DataLoader(train, batch_size=2, shuffle=True, collate_fn=train.collate_fn())
        for batch in loader:
            loss = loss_fn(model(batch.features), batch.relevance, batch.n).mean()
            optimizer.zero_grad()
            loss.backward()
            optimizer.step()
        trace.append(evaluate())
        log("Test nDCG after epoch %d: %.4f" % (epoch + 1, trace[-1]))
    return trace


if __name__ == "__main__":
    logging.basicConfig(format="[%(levelname)s, %(module)s] %(message)s", level=logging.INFO)
    args = sys.argv[1:]
    run(*(args[:2] if len(args) >= 2 else ()))
