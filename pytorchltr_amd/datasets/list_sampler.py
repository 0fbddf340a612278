"""List samplers: which documents of an over-long query survive truncation to
``max_list_size`` (reference: pytorchltr/datasets/list_sampler.py:5-61).

These only produce index vectors from a query's relevance labels; they run on the host with
torch's CPU generator exactly like the reference's (same calls in the same order, so a seeded
generator yields the same indices), and the device-side collate kernel gathers by them.
"""
from typing import Optional

import torch as _torch


class ListSampler:
    """Keeps the first ``max_list_size`` documents (reference :5-16)."""

    def __init__(self, max_list_size: Optional[int] = None):
        self._max_list_size = max_list_size

    def max_list_size(self, relevance):
        size = int(relevance.shape[0])
        if self._max_list_size is not None:
            size = min(self._max_list_size, size)
        return size

    def __call__(self, relevance: _torch.LongTensor) -> _torch.LongTensor:
        return _torch.arange(self.max_list_size(relevance), dtype=_torch.long)


class UniformSampler(ListSampler):
    """Uniformly random subset without replacement (reference :19-27)."""

    def __init__(self, max_list_size: Optional[int] = None,
                 generator: Optional[_torch.Generator] = None):
        super().__init__(max_list_size)
        self.rng_kw = {} if generator is None else {"generator": generator}

    def __call__(self, relevance: _torch.LongTensor) -> _torch.LongTensor:
        order = _torch.randperm(int(relevance.shape[0]), **self.rng_kw)
        return order[:self.max_list_size(relevance)]


class BalancedRelevanceSampler(UniformSampler):
    """Random subset that round-robins over the relevance grades so every grade is represented
    as evenly as possible (reference :30-61)."""

    def __init__(self, max_list_size: Optional[int] = None,
                 generator: Optional[_torch.Generator] = None):
        super().__init__(max_list_size, generator)

    def __call__(self, relevance: _torch.LongTensor) -> _torch.LongTensor:
        n_docs = int(relevance.shape[0])
        limit = self.max_list_size(relevance)
        # two RNG draws, in the reference's order: a shuffle of the grades, then of the documents
        grades = _torch.unique(relevance)
        grades = grades[_torch.randperm(int(grades.shape[0]), **self.rng_kw)]
        doc_order = _torch.randperm(n_docs, **self.rng_kw)
        shuffled = relevance[doc_order]
        # column r of `slots` holds the r-th shuffled document of every grade (-1 = none):
        # reading it column by column interleaves the grades
        slots = _torch.full((int(grades.shape[0]), n_docs), -1, dtype=_torch.long)
        for row, grade in enumerate(grades):
            members = _torch.nonzero(shuffled == grade, as_tuple=False).reshape(-1)[:limit]
            slots[row, :members.shape[0]] = members
        picked = slots.t().reshape(-1)
        picked = picked[picked >= 0]
        return doc_order[picked][:limit]
