"""List samplers: which documents of an over-long query survive truncation to
``max_list_size`` (reference: pytorchltr/datasets/list_sampler.py:5-61).

A sampler maps a query's relevance labels to an index vector; the device-side collate kernel
gathers by it.  They run on the host and draw from torch's CPU generator with the same calls in
the same order as the reference's samplers, so a seeded generator yields the same indices
(pinned by index vectors recorded from the reference, tests/golden/collate_vectors.npz).
"""
from typing import Optional

import torch as _torch


class _SamplerBase:
    """Shared state: the size limit and the (optional) generator every draw comes from."""

    def __init__(self, limit: Optional[int], generator: Optional[_torch.Generator]):
        self._limit = limit
        self._generator = generator

    def max_list_size(self, relevance) -> int:
        """Documents that survive: the query's length, capped by the limit."""
        count = int(relevance.shape[0])
        return count if self._limit is None else min(self._limit, count)

    def _shuffle(self, count: int) -> _torch.Tensor:
        if self._generator is None:
            return _torch.randperm(count)                 # torch's global CPU generator
        return _torch.randperm(count, generator=self._generator)


class ListSampler(_SamplerBase):
    """Keeps the leading documents (reference :5-16)."""

    def __init__(self, max_list_size: Optional[int] = None):
        super().__init__(max_list_size, None)

    def __call__(self, relevance: _torch.Tensor) -> _torch.Tensor:
        return _torch.arange(self.max_list_size(relevance), dtype=_torch.long)


class UniformSampler(ListSampler):
    """A uniformly random subset, without replacement (reference :19-27): one shuffle of the
    documents, cut at the limit."""

    def __init__(self, max_list_size: Optional[int] = None,
                 generator: Optional[_torch.Generator] = None):
        _SamplerBase.__init__(self, max_list_size, generator)

    def __call__(self, relevance: _torch.Tensor) -> _torch.Tensor:
        return self._shuffle(int(relevance.shape[0]))[:self.max_list_size(relevance)]


class BalancedRelevanceSampler(UniformSampler):
    """A random subset that deals the relevance grades out in turn, so that every grade is
    represented as evenly as its documents allow (reference :30-61).

    Two draws, in the reference's order: a shuffle of the distinct grades (their turn order),
    then a shuffle of the documents.  Each shuffled document gets the pair (how many documents
    of its grade came before it, its grade's turn); documents are taken in increasing order of
    that pair -- first one document per grade, then a second one, ... -- until the limit."""

    def __call__(self, relevance: _torch.Tensor) -> _torch.Tensor:
        limit = self.max_list_size(relevance)
        grades = _torch.unique(relevance)                          # ascending
        n_grades = int(grades.shape[0])
        turn_of = _torch.empty(n_grades, dtype=_torch.long)
        turn_of[self._shuffle(n_grades)] = _torch.arange(n_grades)  # grade index -> its turn
        doc_order = self._shuffle(int(relevance.shape[0]))
        shuffled = relevance[doc_order]
        turn = turn_of[_torch.searchsorted(grades, shuffled)]
        member = turn.unsqueeze(1) == _torch.arange(n_grades).unsqueeze(0)
        seen_before = (_torch.cumsum(member, dim=0) - 1).gather(1, turn.unsqueeze(1)).squeeze(1)
        eligible = _torch.nonzero(seen_before < limit, as_tuple=False).reshape(-1)
        deal = _torch.argsort(seen_before[eligible] * n_grades + turn[eligible])
        return doc_order[eligible[deal]][:limit]
