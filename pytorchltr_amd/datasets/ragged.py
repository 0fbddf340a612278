"""Device-side collate: ragged query storage -> padded batch, in one HIP launch.

The reference pads a batch with a Python loop over samples on the host
(``SVMRankDataset.collate_fn``, pytorchltr/datasets/svmrank/svmrank.py:126-207, dense path):
``features (B, L, F)`` zero-padded, ``relevance (B, L)`` int64 zero-padded, ``n = min(n_i, L)``,
``L = max_i min(max_list_size, n_i)``, over-long queries truncated by a ``ListSampler``.
``RaggedQueries`` keeps the whole split on the GPU as concatenated rows plus offsets and
produces the same batch with ``ltr_collate_pad_f32`` (a gather/pad copy at HBM speed); only the
index vectors of truncated queries are drawn on the host, by the same sampler calls in the same
order as the reference.
"""
from typing import List, Optional, Sequence

import torch as _torch

from pytorchltr_amd import _C
from pytorchltr_amd.datasets.list_sampler import ListSampler


class SVMRankBatch:
    """Same fields as the reference's batch object (svmrank.py:30-40)."""

    def __init__(self, features, relevance, n, qid, sparse=False):
        self.features = features
        self.relevance = relevance
        self.n = n
        self.qid = qid
        self.sparse = sparse


class RaggedQueries(_torch.utils.data.Dataset):
    """A split of a learning-to-rank dataset resident on the device.

    Args:
        features: (N, F) float32, the documents of all queries, query after query (None with `csr`).
        relevance: (N,) int64 labels.
        offsets: (Q + 1,) int64, documents of query q are rows offsets[q]:offsets[q+1].
        qids: optional (Q,) int64 query ids (default 0..Q-1).
        device: ROCm device the split lives on.
        csr: optional (indptr (N + 1), indices (nnz), values (nnz)): the split stored sparsely -- the reference's
            ``sparse=True`` datasets (svmrank.py:62-76,162-176); batches still come out dense and padded.
        num_features: F of a csr split (default: largest feature id + 1).
        pad_features_to: 4 lays the rows of every batch out a multiple of four floats apart (zero columns behind the
            real ones; the batch's ``features`` is still the (B, L, F) tensor -- a view of the padded one): feature
            counts that are not a multiple of 4 (Example3: 5, the reference's test file: 45, MQ2007 / MQ2008: 46) then
            take the 16-byte-vector kernels of this package (register tile, streaming scorer) instead of the scalar
            ones; ``torch.nn.Linear`` takes the view like any other tensor.  Default 1: rows packed like the reference's.
    """

    def __init__(self, features, relevance, offsets, qids=None, device="cuda", csr=None, num_features=None,
                 pad_features_to=1):
        relevance = _torch.as_tensor(relevance, dtype=_torch.int64)
        offsets = _torch.as_tensor(offsets, dtype=_torch.int64).cpu()
        self.csr = None
        if csr is not None:
            # the sparse branch (svmrank.py:162-176): the split as CSR, collated into DENSE padded batches
            indptr, indices, values = csr
            indptr = _torch.as_tensor(indptr, dtype=_torch.int64)
            indices = _torch.as_tensor(indices, dtype=_torch.int32)
            values = _torch.as_tensor(values, dtype=_torch.float32)
            if indptr.dim() != 1 or indptr.numel() != relevance.shape[0] + 1 or int(indptr[0]) != 0 or \
                    int(indptr[-1]) != indices.numel() or indices.numel() != values.numel() or \
                    bool((indptr[1:] < indptr[:-1]).any()):
                raise ValueError("csr must be (indptr (N + 1), indices (nnz), values (nnz)) over the N documents")
            if num_features is None:
                num_features = int(indices.max()) + 1 if indices.numel() else 1
            if indices.numel() and (int(indices.min()) < 0 or int(indices.max()) >= num_features):
                raise ValueError("feature ids must lie in [0, num_features)")
            self.num_features = int(num_features)
            features = _torch.empty(relevance.shape[0], 0)
        else:
            features = _torch.as_tensor(features, dtype=_torch.float32)
            self.num_features = None if features.dim() != 2 else int(features.shape[1])
        if features.dim() != 2 or relevance.dim() != 1 or features.shape[0] != relevance.shape[0]:
            raise ValueError("features must be (N, F) and relevance (N,)")
        if offsets.dim() != 1 or offsets.numel() < 1 or int(offsets[0]) != 0 or \
                int(offsets[-1]) != features.shape[0] or bool((offsets[1:] < offsets[:-1]).any()):
            raise ValueError("offsets must be non-decreasing, start at 0 and end at N")
        if pad_features_to not in (1, 4):
            raise ValueError("pad_features_to must be 1 or 4")
        if self.num_features is None:
            self.num_features = int(features.shape[1])
        self._row_width = (self.num_features + pad_features_to - 1) // pad_features_to * pad_features_to
        if csr is None and self._row_width != self.num_features:
            wide = _torch.zeros(features.shape[0], self._row_width, dtype=_torch.float32)
            wide[:, :self.num_features] = features
            features = wide
        self._q = offsets.numel() - 1
        self._offsets_host = offsets
        self._counts_host = offsets[1:] - offsets[:-1]
        self._rel_host = relevance.cpu()                 # the samplers look at labels on the host
        self._qids_host = (_torch.arange(self._q, dtype=_torch.int64) if qids is None
                           else _torch.as_tensor(qids, dtype=_torch.int64).cpu())
        dev = _torch.device(device)
        self.features = features.to(dev).contiguous()
        self.relevance = relevance.to(dev).contiguous()
        self.offsets = offsets.to(dev)
        if csr is not None:
            self.csr = (indptr.to(dev).contiguous(), indices.to(dev).contiguous(), values.to(dev).contiguous())
        _C.require_device(self.relevance, "relevance")

    @classmethod
    def from_dense_as_csr(cls, features, relevance, offsets, qids=None, device="cuda"):
        """A dense split stored sparsely (its non-zeros as CSR) -- e.g. the one-hot / bag-of-words feature
        sets the reference loads with ``sparse=True``."""
        x = _torch.as_tensor(features, dtype=_torch.float32)
        nz = x != 0
        counts = nz.sum(dim=1)
        indptr = _torch.cat([_torch.zeros(1, dtype=_torch.int64), _torch.cumsum(counts, 0)])
        rows, cols = nz.nonzero(as_tuple=True)
        return cls(None, relevance, offsets, qids=qids, device=device,
                   csr=(indptr, cols.to(_torch.int32), x[rows, cols]), num_features=x.shape[1])

    def __len__(self):
        return self._q

    def _check_index(self, index):
        i = int(index)
        if i < 0:
            i += self._q
        if not 0 <= i < self._q:
            raise IndexError("query index %d out of range for %d queries" % (int(index), self._q))
        return i

    def __getitem__(self, index):
        """Items are just query indices: the batch is assembled on the device by collate."""
        return self._check_index(index)

    def query_relevance(self, index):
        index = self._check_index(index)
        lo, hi = int(self._offsets_host[index]), int(self._offsets_host[index + 1])
        return self._rel_host[lo:hi]

    def plan(self, indices: Sequence[int], list_sampler: Optional[ListSampler] = None):
        """Host part of a collate: list size and, for truncated queries, the sampled document
        indices -- same calls in the same order as the reference's _collate_fn."""
        if list_sampler is None:
            list_sampler = ListSampler()
        rels = [self.query_relevance(i) for i in indices]
        list_size = max([list_sampler.max_list_size(r) for r in rels]) if rels else 0
        select = None
        for row, rel in enumerate(rels):
            if rel.shape[0] > list_size:
                picked = list_sampler(rel)
                if select is None:
                    select = _torch.arange(list_size, dtype=_torch.int64).repeat(len(rels), 1)
                select[row, :picked.shape[0]] = picked
        return list_size, select

    def collate(self, indices: Sequence[int], list_sampler: Optional[ListSampler] = None,
                sort_by_length: bool = False) -> SVMRankBatch:
        """Pads the given queries into one batch.  ``sort_by_length=True`` orders the batch by
        decreasing document count first (stable): per-query results do not depend on the order,
        and neighbouring workgroups then run for similar times -- the fused C2 step measured
        14.1 -> 12.4 us on length-sorted batches.  Off by default (the reference keeps the
        sampler's order)."""
        idx = _torch.as_tensor([self._check_index(i) for i in indices], dtype=_torch.int64)
        if sort_by_length and idx.numel() > 1:
            order = _torch.sort(self._counts_host[idx], descending=True, stable=True).indices
            idx = idx[order]
        B = idx.numel()
        list_size, select = self.plan(idx.tolist(), list_sampler)
        dev = self.relevance.device
        F = self._row_width                         # (the row width in memory: num_features, or padded to a multiple of 4)
        out_x = _torch.empty(B, list_size, F, dtype=_torch.float32, device=dev)
        out_y = _torch.empty(B, list_size, dtype=_torch.int64, device=dev)
        out_n = _torch.empty(B, dtype=_torch.int64, device=dev)
        qidx = idx.to(dev)
        sel = None if select is None else select.to(dev).contiguous()
        if B > 0 and list_size > 0 and self.csr is not None:
            with _C.device_ctx(out_x):
                _C.check(_C.lib().ltr_collate_pad_csr_f32(
                    _C.ptr(self.csr[0]), _C.ptr(self.csr[1]), _C.ptr(self.csr[2]), _C.ptr(self.relevance),
                    _C.ptr(self.offsets), _C.ptr(qidx), _C.ptr(sel), self._q, B, list_size, F, _C.ptr(out_x),
                    _C.ptr(out_y), _C.ptr(out_n), _C.stream_of(out_x)))
        elif B > 0 and list_size > 0:
            with _C.device_ctx(out_x):
                _C.check(_C.lib().ltr_collate_pad_f32(
                    _C.ptr(self.features), _C.ptr(self.relevance), _C.ptr(self.offsets),
                    _C.ptr(qidx), _C.ptr(sel), self._q, B, list_size, F, _C.ptr(out_x),
                    _C.ptr(out_y), _C.ptr(out_n), _C.stream_of(out_x)))
        elif B > 0:
            out_n.zero_()
        qid = self._qids_host[idx].to(dev)
        if F != self.num_features:
            out_x = out_x[:, :, :self.num_features]          # the reference's (B, L, F) batch, rows F4 floats apart
            out_x._ltr_zero_padded_rows = True               # (the hidden columns ARE zeros: what fused._padded_rows asks for)
        return SVMRankBatch(out_x, out_y, out_n, qid, False)

    def collate_fn(self, list_sampler: Optional[ListSampler] = None, sort_by_length: bool = False):
        """collate_fn for ``torch.utils.data.DataLoader(ragged, batch_size=..., collate_fn=...)``."""
        def _collate(batch: List[int]) -> SVMRankBatch:
            return self.collate(batch, list_sampler, sort_by_length)
        return _collate
