"""Average Relevant Position on MI355X (reference: evaluation/arp.py:7-42)."""
import torch as _torch

from pytorchltr_amd import _C
from pytorchltr_amd import _ties
from pytorchltr_amd._prepare import prepare as _prepare


def arp(scores: _torch.FloatTensor, relevance: _torch.LongTensor,
        n: _torch.LongTensor) -> _torch.FloatTensor:
    r"""ARP: :math:`\frac{1}{\sum_i y_i} \sum_i y_{\pi_i} \cdot i` over the real documents.

    Args:
        scores: (batch, list_size[, 1]) scores.
        relevance: (batch, list_size) relevance labels.
        n: (batch,) number of documents per query.

    Returns:
        (batch,) ARP per query (0 when a query has no relevant document).
    """
    s, r, nn = _prepare(scores, relevance, n)
    B, L = s.shape
    out = _torch.empty(B, dtype=_torch.float32, device=s.device)
    if B > 0:
        sd = _ties.draw_seed(L, s.device)             # random tie-break, as the reference (arp.py:32)
        with _C.device_ctx(s):
            if sd is None:
                _C.check(_C.lib().ltr_arp_f32(_C.ptr(s), _C.ptr(r), _C.label_dtype(r), _C.ptr(nn), B, L,
                                              _C.ptr(out), _C.stream_of(s)))
            else:
                _C.check(_C.lib().ltr_arp_seed_f32(_C.ptr(s), _C.ptr(r), _C.label_dtype(r), _C.ptr(nn), sd[0],
                                                   _C.ptr(sd[1]), B, L, _C.ptr(out), _C.stream_of(s)))
    return out
