"""Listwise softmax cross-entropy (ListNet top-one) on MI355X.

NOT part of the reference: ``pytorchltr/loss/__init__.py:1-7`` exports the seven pairwise classes
only and nothing in the reference's code, tests or docs defines a listwise loss.  The project
brief names "listwise softmax cross-entropy" on the hot path, so this module provides it with the
same call signature as the reference's losses -- **parity unpinned: there is no reference
implementation to pin it to**; the specification is ``include/ltr_hip.h``
(``ltr_listwise_softmax_f32``), restated in fp64 by ``oracle/ltr_oracle.c`` and checked against
finite differences.
"""
import torch as _torch

from pytorchltr_amd._autograd import LISTWISE_SOFTMAX as _LISTWISE_SOFTMAX
from pytorchltr_amd._autograd import pairwise_loss as _loss


class ListwiseSoftmaxLoss(_torch.nn.Module):
    r"""ListNet top-one cross-entropy between the label and the score distributions of a query:

    .. math::
        l(\mathbf{s}, \mathbf{y}) = -\sum_{j < n} \mathrm{softmax}(\mathbf{y})_j
        \ln \mathrm{softmax}(\mathbf{s})_j

    over the ``n`` real documents (padded documents take no part); ``n = 0`` gives 0.

    Shape:
        - scores: :math:`(N, \texttt{list\_size})` or :math:`(N, \texttt{list\_size}, 1)`
        - relevance: :math:`(N, \texttt{list\_size})`
        - n: :math:`(N)`
        - output: :math:`(N)`
    """

    def __init__(self):
        super().__init__()

    def forward(self, scores: _torch.FloatTensor, relevance: _torch.LongTensor,
                n: _torch.LongTensor) -> _torch.FloatTensor:
        return _loss(scores, relevance, n, _LISTWISE_SOFTMAX, 1.0)


ListNetLoss = ListwiseSoftmaxLoss
