"""Additive pairwise losses on MI355X.

API parity with the reference's loss/pairwise_additive.py: parameter-free ``nn.Module``s
whose ``forward(scores, relevance, n)`` returns one loss per query, shape (batch,).  Where
the reference expands all (i, j) document pairs into (B, L, L, 2) tensors (:68-69) and masks
them (:75-81), these modules launch one fused HIP kernel (``ltr_pairwise_loss_f32``) that
also produces the gradient the reference gets from autograd.
"""
import torch as _torch

from pytorchltr_amd import _C
from pytorchltr_amd._autograd import pairwise_loss as _pairwise_loss


class _PairwiseAdditiveLoss(_torch.nn.Module):
    """Base of the linearly decomposable pairwise losses (reference :5-90)."""
    _kind = None

    def __init__(self):
        super().__init__()

    def _sigma(self):
        return 1.0

    def forward(self, scores: _torch.FloatTensor, relevance: _torch.LongTensor,
                n: _torch.LongTensor) -> _torch.FloatTensor:
        """Per-query loss.

        Args:
            scores: (batch, list_size) or (batch, list_size, 1) scores.
            relevance: (batch, list_size) or (batch, list_size, 1) relevance labels.
            n: (batch,) number of real (non-padded) documents per query.
        """
        if self._kind is None:
            raise NotImplementedError
        return _pairwise_loss(scores, relevance, n, self._kind, self._sigma())


class PairwiseHingeLoss(_PairwiseAdditiveLoss):
    r"""RankSVM hinge loss (reference :93-113):
    :math:`l(s, y) = \sum_{y_i > y_j} \max(0, 1 - (s_i - s_j))`."""
    _kind = _C.HINGE


class PairwiseDCGHingeLoss(PairwiseHingeLoss):
    r"""DCG-modified hinge loss (reference :116-133):
    :math:`l(s, y) = -1 / \ln(2 + \sum_{y_i > y_j} \max(0, 1 - (s_i - s_j)))`."""
    _kind = _C.DCG_HINGE


class PairwiseLogisticLoss(_PairwiseAdditiveLoss):
    r"""RankNet logistic loss (reference :136-163):
    :math:`l(s, y) = \sum_{y_i > y_j} \log_2(1 + e^{-\sigma (s_i - s_j)})`."""
    _kind = _C.LOGISTIC

    def __init__(self, sigma: float = 1.0):
        """
        Args:
            sigma: Steepness of the logistic curve.
        """
        super().__init__()
        self.sigma = sigma

    def _sigma(self):
        return self.sigma
