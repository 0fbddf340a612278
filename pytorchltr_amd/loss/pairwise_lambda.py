"""LambdaLoss family on MI355X.

API parity with the reference's loss/pairwise_lambda.py.  The reference sorts by score,
gathers scores and labels into rank order and evaluates a weighted log-sigmoid on every
(rank i, rank j) pair (:67-89).  Here the rank of every document is obtained by an in-LDS
counting rank and the pair weights are evaluated in document order inside one fused kernel;
the ranking is a constant w.r.t. autograd in both implementations.

Tie rule (documented deviation): score ties are broken by document index, not by the
reference's global-RNG permutation (utils/tensor_operations.py:43-45).
"""
import torch as _torch

from pytorchltr_amd import _C
from pytorchltr_amd._autograd import pairwise_loss as _pairwise_loss


class LambdaLoss(_torch.nn.Module):
    """LambdaLoss base (reference :6-92)."""
    _kind = None

    def __init__(self, sigma: float = 1.0):
        """
        Args:
            sigma: Steepness of the logistic curve.
        """
        super().__init__()
        self.sigma = sigma

    def forward(self, scores: _torch.FloatTensor, relevance: _torch.LongTensor,
                n: _torch.LongTensor) -> _torch.FloatTensor:
        """Per-query loss; arguments as for the additive losses."""
        if self._kind is None:
            raise NotImplementedError
        return _pairwise_loss(scores, relevance, n, self._kind, self.sigma)


class LambdaARPLoss1(LambdaLoss):
    r"""ARP Loss 1 (reference :95-117):
    :math:`-\sum_{i,j} \log_2 \mathrm{sigmoid}(\sigma (s_{\pi_i} - s_{\pi_j}))^{y_{\pi_i}}`."""
    _kind = _C.ARP1


class LambdaARPLoss2(LambdaLoss):
    r"""ARP Loss 2 (reference :120-140):
    :math:`\sum_{y_i > y_j} |y_i - y_j| \log_2(1 + e^{-\sigma (s_i - s_j)})`."""
    _kind = _C.ARP2


class LambdaNDCGLoss1(LambdaLoss):
    r"""NDCG Loss 1 (reference :143-173): exponent :math:`G_{\pi_i} / D_i` with
    :math:`G = (2^y - 1) / \mathrm{maxDCG}` and :math:`D_i = \log_2(2 + i)` (0-based rank)."""
    _kind = _C.NDCG1


class LambdaNDCGLoss2(LambdaLoss):
    r"""NDCG Loss 2 (reference :176-218): exponent :math:`\delta_{ij} |G_{\pi_i} - G_{\pi_j}|`
    over pairs with :math:`y_i > y_j`, :math:`\delta_{ij} = |1/D_{|i-j|} - 1/D_{|i-j|+1}|`,
    :math:`D_k = \log_2(2 + k)` as the reference code (not its docstring) has it."""
    _kind = _C.NDCG2
