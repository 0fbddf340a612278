"""Ranking losses.  The seven classes the reference exports (pytorchltr/loss/__init__.py:1-7),
same constructor arguments and ``forward(scores, relevance, n)``, computed by the HIP kernels
behind ``ltr_pairwise_loss_f32`` -- plus ``ListwiseSoftmaxLoss`` (ListNet), which the project
brief names and the reference does not have (parity unpinned; see loss/listwise.py)."""
from pytorchltr_amd.loss.listwise import ListNetLoss, ListwiseSoftmaxLoss
from pytorchltr_amd.loss.pairwise_additive import (
    PairwiseDCGHingeLoss,
    PairwiseHingeLoss,
    PairwiseLogisticLoss,
)
from pytorchltr_amd.loss.pairwise_lambda import (
    LambdaARPLoss1,
    LambdaARPLoss2,
    LambdaNDCGLoss1,
    LambdaNDCGLoss2,
)

__all__ = [
    "PairwiseHingeLoss", "PairwiseDCGHingeLoss", "PairwiseLogisticLoss",
    "LambdaARPLoss1", "LambdaARPLoss2", "LambdaNDCGLoss1", "LambdaNDCGLoss2",
    "ListwiseSoftmaxLoss", "ListNetLoss",
]
