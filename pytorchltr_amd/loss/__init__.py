"""Ranking losses -- same names as pytorchltr/loss/__init__.py:1-7."""
from pytorchltr_amd.loss.pairwise_additive import PairwiseHingeLoss  # noqa: F401
from pytorchltr_amd.loss.pairwise_additive import PairwiseDCGHingeLoss  # noqa: F401
from pytorchltr_amd.loss.pairwise_additive import PairwiseLogisticLoss  # noqa: F401
from pytorchltr_amd.loss.pairwise_lambda import LambdaARPLoss1  # noqa: F401
from pytorchltr_amd.loss.pairwise_lambda import LambdaARPLoss2  # noqa: F401
from pytorchltr_amd.loss.pairwise_lambda import LambdaNDCGLoss1  # noqa: F401
from pytorchltr_amd.loss.pairwise_lambda import LambdaNDCGLoss2  # noqa: F401
