"""Tensor helpers -- same names as pytorchltr/utils/__init__.py:1-5."""
from pytorchltr_amd.utils.tensor_operations import mask_padded_values  # noqa: F401
from pytorchltr_amd.utils.tensor_operations import tiebreak_argsort  # noqa: F401
from pytorchltr_amd.utils.tensor_operations import rank_by_score  # noqa: F401
from pytorchltr_amd.utils.tensor_operations import batch_pairs  # noqa: F401
from pytorchltr_amd.utils.tensor_operations import rank_by_plackettluce  # noqa: F401
