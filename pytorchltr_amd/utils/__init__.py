"""Tensor helpers: the five names pytorchltr/utils/__init__.py:1-5 exports, same signatures."""
from pytorchltr_amd.utils.tensor_operations import (
    batch_pairs,
    mask_padded_values,
    rank_by_plackettluce,
    rank_by_score,
    tiebreak_argsort,
)

from pytorchltr_amd._ties import get_tie_breaking, set_tie_breaking, tie_breaking

__all__ = ["mask_padded_values", "tiebreak_argsort", "rank_by_score", "rank_by_plackettluce",
           "batch_pairs", "tie_breaking", "set_tie_breaking", "get_tie_breaking"]
