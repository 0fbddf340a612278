"""Data layer pieces adjacent to the hot path: list samplers, the
device-side collate/padding of ragged query storage into the padded (B, L, F) batch the loss
and metric kernels consume, and the SVMrank text parser that fills that storage."""
from pytorchltr_amd.datasets.list_sampler import BalancedRelevanceSampler, ListSampler, UniformSampler
from pytorchltr_amd.datasets.ragged import RaggedQueries, SVMRankBatch
from pytorchltr_amd.datasets.svmrank import load_svmrank, parse_svmrank_file

__all__ = ["ListSampler", "UniformSampler", "BalancedRelevanceSampler", "RaggedQueries",
           "SVMRankBatch", "load_svmrank", "parse_svmrank_file"]
