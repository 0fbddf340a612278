"""Ranking metrics and the pytrec_eval export: the names pytorchltr/evaluation/__init__.py:1-4
exports, same signatures."""
from pytorchltr_amd.evaluation.arp import arp
from pytorchltr_amd.evaluation.dcg import dcg, ndcg
from pytorchltr_amd.evaluation.trec import generate_pytrec_eval

__all__ = ["arp", "dcg", "ndcg", "generate_pytrec_eval"]
