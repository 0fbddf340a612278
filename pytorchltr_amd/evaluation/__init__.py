"""Ranking metrics -- same names as pytorchltr/evaluation/__init__.py:1-4."""
from pytorchltr_amd.evaluation.arp import arp  # noqa: F401
from pytorchltr_amd.evaluation.dcg import ndcg  # noqa: F401
from pytorchltr_amd.evaluation.dcg import dcg  # noqa: F401
from pytorchltr_amd.evaluation.trec import generate_pytrec_eval  # noqa: F401
