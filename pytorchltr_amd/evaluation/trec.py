"""pytrec_eval export (mirror of pytorchltr/evaluation/trec.py:11-86).

Pure host-side formatting -- no kernel: the batch is brought to the host in one transfer per
tensor (the reference indexes the tensors element by element, i.e. one device sync per document
when they live on a GPU) and turned into the ``qrel`` / ``run`` dictionaries pytrec_eval takes.
"""
from typing import Dict, Optional, Tuple

import torch as _torch

_PYTREC_RETURN_TYPE = Tuple[Dict[str, Dict[str, int]], Dict[str, Dict[str, float]]]


def generate_pytrec_eval(scores: _torch.Tensor, relevance: _torch.Tensor, n: _torch.Tensor,
                         qids: Optional[_torch.Tensor] = None, qid_offset: int = 0,
                         q_prefix: str = "q", d_prefix: str = "d") -> _PYTREC_RETURN_TYPE:
    """Same arguments and return value as the reference: query ids are ``q_prefix + qid`` (the
    given ``qids`` or the row index plus ``qid_offset``), document ids ``d_prefix + position``,
    and only the first ``n[i]`` documents of a row are listed."""
    batch = scores.shape[0]
    score_rows = scores.detach().reshape(batch, -1).cpu().tolist()      # at the input's precision, like float(scores[i, d])
    label_rows = relevance.detach().reshape(batch, -1).cpu().tolist()
    counts = n.detach().reshape(-1).cpu().tolist()
    ids = None if qids is None else qids.detach().reshape(-1).cpu().tolist()
    qrel, run = {}, {}
    for i in range(batch):
        key = "%s%d" % (q_prefix, int(ids[i]) if ids is not None else i + qid_offset)
        docs = range(int(counts[i]))
        qrel[key] = {"%s%d" % (d_prefix, d): int(label_rows[i][d]) for d in docs}
        run[key] = {"%s%d" % (d_prefix, d): float(score_rows[i][d]) for d in docs}
    return qrel, run
