"""generate_pytrec_eval: the reference's known answers (tests/evaluation/test_trec.py) and its
docstring example.  Host-side formatting only, so it runs in the CPU tier."""
import torch

from pytorchltr_amd.evaluation import generate_pytrec_eval

SCORES = torch.tensor([[10.0, 5.0, 2.0, 3.0, 4.0], [5.0, 6.0, 4.0, 2.0, 5.5]])
YS = torch.tensor([[0, 1, 1, 0, 1], [3, 1, 0, 1, 0]])
N = torch.tensor([5, 4])


def test_default_prefixes():
    qrels, run = generate_pytrec_eval(SCORES, YS, N)
    assert qrels == {"q0": {"d0": 0, "d1": 1, "d2": 1, "d3": 0, "d4": 1},
                     "q1": {"d0": 3, "d1": 1, "d2": 0, "d3": 1}}
    assert run == {"q0": {"d0": 10.0, "d1": 5.0, "d2": 2.0, "d3": 3.0, "d4": 4.0},
                   "q1": {"d0": 5.0, "d1": 6.0, "d2": 4.0, "d3": 2.0}}


def test_no_prefix_and_offset():
    qrels, run = generate_pytrec_eval(SCORES, YS, N, q_prefix="", d_prefix="")
    assert qrels == {"0": {"0": 0, "1": 1, "2": 1, "3": 0, "4": 1}, "1": {"0": 3, "1": 1, "2": 0, "3": 1}}
    assert run == {"0": {"0": 10.0, "1": 5.0, "2": 2.0, "3": 3.0, "4": 4.0},
                   "1": {"0": 5.0, "1": 6.0, "2": 4.0, "3": 2.0}}
    qrels, _ = generate_pytrec_eval(SCORES, YS, N, qid_offset=7)
    assert sorted(qrels) == ["q7", "q8"]


def test_explicit_qids_and_trailing_unit_dim():
    qid = torch.tensor([15623, 49998])
    qrels, run = generate_pytrec_eval(SCORES.unsqueeze(-1), YS, N, qid, q_prefix="")
    assert qrels == {"15623": {"d0": 0, "d1": 1, "d2": 1, "d3": 0, "d4": 1},
                     "49998": {"d0": 3, "d1": 1, "d2": 0, "d3": 1}}
    assert run["49998"] == {"d0": 5.0, "d1": 6.0, "d2": 4.0, "d3": 2.0}
    assert all(isinstance(v, int) for v in qrels["15623"].values())
    assert all(isinstance(v, float) for v in run["15623"].values())


def test_scores_keep_their_input_precision():
    """the reference exports float(scores[i, d]) at the input's precision."""
    import torch
    from pytorchltr_amd.evaluation import generate_pytrec_eval
    s = torch.tensor([[0.1234567890123, 2.0]], dtype=torch.float64)
    qrel, run = generate_pytrec_eval(s, torch.tensor([[1, 0]]), torch.tensor([2]))
    assert run["q0"]["d0"] == 0.1234567890123                  # not rounded through fp32
    s32 = s.float()
    _, run32 = generate_pytrec_eval(s32, torch.tensor([[1, 0]]), torch.tensor([2]))
    assert run32["q0"]["d0"] == float(s32[0, 0])
