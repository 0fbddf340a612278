"""Time the metric kernels at the bench shapes (graph-batched, per launch)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import WORKLOADS, synth
from _benchutil import time_launches
from pytorchltr_amd import _C
dev = torch.device("cuda:0")
lib = _C.lib()
for w in ("c2", "c4", "c5"):
    B, L, F, kind = WORKLOADS[w]
    scores, rel, n, X = synth(B, L, 4, 0, dev)
    out = torch.empty(B, L, device=dev)
    rk = torch.empty(B, L, dtype=torch.int64, device=dev)
    cs = lambda: torch.cuda.current_stream().cuda_stream
    fns = {
        "ndcg@10": lambda: _C.check(lib.ltr_dcg_f32(scores.data_ptr(), rel.data_ptr(), 0, n.data_ptr(), B, L, 10, 1, 1, out.data_ptr(), cs())),
        "dcg@10": lambda: _C.check(lib.ltr_dcg_f32(scores.data_ptr(), rel.data_ptr(), 0, n.data_ptr(), B, L, 10, 1, 0, out.data_ptr(), cs())),
        "ndcg curve": lambda: _C.check(lib.ltr_dcg_f32(scores.data_ptr(), rel.data_ptr(), 0, n.data_ptr(), B, L, 0, 1, 1, out.data_ptr(), cs())),
        "arp": lambda: _C.check(lib.ltr_arp_f32(scores.data_ptr(), rel.data_ptr(), 0, n.data_ptr(), B, L, out.data_ptr(), cs())),
        "rank_by_score": lambda: _C.check(lib.ltr_rank_by_score_f32(scores.data_ptr(), n.data_ptr(), B, L, rk.data_ptr(), cs())),
    }
    res = []
    for name, fn in fns.items():
        for _ in range(5):
            fn()
        t, _ = time_launches(fn, per_graph=20, replays=10)
        res.append("%s %.2f us (%.1f Mq/s)" % (name, t, B / t))
    print(w, "B=%d L=%d |" % (B, L), " | ".join(res), flush=True)
