"""Time the loss-only kernels (graph-batched, per launch) for every kind at several shapes."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synth
from _benchutil import time_launches
from pytorchltr_amd import _C
dev = torch.device("cuda:0")
lib = _C.lib()
KINDS = ("hinge", "dcg_hinge", "logistic", "arp1", "arp2", "ndcg1", "ndcg2")
shapes = [(1024, 128), (512, 512), (256, 1000)] if len(sys.argv) < 2 else [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for B, L in shapes:
    scores, rel, n, X = synth(B, L, 4, 0, dev)
    loss = torch.empty(B, device=dev); ds = torch.empty(B, L, device=dev)
    res = []
    for k in KINDS:
        kid = getattr(_C, k.upper())
        fn = lambda: _C.check(lib.ltr_pairwise_loss_f32(kid, 1.0, scores.data_ptr(), rel.data_ptr(), 0, n.data_ptr(), B, L,
                                                        loss.data_ptr(), ds.data_ptr(), torch.cuda.current_stream().cuda_stream))
        for _ in range(5):
            fn()
        t, _ = time_launches(fn, per_graph=20, replays=10)
        wsb = lib.ltr_pairwise_loss_workspace_bytes(kid, B, L)
        if wsb > 0:
            ws = torch.empty(wsb // 4, device=dev)
            fs = lambda: _C.check(lib.ltr_pairwise_loss_ws_f32(kid, 1.0, scores.data_ptr(), rel.data_ptr(), 0, n.data_ptr(), B, L,
                                                               loss.data_ptr(), ds.data_ptr(), ws.data_ptr(), wsb, torch.cuda.current_stream().cuda_stream))
            for _ in range(5):
                fs()
            t2, _ = time_launches(fs, per_graph=20, replays=10)
            res.append("%s %.1f (split %.1f)" % (k, t, t2))
        else:
            res.append("%s %.1f" % (k, t))
    print("B=%d L=%d | " % (B, L) + " | ".join(res) + "  (us)", flush=True)
