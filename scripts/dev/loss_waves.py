"""The loss-only kernel with 1 / 2 / 4 / 8 waves per query (symmetric pass, ltr_pairwise_loss_f32_cfg) over the batch size:
python scripts/dev/loss_waves.py hinge:128 ndcg2:128 hinge:256 ...   (kind:list_len; ragged lists)"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch
from _benchutil import time_launches
from pytorchltr_amd import _C
dev = torch.device("cuda:0")
lib = _C.lib()
for case in sys.argv[1:]:
    kind, L = case.split(":"); L = int(L)
    k = getattr(_C, kind.upper())
    for B in [int(b) for b in os.environ.get("BS", "1024,4096,16384,65536,262144").split(",")]:
        g = torch.Generator().manual_seed(0)
        scores = torch.randn(B, L, generator=g).to(dev)
        rel = torch.randint(0, 5, (B, L), generator=g).to(dev)
        n = torch.randint(1, L + 1, (B,), generator=g).to(dev)
        loss = torch.empty(B, device=dev); ds = torch.empty(B, L, device=dev)
        row = dict(kind=kind, L=L, B=B)
        ref = None
        for waves in (0, 1, 2, 4, 8, 16):
            def f():
                if waves == 0:
                    rc = lib.ltr_pairwise_loss_f32(k, 1.0, scores.data_ptr(), rel.data_ptr(), 0, n.data_ptr(), B, L, loss.data_ptr(), ds.data_ptr(),
                                                   torch.cuda.current_stream().cuda_stream)
                else:
                    rc = lib.ltr_pairwise_loss_f32_cfg(k, 1.0, scores.data_ptr(), rel.data_ptr(), 0, n.data_ptr(), B, L, loss.data_ptr(), ds.data_ptr(),
                                                       64, 0, waves, torch.cuda.current_stream().cuda_stream)
                return rc
            if f() != 0:
                continue
            for _ in range(2): f()
            t = min(time_launches(f, per_graph=10 if B <= 16384 else 3, replays=5 if B <= 16384 else 3)[0] for _ in range(2))
            sig = (float(loss.double().sum()), float(ds.double().abs().sum()))
            if ref is None: ref = sig
            row["default" if waves == 0 else "w%d" % waves] = round(t, 2)
            if abs(sig[0] - ref[0]) > 1e-6 * abs(ref[0]) or abs(sig[1] - ref[1]) > 1e-5 * abs(ref[1]): row["w%d_DIFF" % waves] = sig
        print(json.dumps(row), flush=True)
