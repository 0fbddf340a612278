"""A/B of the loss-only and metric kernels between library builds (graph replays):
python scripts/dev/loss_ab.py lib1.so lib2.so -- ndcg2:1024x128 ndcg10:65536x512 arp:4096x1000f ...   (trailing f: full lists)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch
from _benchutil import time_launches
from pytorchltr_amd import _C
dev = torch.device("cuda:0")
sep = sys.argv.index("--")
libs, cases = sys.argv[1:sep], sys.argv[sep + 1:]
handles = []
for path in libs:
    lib = ctypes.CDLL(path)
    for name, (res, argt) in _C.SIGNATURES.items():
        if hasattr(lib, name):
            getattr(lib, name).restype = res; getattr(lib, name).argtypes = argt
    handles.append(lib)
for case in cases:
    kind, shp = case.split(":")
    full = shp.endswith("f")
    B, L = (int(v) for v in shp.rstrip("f").split("x")[:2])
    g = torch.Generator().manual_seed(0)
    scores = torch.randn(B, L, generator=g).to(dev)
    rel = torch.randint(0, 5, (B, L), generator=g).to(dev)
    n = (torch.full((B,), L) if full else torch.randint(1, L + 1, (B,), generator=g)).to(dev)
    out = [case]
    ref = None
    for path, lib in zip(libs, handles):
        loss = torch.empty(B, device=dev); ds = torch.empty(B, L, device=dev)
        if kind == "ndcg10":
            def f():
                rc = lib.ltr_dcg_f32(scores.data_ptr(), rel.data_ptr(), 0, n.data_ptr(), B, L, 10, 1, 1, loss.data_ptr(), torch.cuda.current_stream().cuda_stream)
                assert rc == 0, rc
        elif kind == "arp":
            def f():
                rc = lib.ltr_arp_f32(scores.data_ptr(), rel.data_ptr(), 0, n.data_ptr(), B, L, loss.data_ptr(), torch.cuda.current_stream().cuda_stream)
                assert rc == 0, rc
        else:
            k = getattr(_C, kind.upper())
            def f():
                rc = lib.ltr_pairwise_loss_f32(k, 1.0, scores.data_ptr(), rel.data_ptr(), 0, n.data_ptr(), B, L, loss.data_ptr(), ds.data_ptr(),
                                               torch.cuda.current_stream().cuda_stream)
                assert rc == 0, rc
        ds.zero_()
        for _ in range(3): f()
        reps = 10 if B <= 16384 else 3
        ts = sorted(time_launches(f, per_graph=20 if B <= 16384 else 4, replays=reps)[0] for _ in range(3))
        sig = (float(loss.double().sum()), float(ds.double().abs().sum()))
        if ref is None: ref = sig
        out.append("%s %.2f/%.2f/%.2f%s" % (os.path.basename(path).replace("libltr_", "").replace(".so", ""), *ts, "" if sig == ref else " DIFF"))
    print(" | ".join(out), flush=True)
