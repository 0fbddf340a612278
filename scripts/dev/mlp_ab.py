"""A/B of the fused MLP step between library builds, cold rotation (bench.mlp_extra's launch): python scripts/dev/mlp_ab.py lib1.so lib2.so"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from pytorchltr_amd import _C
dev = torch.device("cuda:0")
B, L, F, H1, H2 = 1024, 128, 136, 50, 10
nbuf = bench.nbuf_for(B, L, F)
bat = bench.make_batches(B, L, F, nbuf, 0, dev)
g = torch.Generator().manual_seed(0)
params = [(torch.rand(*s, generator=g) - 0.5).to(dev) for s in ((H1, F), (H1,), (H2, H1), (H2,), (1, H2), (1,))]
for rep in range(2):
    for path in sys.argv[1:]:
        lib = ctypes.CDLL(path)
        for name, (res, argt) in _C.SIGNATURES.items():
            if hasattr(lib, name):
                getattr(lib, name).restype = res; getattr(lib, name).argtypes = argt
        P = lib.ltr_mlp_param_count(F, H1, H2); wsb = lib.ltr_mlp_workspace_bytes(B, F, H1, H2)
        ws = torch.empty(wsb // 4, device=dev); grads = torch.empty(P, device=dev); loss = torch.empty(B, device=dev); ls = torch.zeros(1, device=dev)
        def launch(i):
            b = bat[i % nbuf]
            rc = lib.ltr_mlp_pairwise_f32(0, 1.0, b["X"].data_ptr(), *[p.data_ptr() for p in params], b["rel"].data_ptr(), 0, b["n"].data_ptr(), None,
                                          B, L, F, H1, H2, loss.data_ptr(), None, grads.data_ptr(), ls.data_ptr(), ws.data_ptr(), wsb, torch.cuda.current_stream().cuda_stream)
            assert rc == 0, rc
        for i in range(nbuf): launch(i)
        us, _ = bench.time_launches(launch, nbuf, rounds=4, replays=10)
        print("%-50s %.2f us/step  (loss sum %.6g, |grads| %.6g)" % (os.path.basename(path), us, float(loss.double().sum()), float(grads.double().abs().sum())), flush=True)
