"""A/B of the fused scorer + loss kernel between library builds, cold rotation:
python scripts/lib_ab.py lib1.so lib2.so -- hinge:32x1000x220 dcg_hinge:256x1000x220 ..."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
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
    B, L, F = (int(v) for v in shp.split("x"))
    nbuf = bench.nbuf_for(B, L, F)
    bat = bench.make_batches(B, L, F, nbuf, 0, dev)
    g = torch.Generator().manual_seed(1)
    W = ((torch.rand(F, generator=g) * 2 - 1) / F ** 0.5).to(dev); bias = torch.zeros(1, device=dev)
    out = [case]
    ref = None
    for path, lib in zip(libs, handles):
        loss = torch.empty(B, device=dev)
        part = torch.empty(lib.ltr_linear_workspace_bytes(B, L, F) // 4 + 64, device=dev)
        k = getattr(_C, kind.upper())
        def f(i):
            b = bat[i % nbuf]
            rc = lib.ltr_linear_partials_f32(k, 1.0, b["X"].data_ptr(), W.data_ptr(), bias.data_ptr(), b["rel"].data_ptr(), 0, b["n"].data_ptr(),
                                             B, L, F, loss.data_ptr(), None, part.data_ptr(), torch.cuda.current_stream().cuda_stream)
            assert rc == 0, rc
        for i in range(4): f(i)
        ts = [bench.time_launches(f, nbuf, rounds=max(2, 16 // nbuf), replays=10)[0] for _ in range(3)]
        f(0); torch.cuda.synchronize()
        PF = (F + 1 + 3) // 4 * 4
        sig = (float(loss.double().sum()), float(part[:B * PF].double().abs().sum()))
        if ref is None: ref = sig
        same = abs(sig[0] - ref[0]) <= 1e-6 * abs(ref[0]) and abs(sig[1] - ref[1]) <= 1e-5 * abs(ref[1])
        out.append("%s %.2f/%.2f/%.2f%s" % (os.path.basename(path).replace("libltr_", "").replace(".so", ""), *sorted(ts), "" if same else " MISMATCH %r vs %r" % (sig, ref)))
    print(" | ".join(out), flush=True)
