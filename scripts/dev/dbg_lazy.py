import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from tests.conftest import synth
from pytorchltr_amd import _C
lib = _C.lib(); dev = torch.device("cuda:0")
def run(kind, B, L, F, lazy):
    kind_id = getattr(_C, kind.upper()); lr = 0.05
    s, y, n, X, W, b = synth(B, L, 11, F=F)
    Xd, yd, nd = X.to(dev), y.to(dev), n.to(dev)
    Wd, bd = W.clone().to(dev), b.clone().to(dev)
    nws = lib.ltr_linear_workspace_bytes(B, L, F)
    ws = torch.zeros(nws // 4 + 64, device=dev); loss = torch.empty(B, device=dev); bucket = torch.zeros(F + 2, device=dev)
    st = _C.stream_of(Xd)
    if lazy:
        _C.check(lib.ltr_linear_sgd_lazy_step_f32(kind_id, 1.0, Xd.data_ptr(), Wd.data_ptr(), bd.data_ptr(), yd.data_ptr(), _C.LABEL_I64, nd.data_ptr(), B, L, F, lr, loss.data_ptr(), bucket.data_ptr(), ws.data_ptr(), ws.numel()*4, 0, st))
        part0 = ws.clone()
        _C.check(lib.ltr_linear_sgd_lazy_step_f32(kind_id, 1.0, Xd.data_ptr(), Wd.data_ptr(), bd.data_ptr(), yd.data_ptr(), _C.LABEL_I64, nd.data_ptr(), B, L, F, lr, loss.data_ptr(), bucket.data_ptr(), ws.data_ptr(), ws.numel()*4, B, st))
    else:
        _C.check(lib.ltr_linear_sgd_step_f32(kind_id, 1.0, Xd.data_ptr(), Wd.data_ptr(), bd.data_ptr(), yd.data_ptr(), _C.LABEL_I64, nd.data_ptr(), None, B, L, F, lr, loss.data_ptr(), bucket.data_ptr(), ws.data_ptr(), ws.numel()*4, None, st))
        part0 = ws.clone()
    torch.cuda.synchronize()
    return bucket.cpu().numpy(), part0.cpu().numpy()
for shp in [("dcg_hinge", 600, 60, 64), ("hinge", 600, 60, 64), ("hinge", 600, 128, 136), ("hinge", 512, 60, 64), ("dcg_hinge", 300, 60, 64), ("hinge", 1024, 60, 64)]:
    e, pe = run(*shp, False); l, pl = run(*shp, True)
    PF = (shp[3] + 1 + 3)//4*4
    d = np.nonzero(e != l)[0]
    print(shp, "bucket diffs at", d[:10], "of", len(e), "| partial rows equal:", np.array_equal(pe[:shp[1]*PF], pl[:shp[1]*PF]))
