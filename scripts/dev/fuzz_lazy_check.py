import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pytorchltr_amd import _C
dev = torch.device("cuda:0"); lib = _C.lib()
for kind, B, L, F, pat in (("ndcg1", 1025, 128, 220, "full"), ("hinge", 1500, 64, 136, "short"), ("ndcg2", 1025, 160, 220, "zero_some"), ("hinge", 5000, 128, 64, "uniform")):
    g = torch.Generator().manual_seed(5)
    bat = []
    for i in range(2):
        X = torch.randn(B, L, F, generator=g); y = torch.randint(0, 5, (B, L), generator=g)
        n = {"uniform": torch.randint(1, L + 1, (B,), generator=g), "full": torch.full((B,), L), "short": torch.randint(1, min(L, 8) + 1, (B,), generator=g), "zero_some": torch.randint(0, L + 1, (B,), generator=g)}[pat]
        bat.append([t.to(dev) for t in (X, y, n)])
    W0 = (torch.rand(F, generator=g) - 0.5) / F ** 0.5; b0 = torch.zeros(1)
    kid = getattr(_C, kind.upper()); lr = 0.01; st = _C.stream_of(bat[0][0]); nws = lib.ltr_linear_workspace_bytes(B, L, F)
    def run(lazy):
        Wd, bd = W0.clone().to(dev), b0.clone().to(dev)
        ws = torch.full((nws // 4 + 64,), float("nan"), device=dev); loss = torch.empty(B, device=dev); bucket = torch.zeros(F + 2, device=dev)
        out = []; pending = 0
        for k in range(4):
            Xd, yd, nd = bat[k % 2]
            if lazy:
                _C.check(lib.ltr_linear_sgd_lazy_step_f32(kid, 1.0, Xd.data_ptr(), Wd.data_ptr(), bd.data_ptr(), yd.data_ptr(), _C.LABEL_I64, nd.data_ptr(), B, L, F, lr, loss.data_ptr(), bucket.data_ptr(), ws.data_ptr(), ws.numel() * 4, pending, st)); pending = B
                _C.check(lib.ltr_linear_sgd_flush_f32(kid, Wd.data_ptr(), bd.data_ptr(), pending, L, F, lr, loss.data_ptr(), bucket.data_ptr(), ws.data_ptr(), st)); pending = 0
            else:
                _C.check(lib.ltr_linear_sgd_step_f32(kid, 1.0, Xd.data_ptr(), Wd.data_ptr(), bd.data_ptr(), yd.data_ptr(), _C.LABEL_I64, nd.data_ptr(), None, B, L, F, lr, loss.data_ptr(), bucket.data_ptr(), ws.data_ptr(), ws.numel() * 4, None, st))
            torch.cuda.synchronize(); out.append((Wd.cpu().numpy().copy(), bucket.cpu().numpy().copy(), loss.cpu().numpy().copy()))
        return out
    e, l = run(False), run(True)
    for k in range(4):
        print(kind, B, L, F, pat, 'step', k, 'W rel', np.abs(e[k][0]-l[k][0]).max()/np.abs(e[k][0]).max(), 'bucket rel', np.abs(e[k][1]-l[k][1]).max()/np.abs(e[k][1]).max(), 'loss maxdiff', np.abs(e[k][2]-l[k][2]).max(), 'loss max', np.abs(e[k][2]).max())
