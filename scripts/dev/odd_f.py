import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/scripts")
import torch, bench
from pytorchltr_amd import _C
from pytorchltr_amd.fused import linear_loss_step
dev = torch.device("cuda:0")
for B, L, F in [(1024,128,136),(1024,128,137),(1024,128,138),(1024,128,46),(1024,128,48),(1024,128,45),(1024,100,220),(1024,100,221),(256,1000,46),(256,1000,48)]:
    nbuf = bench.nbuf_for(B, L, F)
    bat = []
    for i in range(nbuf):
        g = torch.Generator().manual_seed(i)
        bat.append((torch.randn(B, L, F, generator=g).to(dev), torch.randint(0, 5, (B, L), generator=g).to(dev), torch.randint(1, L + 1, (B,), generator=g).to(dev)))
    W = torch.randn(F, device=dev) * 0.1; b = torch.zeros(1, device=dev)
    h = _C.lib()
    loss = torch.empty(B, device=dev); part = torch.empty(h.ltr_linear_workspace_bytes(B, L, F) // 4 + 64, device=dev)
    def f(i):
        X, y, n = bat[i % nbuf]
        rc = h.ltr_linear_partials_f32(0, 1.0, X.data_ptr(), W.data_ptr(), b.data_ptr(), y.data_ptr(), 0, n.data_ptr(), B, L, F, loss.data_ptr(), None, part.data_ptr(), torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc
    for i in range(3): f(i)
    us, _ = bench.time_launches(f, nbuf, rounds=max(2, 16 // nbuf), replays=10)
    print("%dx%dx%d hinge plan %d: %.2f us" % (B, L, F, h.ltr_linear_fused_plan(0, B, L, F), us), flush=True)
