import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from pytorchltr_amd.fused import use_linear_scorer
from pytorchltr_amd.loss import PairwiseHingeLoss
from pytorchltr_amd.optim import SGD
mode = sys.argv[1]
dev = torch.device("cuda:0")
B, L, F, kind = bench.WORKLOADS["c2"]
nbuf = bench.nbuf_for(B, L, F)
bat = bench.make_batches(B, L, F, nbuf, 0, dev)
model = use_linear_scorer(torch.nn.Linear(F, 1).to(dev))
opt = SGD(model.parameters(), lr=1e-6) if mode == "lazy" else torch.optim.SGD(model.parameters(), lr=1e-6)
loss_fn = PairwiseHingeLoss()
def run(n):
    for i in range(n):
        b = bat[i % nbuf]
        loss = loss_fn(model(b["X"]), b["rel"], b["n"]).mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
    torch.cuda.synchronize()
run(200)
import time
t0 = time.perf_counter(); run(2000); print(mode, "us/step", (time.perf_counter() - t0) / 2000 * 1e6)
pr = cProfile.Profile(); pr.enable(); run(2000); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:6000])
