"""The reference's loop body under the profiler:  rocprofv3 --kernel-trace --stats -- python scripts/dropin_loop.py lazy|torch [steps]
`loss = loss_fn(model(xs), ys, n).mean(); optimizer.zero_grad(); loss.backward(); optimizer.step()` (examples/01-basic-usage.py:70-75)
on the C2 workload's rotating batches, model = use_linear_scorer(nn.Linear(136, 1)), optimizer = pytorchltr_amd.optim.SGD (`lazy`)
or torch.optim.SGD (`torch`): the kernel trace shows the launches per step of either."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pytorchltr_amd.fused import use_linear_scorer
from pytorchltr_amd.loss import PairwiseHingeLoss
from pytorchltr_amd.optim import SGD
mode = sys.argv[1] if len(sys.argv) > 1 else "lazy"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
dev = torch.device("cuda:0")
B, L, F, kind = bench.WORKLOADS["c2"]
nbuf = bench.nbuf_for(B, L, F)
bat = bench.make_batches(B, L, F, nbuf, 0, dev)
model = use_linear_scorer(torch.nn.Linear(F, 1).to(dev))
opt = SGD(model.parameters(), lr=1e-6) if mode == "lazy" else torch.optim.SGD(model.parameters(), lr=1e-6)
loss_fn = PairwiseHingeLoss()
for i in range(steps):
    b = bat[i % nbuf]
    loss = loss_fn(model(b["X"]), b["rel"], b["n"]).mean()
    opt.zero_grad()
    loss.backward()
    opt.step()
torch.cuda.synchronize()
print(mode, steps, "steps done; final |w|", float(model.weight.detach().abs().sum()))
