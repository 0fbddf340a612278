"""The lazy step's launch under the profiler:  rocprofv3 --kernel-trace --stats -- python scripts/lazy_trace.py lazy|eager [steps]
`lazy`: N steps of ltr_linear_sgd_lazy_step_f32 on the C2 workload's rotating batches (+ the flush) -- every launch of
linear_regtile2_kernel in the trace is a lazy one; `eager`: the same steps through ltr_linear_sgd_step_f32 (kernel + reduction)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
mode = sys.argv[1] if len(sys.argv) > 1 else "lazy"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
dev = torch.device("cuda:0")
B, L, F, kind = bench.WORKLOADS["c2"]
nbuf = bench.nbuf_for(B, L, F)
bat = bench.make_batches(B, L, F, nbuf, 0, dev)
fs = bench.FusedStep(kind, B, L, F, dev)
for i in range(steps):
    if mode == "lazy":
        fs.lazy_step(bat[i % nbuf])
    else:
        fs.sgd_step(bat[i % nbuf])
if mode == "lazy":
    fs.lazy_flush()
torch.cuda.synchronize()
print(mode, steps, "steps done")
