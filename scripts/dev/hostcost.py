"""Host cost (us per call) of the HIP / RCCL calls a per-step overlapped all-reduce is made of."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29611")
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from pytorchltr_amd.distributed import RcclOverlap
ov = RcclOverlap(136, count=1024, device=dev)
print("raw ok", ov.ok, ov.why)
hip = ctypes.CDLL("libamdhip64.so")
rccl = ov.rccl
s1 = torch.cuda.Stream(); s2 = torch.cuda.Stream()
ev = ctypes.c_void_p()
hip.hipEventCreateWithFlags(ctypes.byref(ev), 2)
buf = torch.zeros(256, device=dev)
N = 2000
def t(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(N): fn()
    dt = (time.perf_counter() - t0) / N * 1e6
    torch.cuda.synchronize(); return round(dt, 2)
st1, st2 = ctypes.c_void_p(s1.cuda_stream), ctypes.c_void_p(s2.cuda_stream)
hip.hipEventRecord.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
hip.hipStreamWaitEvent.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint]
print("hipEventRecord", t(lambda: hip.hipEventRecord(ev, st1)))
print("hipStreamWaitEvent (other stream)", t(lambda: hip.hipStreamWaitEvent(st2, ev, 0)))
print("record+wait pair", t(lambda: (hip.hipEventRecord(ev, st1), hip.hipStreamWaitEvent(st2, ev, 0))))
print("ncclAllReduce same stream", t(lambda: rccl.ncclAllReduce(buf.data_ptr(), buf.data_ptr(), 138, 7, 0, ov.comm, st1)))
x = torch.zeros(1024, device=dev)
print("tiny torch kernel launch (x.add_)", t(lambda: x.add_(1.0)))
print("c10d all_reduce blocking", t(lambda: dist.all_reduce(buf)))
ov.close(); dist.destroy_process_group()
