"""Stress of the data-parallel lazy step on ONE GPU: WORLD ranks (processes) on cuda:0 over gloo, `steps` back-to-back lazy steps
per rank without a host synchronisation in between, for several shard sizes.  Usage: python scripts/dp_stress.py [world] [steps]
(spawns its ranks).  Prints one line per shard size: device status, weights identical across ranks, us per step."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker():
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    from pytorchltr_amd import _C
    from pytorchltr_amd.distributed import MailboxOverlap
    from pytorchltr_amd.fused import LazySGD
    lib = _C.lib()
    steps = int(os.environ["DP_STEPS"])
    L, F = 128, 136
    for Bs in [int(x) for x in os.environ.get("DP_SHARDS", "64,256,448").split(",")]:
        g = torch.Generator().manual_seed(100 + rank)
        nb = 4
        X = [torch.randn(Bs, L, F, generator=g).to(dev) for _ in range(nb)]
        y = [torch.randint(0, 5, (Bs, L), generator=g).to(dev) for _ in range(nb)]
        n = [torch.randint(1, L + 1, (Bs,), generator=g).to(dev) for _ in range(nb)]
        W = (torch.rand(F, generator=torch.Generator().manual_seed(5)) * 0.1).to(dev)
        b = torch.zeros(1, device=dev)
        mb = MailboxOverlap(F, count=Bs, device=dev)
        assert mb.ok, mb.why
        opt = LazySGD(W, b, 1e-6, loss="hinge", mailbox=mb)
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        err = None
        try:
            for i in range(steps):
                opt.step(X[i % nb], y[i % nb], n[i % nb])
            opt.flush()
        except RuntimeError as exc:
            err = repr(exc)[:120]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        st = int(lib.ltr_device_status(1))
        wb = [None] * world
        dist.all_gather_object(wb, torch.cat([W, b]).cpu().numpy().tobytes())
        if rank == 0:
            print(json.dumps({"world": world, "shard": Bs, "steps": steps, "status": st, "err": err,
                              "weights_identical": all(x == wb[0] for x in wb), "us_per_step": dt / steps * 1e6}), flush=True)
        dist.barrier()
        mb.close()
    dist.destroy_process_group()


def main():
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(world):
        env = dict(os.environ)
        env.update({"RANK": str(r), "WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
                    "HSA_ENABLE_IPC_MODE_LEGACY": "0", "DP_WORKER": "1", "DP_STEPS": str(steps)})
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)], env=env))
    for p in procs:
        p.wait()


if __name__ == "__main__":
    worker() if os.environ.get("DP_WORKER") == "1" else main()
