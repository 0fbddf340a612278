#!/usr/bin/env python
"""B-sweep of the two headline kernels at the C2 shape (list_len 128, 136 features): the
B = 1024 batch of BASELINE.json is launch/latency dominated; this shows where the kernels land
when the grid is large enough to be throughput-bound."""
import os, sys, json, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from _benchutil import time_launches
from pytorchltr_amd import _C
dev = torch.device("cuda:0")
lib = _C.lib()
L, F = 128, 136
rows = []
for kind_name, kind in (("hinge", _C.HINGE), ("ndcg2", _C.NDCG2)):
    for full in (False, True):
        for B in (256, 1024, 4096, 16384, 65536, 262144, 1048576):
            g = torch.Generator().manual_seed(0)
            if B > 262144:
                # the loss-only and the metric kernels up to 2^20 queries (no feature tensor: 73 GB at this shape)
                scores = torch.randn(B, L, generator=g).to(dev)
                rel = torch.randint(0, 5, (B, L), generator=g).to(dev)
                n = (torch.full((B,), L) if full else torch.randint(1, L + 1, (B,), generator=g)).to(dev)
                loss = torch.empty(B, device=dev)
                ds = torch.empty(B, L, device=dev)
                cs = lambda: torch.cuda.current_stream().cuda_stream

                def lossk():
                    _C.check(lib.ltr_pairwise_loss_f32(kind, 1.0, scores.data_ptr(), rel.data_ptr(), 0, n.data_ptr(),
                                                       B, L, loss.data_ptr(), ds.data_ptr(), cs()))

                def ndcgk():
                    _C.check(lib.ltr_dcg_f32(scores.data_ptr(), rel.data_ptr(), 0, n.data_ptr(), B, L, 10, 1, 1, loss.data_ptr(), cs()))

                def arpk():
                    _C.check(lib.ltr_arp_f32(scores.data_ptr(), rel.data_ptr(), 0, n.data_ptr(), B, L, loss.data_ptr(), cs()))
                row = dict(kind=kind_name, full_lists=full, B=B)
                for name, fn, alg in (("loss", lossk, B * (16 * L + 16)), ("ndcg10", ndcgk, B * (12 * L + 12)), ("arp", arpk, B * (12 * L + 12))):
                    if name != "loss" and kind_name != "hinge":
                        continue                      # (the metrics do not depend on the loss kind: once)
                    for _ in range(2):
                        fn()
                    t, _ = time_launches(fn, per_graph=2, replays=3)
                    row["%s_us" % name] = t
                    row["%s_qps" % name] = B / t * 1e6
                    row["%s_alg_GBs" % name] = alg / t / 1e3
                rows.append(row)
                print(json.dumps(row), flush=True)
                del scores, rel, ds
                torch.cuda.empty_cache()
                continue
            scores = torch.randn(B, L, generator=g).to(dev)
            rel = torch.randint(0, 5, (B, L), generator=g).to(dev)
            n = (torch.full((B,), L) if full else torch.randint(1, L + 1, (B,), generator=g)).to(dev)
            X = torch.randn(B, L, F, device=dev)
            W = torch.randn(F, device=dev) * 0.1
            bias = torch.zeros(1, device=dev)
            loss = torch.empty(B, device=dev)
            ds = torch.empty(B, L, device=dev)
            part = torch.empty(lib.ltr_linear_workspace_bytes(B, L, F) // 4, device=dev)
            cs = lambda: torch.cuda.current_stream().cuda_stream

            def fused():
                _C.check(lib.ltr_linear_partials_f32(kind, 1.0, X.data_ptr(), W.data_ptr(), bias.data_ptr(),
                                                     rel.data_ptr(), 0, n.data_ptr(), B, L, F, loss.data_ptr(),
                                                     None, part.data_ptr(), cs()))

            def lossk():
                _C.check(lib.ltr_pairwise_loss_f32(kind, 1.0, scores.data_ptr(), rel.data_ptr(), 0, n.data_ptr(),
                                                   B, L, loss.data_ptr(), ds.data_ptr(), cs()))
            reps = 10 if B <= 16384 else 3
            per = 20 if B <= 16384 else 4
            for _ in range(3):
                fused(); lossk()
            tf, _ = time_launches(fused, per_graph=per, replays=reps)
            tl, _ = time_launches(lossk, per_graph=per, replays=reps)
            alg_f = B * (4 * L * F + 8 * L + 12 + 4 * (F + 1))
            real_f = int(n.sum()) * (4 * F + 8) + B * (12 + 4 * (F + 1))
            alg_l = B * (16 * L + 16)
            # round 5: the synchronous-SGD step on this (warm, single) batch, eager wall clock -- as two launches
            # (ltr_linear_sgd_step_f32) and with the update riding in the next launch (ltr_linear_sgd_lazy_step_f32 + one flush;
            # batches of more than one round of workgroups flush first and run the plain launch: the same two launches)
            import time
            bucket = torch.zeros(F + 2, device=dev)
            Wc, bc = W.clone(), bias.clone()
            steps = 400 if B <= 16384 else 40

            def sgd2():
                _C.check(lib.ltr_linear_sgd_step_f32(kind, 1.0, X.data_ptr(), Wc.data_ptr(), bc.data_ptr(), rel.data_ptr(), 0,
                                                     n.data_ptr(), None, B, L, F, 1e-6, loss.data_ptr(), bucket.data_ptr(),
                                                     part.data_ptr(), part.numel() * 4, None, cs()))
            pend = [0]

            def sgdl():
                _C.check(lib.ltr_linear_sgd_lazy_step_f32(kind, 1.0, X.data_ptr(), Wc.data_ptr(), bc.data_ptr(), rel.data_ptr(), 0,
                                                          n.data_ptr(), B, L, F, 1e-6, loss.data_ptr(), bucket.data_ptr(),
                                                          part.data_ptr(), part.numel() * 4, pend[0], cs()))
                pend[0] = B

            def flush():
                _C.check(lib.ltr_linear_sgd_flush_f32(kind, Wc.data_ptr(), bc.data_ptr(), pend[0], L, F, 1e-6, loss.data_ptr(),
                                                      bucket.data_ptr(), part.data_ptr(), cs()))
                pend[0] = 0
            st = {}
            for name, fn, fin in (("two_launch", sgd2, None), ("lazy", sgdl, flush)):
                for _ in range(10):
                    fn()
                if fin:
                    fin()
                torch.cuda.synchronize()
                best = 1e30
                for _ in range(3):
                    t0 = time.perf_counter()
                    for _i in range(steps):
                        fn()
                    if fin:
                        fin()
                    torch.cuda.synchronize()
                    best = min(best, (time.perf_counter() - t0) / steps * 1e6)
                st[name] = best
            rows.append(dict(kind=kind_name, full_lists=full, B=B, fused_us=tf, fused_qps=B / tf * 1e6,
                             fused_alg_GBs=alg_f / tf / 1e3, fused_read_GBs=real_f / tf / 1e3,
                             loss_us=tl, loss_qps=B / tl * 1e6, loss_alg_GBs=alg_l / tl / 1e3,
                             sgd_step_two_launch_us=st["two_launch"], sgd_step_lazy_us=st["lazy"],
                             sgd_step_lazy_qps=B / st["lazy"] * 1e6))
            print(json.dumps(rows[-1]), flush=True)
            del X
