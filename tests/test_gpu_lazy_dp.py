"""GPU tier: the data-parallel LAZY step (include/ltr_hip.h: ltr_linear_sgd_lazy_step_dp_f32 / ltr_linear_sgd_flush_dp_f32) --
ONE launch per step at every number of ranks: the reducer workgroups in front of the fused launch all-reduce their column sums
through the peers' mailboxes (HIP IPC), the first query's workgroup writes the weights.  Several processes on cuda:0 (what a
one-GPU box cannot show is the xGMI hop): weights bit-identical to the eager three-launch mailbox step,
bit-identical across the ranks, the oracle's gradient on the concatenated batches at every step, the tag wrap of the lazy half, a forced
give-up that leaves W / bias untouched as a whole (examples/01-basic-usage.py:66-75, sharded per SURVEY.md 8(e))."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_workers(world, extra_env=None, timeout=600):
    assert torch.cuda.is_available()
    port = 29100 + (os.getpid() + 11 * world) % 180
    procs = []
    for r in range(world):
        env = dict(os.environ)
        env.update({"RANK": str(r), "WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
                    "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_lazy_dp_worker.py")], env=env, cwd=ROOT,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    outs = []
    for p in procs:
        try:
            so, se = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, se.decode()[-3000:]
        outs.append(json.loads([ln for ln in so.decode().splitlines() if ln.startswith("{")][-1]))
    return outs


def _check(outs):
    if not all(o["ok"] for o in outs):
        pytest.skip("mailbox set-up not possible here: %s" % outs[0]["why"])
    for o in outs:
        bad = [k for k in ("lazy_equals_eager_bitwise", "last_bucket_equals_eager_bitwise", "flushed_steps_equal_eager_bitwise",
                           "oracle_steps_ok", "weights_identical_across_ranks", "timeout_weights_untouched",
                           "timeout_flush_weights_untouched", "after_timeout_run_equals_eager") if not o[k]]
        bad += [k for k in ("status", "status_end") if o[k] != 0]
        bad += [k for k in ("timeout_status", "timeout_flush_status") if o[k] == 0]    # LTR_ERR_TIMEOUT must have been raised
        assert not bad, (bad, {k: o[k] for k in ("rank", "per_step_grad_maxdiff")})


def test_two_ranks_one_launch_per_step():
    _check(_run_workers(2))


def test_two_ranks_tag_wrap_of_the_lazy_half():
    _check(_run_workers(2, {"LAZY_DP_WRAP": "1"}))


def test_eight_ranks_one_launch_per_step():
    _check(_run_workers(8, timeout=900))


def test_two_ranks_ndcg2_on_the_256_thread_tile():
    """LambdaNDCG2 takes the 256-thread register tile: the reducers and the writer workgroup run with that shape too."""
    _check(_run_workers(2, {"LAZY_DP_KIND": "ndcg2", "LAZY_DP_B": "64"}))


def test_two_ranks_rows_by_query_off_the_register_tile():
    """F = 45 (rows not whole float4): no lazy layout -- every step flushes through the reduction launch and the plain
    mailbox all-reduce, the same weights as the eager step."""
    _check(_run_workers(2, {"LAZY_DP_F": "45", "LAZY_DP_B": "32", "LAZY_DP_L": "40"}))
