"""GPU tier: the mailbox all-reduce (include/ltr_hip.h: ltr_mailbox_*) between TWO processes on one GPU -- HIP IPC
maps each process's mailbox into the other, the all-reduce kernels exchange tagged granules through the mapped
memory exactly as they would over xGMI between two GPUs (what a one-GPU box cannot show is the fabric latency and
the cross-device visibility of the fine-grained stores).  Checked: 60 all-reduces of known
vectors with uneven arrival (bit-exact against the rank-order fp32 sum, bit-identical across the ranks), then four
synchronous-SGD steps over two shards against the oracle's trajectory on the whole batch
(examples/01-basic-usage.py:66-75, sharded per SURVEY.md 8(e))."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_workers(world, extra_env=None, timeout=420):
    assert torch.cuda.is_available()
    port = 29600 + (os.getpid() + 7 * world) % 300
    procs = []
    for r in range(world):
        env = dict(os.environ)
        env.update({"RANK": str(r), "WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
                    "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_mailbox_worker.py")], env=env, cwd=ROOT,
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


def _check_full(outs, world):
    if not all(o["ok"] for o in outs):
        pytest.skip("mailbox set-up not possible here: %s" % outs[0]["why"])
    for o in outs:
        assert o["allreduce_mismatches"] == 0, o             # incl. the tag wrap 0xFFFFFFFF -> 2 half way through
        assert o["bit_identical_across_ranks"], o
        assert o["sgd_trajectory_ok"], o
        assert o["weights_identical_across_ranks"], o
        assert o["status"] == 0, o
        # a timed-out all-reduce: LTR_ERR_TIMEOUT (-8 family) in the status, NaN bucket, weights exactly as before
        assert o["timeout_status"] != 0 and o["timeout_bucket_poisoned"] and o["timeout_weights_untouched"], o
        assert o["after_timeout_allreduce_ok"] and o["status_end"] == 0, o


def test_two_processes_on_one_gpu_allreduce_and_sgd():
    _check_full(_run_workers(2), 2)


def test_eight_processes_on_one_gpu_all_peer_slots():
    """the code paths of an 8-GPU node -- eight mailboxes, seven peers stored to and polled in rank
    order, both halves, the tag wrap, uneven arrival -- with all eight processes on cuda:0 (what stays unmeasured is
    the xGMI hop itself)."""
    _check_full(_run_workers(8, timeout=900), 8)


def test_failed_self_check_leaves_the_fallback_usable():
    """MailboxOverlap's self-check failing (here: every wait forced to give up) must clear the sticky
    LTR_ERR_TIMEOUT it provoked -- ok is False on every rank, the status is clean and a plain step runs and is right."""
    outs = _run_workers(2, {"MAILBOX_FORCE_FAIL": "1"})
    for o in outs:
        assert not o["ok"], o
        assert o["status_after_failed_check"] == 0, o
        assert o["fallback_rc"] == 0 and o["fallback_step_ok"], o
