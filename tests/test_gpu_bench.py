"""GPU tier: bench.py's contract -- one JSON line with the fields the driver reads -- and its
N > 1 code path (gradient accumulation + one RCCL all-reduce per `accum` steps) exercised on one
GPU with a forced one-rank process group."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_args, env_extra=None):
    env = dict(os.environ)
    env.update(env_extra or {})
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env["MASTER_PORT"] = str(29850 + os.getpid() % 100)
    pr = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5",
                         "--no-cpu-baseline", "--no-extra"] + extra_args, env=env, cwd=ROOT,
                        stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert pr.returncode == 0, pr.stderr.decode()[-2000:]
    lines = [ln for ln in pr.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, "stdout must carry exactly one line, got %d" % len(lines)
    return json.loads(lines[0])


def test_single_gpu_line_has_the_contract_fields():
    out = _run([])
    assert out["metric"].startswith("queries/sec loss fwd+bwd (B=1024, list_len=128)")
    assert out["n_gpus"] == 1 and out["steps"] == 20 and out["warmup"] == 5
    assert out["unit"] == "queries/s" and out["higher_is_better"] is True and out["dtype"] == "f32"
    assert out["config"]["mode"] == "eager" and out["config"]["batches_in_rotation"] >= 4
    assert out["value"] > 1e6
    r = out["roofline"]
    assert r["bound"] == "hbm" and 0 < r["frac"] < 1 and r["peak"] == 8000.0
    assert r["moved_bytes_per_launch"] < r["padded_formula"]["bytes_per_launch"]
    assert abs(r["achieved"] - r["moved_bytes_per_launch"] / (r["kernel_us_avg"] * 1e-6) / 1e9) < 1.0
    # the timed region is long enough not to depend on --steps
    assert out["ms_per_step"] * out["config"]["steps_timed"] >= 45.0


def test_forced_process_group_accumulates_and_allreduces():
    out = _run(["--accum", "4"], {"LTR_BENCH_FORCE_DIST": "1", "RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
    assert out["config"]["allreduce_every"] == 4
    assert out["config"]["steps_timed"] % 4 == 0
    assert out["value"] > 1e6


def test_shard_mode_splits_the_global_batch():
    out = _run(["--workload", "c4", "--shard"])
    assert out["scaling"] == "strong" and out["config"]["global_batch"] == 256
