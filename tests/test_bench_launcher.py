"""`python bench.py --gpus N` must START N ranks (round-2 review: the flag was parsed and ignored, so
a driver's scaling run would have been N one-GPU runs reporting n_gpus = 1).

Driven here exactly as the documented command, on CPU: `--standin` swaps the device pipeline for a
CPU stand-in and the backend for gloo; the launcher, the torch.distributed.run rendezvous on
127.0.0.1, bench's own timed loop (gnn/bench_core.run_sharded) and the one gather per step are the
code that runs under RCCL on the GPU node."""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def _run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, str(ROOT / "bench.py")] + args, env=env, cwd=str(ROOT),
                          capture_output=True, text=True, timeout=timeout)


def _json_line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{") and '"n_gpus"' in ln]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_gpus_flag_launches_that_many_ranks():
    r = _run(["--gpus", "2", "--standin", "--steps", "2", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = _json_line(r.stdout)
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["warmup"] == 1
    assert line["data"] == "standin" and line["value"] is None  # can never be read as a measurement
    assert line["config"]["global_batch"] == 2 * 4


def test_single_rank_needs_no_launcher():
    r = _run(["--standin", "--steps", "1", "--warmup", "0"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert _json_line(r.stdout)["n_gpus"] == 1


def test_flag_and_world_size_must_agree():
    r = _run(["--gpus", "2", "--standin"], env_extra={"WORLD_SIZE": "3", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0
    assert "disagree" in (r.stderr + r.stdout)


def test_failing_rank_fails_the_launcher():
    r = _run(["--gpus", "2", "--standin", "--steps", "1", "--warmup", "0"],
             env_extra={"LAPWARM_STANDIN_FAIL_RANK": "1"}, timeout=600)
    assert r.returncode != 0
