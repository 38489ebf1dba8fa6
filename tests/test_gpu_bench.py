"""bench.py's contract on a small frame: one JSON line with `roofline`, and a failure of the default schedule's timed region
is reported as a DEGRADED measurement (top-level flag + exit code 3), not as a quiet success."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--tris", "20000", "--width", "256", "--height", "192", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-trace-phase"]


def _run(extra_env):
    env = dict(os.environ)
    env.update(extra_env)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, (p.stdout[-2000:], p.stderr[-2000:])
    return p.returncode, json.loads(lines[0])


def test_bench_line_on_a_small_frame():
    rc, line = _run({})
    assert rc == 0
    assert line["degraded"] is False and line["config"]["schedule_fallback"] is None
    assert line["unit"] == "Msamples/s" and line["value"] > 0 and line["dtype"] == "f32"
    r = line["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["traffic"] is None and r["traffic_source"] is None      # counter bytes exist for the full C2 frame only, and say where they are from


def test_a_failed_default_schedule_is_reported_as_degraded_and_exits_nonzero():
    rc, line = _run({"ER_BENCH_SIMULATE_STREAM_FAILURE": "1"})
    assert rc == 3
    assert line["degraded"] is True
    assert "simulated failure" in line["config"]["schedule_fallback"]
    assert line["config"]["schedule"] == "wavefront" and line["value"] > 0


def test_bench_gpus_2_without_a_launcher_runs_two_ranks():
    """`python3 bench.py --gpus 2` as the driver types it: the file starts its own two ranks (child process), each renders its
    tiles, the planes are gathered to rank 0.  On this one-GPU box the two ranks share GPU 0 and talk over gloo
    (ER_BENCH_REHEARSAL=1; RCCL refuses two ranks on one device) -- the control flow, the sharding and the line are the real ones."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["ER_BENCH_REHEARSAL"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SMALL, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert p.returncode == 0 and len(lines) == 1, (p.stdout[-2000:], p.stderr[-2000:])
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2
    assert len(line["ranks"]["kernel_ms"]) == 2 and len(line["ranks"]["paths"]) == 2
    assert sum(line["ranks"]["paths"]) == 256 * 192 * 3           # the two shares make the frame, once
    assert line["gather"] and "gloo" in line["gather"]
    assert line["config"]["sharding"].endswith("% 2") and line["scaling"] == "strong"


def test_a_gather_that_does_not_return_costs_the_run_its_exit_code_not_its_measurement():
    """The framebuffer gather of a multi-rank bench runs under a deadline (RCCL from the library's C++ side has never run with more than
    one rank where this was built): with the deadline set to nothing the line is still printed, with `gather` saying that the combine was
    not completed, and the process leaves with code 5."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(ER_BENCH_REHEARSAL="1", ER_BENCH_GATHER_TIMEOUT="0.000001", ER_BENCH_EXIT_GRACE="8")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SMALL, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert p.returncode != 0 and len(lines) == 1, (p.returncode, p.stdout[-2000:], p.stderr[-2000:])
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["value"] > 0 and "NOT COMPLETED" in line["gather"]
