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
