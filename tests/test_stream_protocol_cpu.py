"""The streaming schedule's hand-off protocol as a host-thread model under contention and under ThreadSanitizer.

tests/native/ring_model.cpp runs ONE workgroup of csrc/er_stream.hip on CPU threads: ray ring, shade ring, finish ring (with the tracers' short cut for paths
that have left the scene) and the HBM pixel ring with its "entry read" bits -- on the SAME functions the kernel uses (csrc/er_ring.h, compiled with -DER_RING_HOST_MODEL) -- with capacities of 4 to 16 cells, so every ring wraps hundreds to
thousands of times per run, and with the slot records and per-pixel state in plain memory, so that ThreadSanitizer
reports any hand-off the protocol leaves unordered.  This is where the protocol is argued exact; the GPU
suite only keeps one regression run per call pattern.  No GPU needed."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "native", "ring_model.cpp")
HDR = os.path.join(ROOT, "elevenrender_amd", "csrc", "er_ring.h")


def _build(tmp, name, flags, guard_log2=31):
    """guard = polls a ring wait may last before the model calls it a protocol fault.  The checked protocol never needs it, so the
    model's is practically unbounded (2^31 polls, a millisecond of sleep every 8 192: minutes beyond the run's timeout): a wait that never ends is caught
    by the run's own timeout instead, and a thread that is merely off its core for a long while on a busy machine -- which failed
    correct runs twice in round 5 with a guard of 2^25 -- is not a fault.  The negative control needs a short guard: its lost entries
    are waited for until it expires."""
    exe = str(tmp / name)
    subprocess.run(["g++", "-std=c++17", "-pthread", "-DER_RING_HOST_MODEL", f"-DER_RING_GUARD=(1u<<{guard_log2})"] + flags + [SRC, "-o", exe], check=True)
    return exe


@pytest.fixture(scope="module")
def model(tmp_path_factory):
    return _build(tmp_path_factory.mktemp("ring"), "ring_model", ["-O2"])


@pytest.fixture(scope="module")
def model_short_guard(tmp_path_factory):
    return _build(tmp_path_factory.mktemp("ring_neg"), "ring_model_neg", ["-O2"], guard_log2=19)


@pytest.fixture(scope="module")
def model_tsan(tmp_path_factory):
    return _build(tmp_path_factory.mktemp("ring_tsan"), "ring_model_tsan", ["-O1", "-g", "-fsanitize=thread"])


def _run(exe, *args, timeout=300):
    # (ER_MODEL_DEBUG: the model prints its rings' counters once a second -- a run lasts a second or two, so this is a line or two, and
    # if a run ever stalls the assertion's message shows where.  Round 5 saw this file fail twice on a heavily loaded box -- "put guard
    # expired" in the 16-slot / 16-pixel case, once beside a parallel compile job, once in a suite run that took five times its usual
    # time -- and never in ~150 repetitions of the same case alone or beside 12 busy loops: NOTEBOOK.md, round 5.)
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66", ER_MODEL_DEBUG="1")
    env.pop("LD_PRELOAD", None)      # (tools/sanitize_cpu.sh preloads the ASan runtime into python: not into a TSan binary)
    return subprocess.run([exe] + [str(a) for a in args], capture_output=True, text=True, timeout=timeout, env=env)


# slots, pixels, samples, tracer waves, shader waves, log2(ray ring cells)
CONFIGS = [(8, 24, 200, 3, 2, 3), (16, 16, 300, 6, 4, 4), (5, 64, 100, 2, 5, 2), (6, 40, 60, 4, 3, 2), (3, 3, 2000, 3, 3, 2),
           (8, 5, 600, 2, 2, 3), (12, 100, 40, 5, 3, 3)]


@pytest.mark.parametrize("cfg", CONFIGS, ids=lambda c: "x".join(str(v) for v in c))
def test_every_ray_once_every_sample_in_order_every_ring_empty(model, cfg):
    slots, pixels, samples, tracers, shaders, rq_log2 = cfg
    for _ in range(2):
        r = _run(model, slots, pixels, samples, 0, tracers, shaders, rq_log2)
        assert r.returncode == 0, (r.stdout, r.stderr[-2000:])
        assert "0 pixels short" in r.stdout and " 0 lost, 0 protocol errors, 0 ring faults" in r.stdout
        laps = int(r.stdout.split("laps: ray ring ")[1].split(",")[0])
        assert laps >= 50, r.stdout        # the point of the small capacities: positions come round again, many times
        assert int(r.stdout.split("short cuts ")[1]) > 0, r.stdout       # escaped paths did go from the tracers straight to the finish ring


def test_thread_sanitizer_finds_no_unordered_hand_off(model_tsan):
    for cfg in [(8, 24, 60, 3, 2, 3), (6, 40, 30, 4, 3, 2), (4, 4, 300, 3, 3, 2), (16, 16, 100, 6, 4, 4)]:
        slots, pixels, samples, tracers, shaders, rq_log2 = cfg
        r = _run(model_tsan, slots, pixels, samples, 0, tracers, shaders, rq_log2, timeout=600)
        assert "ThreadSanitizer" not in r.stderr, r.stderr[-3000:]
        assert r.returncode == 0, (r.stdout, r.stderr[-2000:])


def test_the_model_has_teeth_round_2_producers_lose_an_entry_in_a_scripted_interleaving(model_short_guard):
    """Negative control, DETERMINISTIC (the threaded version passed or failed with the machine's timing): one thread
    plays the interleaving that broke round 2 -- a reader granted a position stalls before reading, the ring comes round -- on the
    ring functions themselves.  The checked producer of er_ring.h waits for the reader (its guard expires in this script, the entry
    is then still there for the reader); round 2's plain-overwrite producer destroys the unread entry and the reader never finds
    it.  Both outcomes are asserted by the script (exit code 0) and printed."""
    r = _run(model_short_guard, "script", timeout=600)
    assert r.returncode == 0, (r.stdout, r.stderr[-2000:])
    assert "checked producer waited for the reader (guard); the stalled reader then found its entry (payload 100)" in r.stdout
    assert "round 2's producer overwrote the unread cell; the stalled reader never finds its entry (guard)" in r.stdout


def test_round_2_producers_under_threads_are_usually_caught_too(model_short_guard):
    """The same fault under real threads (timing-dependent, so informational: it reports how many of 3 runs the model caught and
    asserts nothing about the count; the deterministic control above is the test)."""
    caught = 0
    for _ in range(3):
        r = _run(model_short_guard, 8, 24, 200, 2, 3, 2, 2, timeout=600)
        assert "pixels short" in r.stdout
        caught += r.returncode != 0
    print("unchecked producers under threads: runs caught by the model:", caught, "of 3")


def test_kernel_and_model_share_the_ring_functions():
    """The kernel must call the functions the model checks, not copies of them."""
    src = open(os.path.join(ROOT, "elevenrender_amd", "csrc", "er_stream.hip")).read()
    assert '#include "er_ring.h"' in src
    for fn in ("er_ring_put", "er_ring_get", "er_ring_grant", "er_ring_reserve", "er_ring_publish", "er_bits_acquire", "er_bits_release"):
        assert fn + "(" in src, fn
        assert fn + "(" in open(SRC).read(), fn
    # nothing clears a ring cell by hand any more (the two late-clear faults of round 2)
    assert "*cell = 0" not in src
    assert os.path.exists(HDR)
