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


def _build(tmp, name, flags, guard_ms=120000):
    """guard = how long a ring wait may last before the model calls it a protocol fault.  Round 6: WALL-CLOCK time (steady_clock), not a
    poll count -- a poll count measures the machine's load (round 5 saw two correct runs end in "put guard expired" with a guard of 2^25
    polls beside a compile job, and answered by making the guard practically unbounded, which hid a real stall just as well).  Now the
    guard is finite again (120 s; the checked protocol never needs it) and the first expiry DUMPS the model's state once -- every ring's
    TAIL / COUNT / HEAD and, per thread, its role and the ring position it waits for with the cell's word -- so that a thread that was
    off its core and a wait cycle can be told apart from the run's own output.  The negative controls need a short guard: their lost
    entries are waited for until it expires."""
    exe = str(tmp / name)
    subprocess.run(["g++", "-std=c++17", "-pthread", "-DER_RING_HOST_MODEL", f"-DER_RING_GUARD_MS={guard_ms}"] + flags + [SRC, "-o", exe], check=True)
    return exe


@pytest.fixture(scope="module")
def model(tmp_path_factory):
    return _build(tmp_path_factory.mktemp("ring"), "ring_model", ["-O2"])


@pytest.fixture(scope="module")
def model_short_guard(tmp_path_factory):
    return _build(tmp_path_factory.mktemp("ring_neg"), "ring_model_neg", ["-O2"], guard_ms=500)


@pytest.fixture(scope="module")
def model_tsan(tmp_path_factory):
    return _build(tmp_path_factory.mktemp("ring_tsan"), "ring_model_tsan", ["-O1", "-g", "-fsanitize=thread"])


def _run(exe, *args, timeout=300):
    # (ER_MODEL_DEBUG: the model prints its rings' counters once a second -- a run lasts a second or two, so this is a line or two, and
    # if a run ever stalls the assertion's message shows where; an expired guard prints the state dump.  Round 5 saw this file fail twice
    # on a heavily loaded box -- "put guard expired" in the 16-slot / 16-pixel case -- with one line of text and no state: NOTEBOOK.md.)
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


# slots, pixels, samples, tracer waves, shader waves, log2(ray ring cells): more slots than pixels, so that slots are free for speculative samples
SPEC_CONFIGS = [(16, 6, 300, 3, 3, 4), (24, 10, 200, 4, 3, 4), (12, 5, 400, 3, 3, 3), (8, 24, 200, 3, 2, 3)]


# ... + speculative samples (0 / 1), the keep rule's slack + 1
KEEP_CONFIGS = [(8, 24, 200, 3, 2, 3, 0, 2), (8, 24, 200, 3, 2, 3, 1, 2), (16, 40, 120, 3, 3, 4, 1, 1), (12, 30, 150, 4, 3, 3, 0, 1), (24, 10, 200, 4, 3, 4, 1, 2)]


@pytest.mark.parametrize("cfg", KEEP_CONFIGS, ids=lambda c: "x".join(str(v) for v in c))
def test_pixels_behind_keep_their_slots_every_sample_once_in_order(model, cfg):
    """Round 6: a pixel that is behind the most advanced one goes on in the slot it has instead of queueing in the pixel ring
    (csrc/er_stream.hip s_front).  In the model, with and without speculative samples beside it: every pixel's samples are still
    accumulated once, in order and from the true stream state, every ring ends empty, and samples WERE begun in the slot their pixel had."""
    slots, pixels, samples, tracers, shaders, rq_log2, spec, keep = cfg
    r = _run(model, slots, pixels, samples, 0, tracers, shaders, rq_log2, spec, keep)
    assert r.returncode == 0, (r.stdout, r.stderr[-2000:])
    assert "0 pixels short" in r.stdout and " 0 lost, 0 protocol errors, 0 ring faults" in r.stdout
    assert int(r.stdout.split("samples begun in the slot their pixel had: ")[1].split()[0]) > 0, r.stdout


@pytest.mark.parametrize("cfg", SPEC_CONFIGS, ids=lambda c: "x".join(str(v) for v in c))
def test_speculative_samples_every_sample_once_in_order_from_the_true_state(model, cfg):
    """Round 6: the speculative samples of the small-share forms of the kernel (csrc/er_stream.hip ST_DRAWS_MASK) in the model: a pixel's next
    sample starts in a free slot (a fourth checked ring) from a GUESSED stream state, and is accumulated only after its predecessor and
    only if the predecessor left that state; the verdict word is exchanged by the committing slot and compare-and-swapped by a speculative
    slot that parks, a parked slot comes back through the finish ring, a dropped one falls free.  The model checks at every accumulation
    that the sample started from the pixel's TRUE state and at the end that every pixel's state is the sum of its samples' draws (each
    accumulated once, in order), that every slot ended free exactly once and every ring empty; guesses must have been right AND wrong."""
    slots, pixels, samples, tracers, shaders, rq_log2 = cfg
    r = _run(model, slots, pixels, samples, 0, tracers, shaders, rq_log2, 1)
    assert r.returncode == 0, (r.stdout, r.stderr[-2000:])
    assert "0 pixels short" in r.stdout and " 0 lost, 0 protocol errors, 0 ring faults" in r.stdout
    started, right, wrong = (int(x) for x in r.stdout.split("speculative samples: ")[1].replace(" started, ", " ").replace(" guesses right, ", " ").replace(" wrong", "").split()[:3])
    assert started >= right + wrong
    if slots > pixels:
        assert right > 20 and wrong > 5, r.stdout


def test_thread_sanitizer_finds_no_unordered_hand_off(model_tsan):
    for cfg in [(8, 24, 60, 3, 2, 3, 0), (6, 40, 30, 4, 3, 2, 0), (4, 4, 300, 3, 3, 2, 0), (16, 16, 100, 6, 4, 4, 0),
                (16, 6, 80, 3, 3, 4, 1), (24, 10, 60, 4, 3, 4, 1), (12, 5, 100, 3, 3, 3, 1),      # (these three: with speculative samples)
                (8, 24, 60, 3, 2, 3, 0, 2), (16, 40, 50, 3, 3, 4, 1, 1)]:                         # (the last two: with the keep rule)
        slots, pixels, samples, tracers, shaders, rq_log2, spec = cfg[:7]
        r = _run(model_tsan, slots, pixels, samples, 0, tracers, shaders, rq_log2, spec, *cfg[7:], timeout=600)
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


def test_a_real_wait_cycle_is_refused_at_start_up_and_explains_itself_when_forced(model_short_guard):
    """VERDICT r5 item 4.  The one capacity precondition the protocol has -- a wave's reservation must fit the ray ring, else its later
    puts wait for readers of its OWN unpublished entries -- is checked when the model starts (the kernel: static_assert in
    er_stream.hip).  Forced on purpose (variant 3) it is a genuine cycle: the guard expires and the dump shows its signature --
    entries reserved and not granted with COUNT 0, the producer waiting for the reader of a cell its own entry fills, every consumer
    idle -- which is NOT what a descheduled reader looks like (there the waited cell's reader is a granted position: HEAD has passed it)."""
    r = _run(model_short_guard, 4, 4, 50, 0, 2, 2, 1, timeout=120)
    assert r.returncode == 2 and "refused" in r.stderr, (r.stdout, r.stderr[-2000:])
    r = _run(model_short_guard, 4, 4, 50, 3, 2, 2, 1, timeout=120)
    assert r.returncode == 1, (r.stdout, r.stderr[-2000:])
    assert "==== ring model state dump (put guard expired) ====" in r.stderr
    assert r.stderr.count("==== ring model state dump") == 1                      # once, by the first thread whose guard expires
    assert "ray ring    cap    2  TAIL 4  COUNT 0  HEAD 0  (reserved and not yet granted: 4)" in r.stderr
    assert "main (camera rays) 0: WAITING" in r.stderr and "ray ring position 2 (cell 0, needs lap 1); cell word last seen: lap 0 FULL" in r.stderr
    assert "tracer 0: not waiting on a ring cell" in r.stderr and "shader 1: not waiting on a ring cell" in r.stderr


def test_the_kernel_asserts_its_ring_capacities():
    """The device's producers never wait at all: its ray ring holds every ray its slots can have in flight (three per slot with the
    light extension), its shade and finish rings every slot -- asserted where the capacities are defined."""
    src = open(os.path.join(ROOT, "elevenrender_amd", "csrc", "er_stream.hip")).read()
    assert "(1u << ST_RQ_LOG2) >= 3u * ER_STREAM_SLOTS" in src
    assert "(1u << ST_SQ_LOG2) >= ER_STREAM_SLOTS" in src


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
