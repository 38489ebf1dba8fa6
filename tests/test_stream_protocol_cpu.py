"""The streaming schedule's pixel-ring exchange (csrc/er_stream.hip: st_take with compare-and-swap, lap-tagged cells, "put the
pixel back, then take one", retire when nothing can be taken) as a host-thread model under contention: tests/native/ring_model.cpp.
No GPU needed; the device code follows the same steps with LDS atomics."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "native", "ring_model.cpp")


@pytest.fixture(scope="module")
def model(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("ring") / "ring_model")
    subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", SRC, "-o", exe], check=True)
    return exe


@pytest.mark.parametrize("slots,pixels,samples", [(8, 8, 1500), (8, 9, 1000), (6, 40, 300), (3, 64, 100), (8, 5, 1000)])
def test_every_pixel_gets_every_sample_and_the_ring_ends_empty(model, slots, pixels, samples):
    for _ in range(3):
        r = subprocess.run([model, str(slots), str(pixels), str(samples), "0"], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stdout


def test_the_model_sees_the_fault_of_the_first_take(model):
    """Subtract-then-restore lets the count dip below zero while several slots ask at once; a slot that has just put its pixel
    back can then be refused, retire, and leave the pixel in the ring.  The model reproduces it (informative: the outcome
    depends on thread timing, so only its output format is checked)."""
    outs = [subprocess.run([model, "8", "8", "1500", "1"], capture_output=True, text=True, timeout=120) for _ in range(4)]
    assert all("pixels short" in o.stdout for o in outs)
    print("first-version take: runs that lost pixels:", sum(1 for o in outs if o.returncode != 0), "of", len(outs))
