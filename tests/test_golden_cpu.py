"""The oracle against the committed regression vectors (tests/golden), CPU only."""
import ctypes as C

import numpy as np
import pytest

from elevenrender_amd import abi
from golden_util import load


def test_rng_streams_golden(oracle_mod):
    z = np.load(__import__("os").path.join(__import__("golden_util").GOLDEN_DIR, "rng_streams.npz"))
    L = oracle_mod.lib()
    for px in range(16):
        st = np.zeros(16, np.uint32)
        va = np.zeros(16, np.float32)
        L.oracle_rng_stream(px, 16, st.ctypes.data_as(C.POINTER(C.c_uint32)), va.ctypes.data_as(C.POINTER(C.c_float)))
        assert (st == z["states"][px]).all() and (va == z["values"][px]).all()


@pytest.mark.parametrize("name", ["cornell_32x32_4spp", "torture_300tri_32x24_4spp"])
def test_oracle_reproduces_golden(oracle_mod, name):
    sc, spp, mb, z = load(name)
    o = oracle_mod.Oracle(sc, math_mode=oracle_mod.MATH_ER, max_bounces=mb)
    o.render(spp)
    for pname, p in abi.PASS_NAMES.items():
        assert (o.read_pass(p).view(np.uint32) == z[f"pass_{pname}"].view(np.uint32)).all(), pname
    assert (o.read_samples() == z["samples"]).all() and (o.read_rng() == z["rng"]).all()
    c = o.counters()
    assert [c["paths"], c["bounce_samples"], c["rays"], c["shaded_hits"], c["hdri_samples"]] == list(z["counters"])
    # the per-bounce trace of the next sample
    tr = z["trace"]
    k = 0
    for px in sorted(set(int(v) for v in tr[:, 0]), key=lambda v: list(tr[:, 0]).index(v)):
        for r in o.trace_pixel(px):
            row = tr[k]
            assert (int(row[0]), int(row[1]), int(row[2]), int(row[3]), int(row[4])) == (px, r.bounce, r.tri, r.shadow_tri, r.opaque)
            assert np.array_equal(np.float32(row[5:8]), np.float32(list(r.position)))
            assert np.array_equal(np.float32(row[14:17]), np.float32(list(r.reduction)), equal_nan=True)
            k += 1
    assert k == len(tr)
    o.close()


def test_golden_brute_force_and_libm_agree(oracle_mod):
    sc, spp, mb, z = load("torture_300tri_32x24_4spp")
    o = oracle_mod.Oracle(sc, math_mode=oracle_mod.MATH_ER, max_bounces=mb, traversal=oracle_mod.TRAV_BRUTE)
    o.render(spp)
    assert (o.read_pass(0).view(np.uint32) == z["pass_beauty"].view(np.uint32)).all()
    o.close()
    o = oracle_mod.Oracle(sc, math_mode=oracle_mod.MATH_LIBM, max_bounces=mb)
    o.render(spp)
    d = np.abs(o.read_pass(0) - z["pass_beauty"])
    assert (d <= 1e-3 + 1e-3 * np.abs(z["pass_beauty"])).all(-1).mean() >= 0.995
    o.close()
