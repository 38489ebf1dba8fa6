"""er_cdf.h (guide table + integer replay) returns exactly what HDRI::binarySearch returns (CPU, host build of the
same inline function the kernel uses, through include/eleven_hip_debug.h)."""
import ctypes as C

import numpy as np
import pytest

from elevenrender_amd import abi, scenes


def reference_search(L, cdf, values, n):
    P = C.POINTER(C.c_float)
    return np.array([L.oracle_hdri_binary_search(cdf.ctypes.data_as(P), float(v), n) for v in values], np.int32)


def fast_search(cdf, values, n):
    lib = abi.load()
    P = C.POINTER(C.c_float)
    lib.er_debug_cdf_search.argtypes = [P, C.c_int, P, C.POINTER(C.c_int32), C.c_int]
    values = np.ascontiguousarray(values, np.float32)
    out = np.empty(values.size, np.int32)
    assert lib.er_debug_cdf_search(cdf.ctypes.data_as(P), n, values.ctypes.data_as(P), out.ctypes.data_as(C.POINTER(C.c_int32)), values.size) == 0
    return out


def make_cdf(L, tex):
    data, w, h, ch, flt = tex
    t = abi.ErTexture(w, h, ch, flt, data.ctypes.data_as(C.POINTER(C.c_float)))
    cdf = np.zeros(w * h + 1, np.float32)
    rs = C.c_float()
    L.oracle_hdri_cdf(C.byref(t), cdf.ctypes.data_as(C.POINTER(C.c_float)), C.byref(rs))
    return cdf, w * h


@pytest.mark.parametrize("kind", ["sky", "noise", "dark_runs", "tiny", "constant"])
def test_fast_cdf_search_equals_reference(oracle_mod, kind):
    L = oracle_mod.lib()
    r = scenes.Rand(77, 0)
    if kind == "sky":
        tex = scenes.sky_hdri(256, 128)
    elif kind == "noise":
        tex = (r.u01(64, 96, 3).astype(np.float32), 96, 64, 3, 0)
    elif kind == "dark_runs":            # long runs of zero-luminance texels: equal consecutive CDF entries
        d = r.u01(64, 64, 3).astype(np.float32)
        d[(r.u01(64, 64) < 0.7)] = 0
        d[10:30] = 0
        tex = (d, 64, 64, 3, 0)
    elif kind == "tiny":
        tex = (np.array([[[0.2, 0.3, 0.1], [1, 2, 3]], [[0, 0, 0], [5, 5, 5]]], np.float32), 2, 2, 3, 0)
    else:
        tex = (np.full((1, 1, 3), 0.5, np.float32), 1, 1, 3, 0)
    cdf, n = make_cdf(L, tex)
    # random values as the RNG produces them (state / 2^32), the CDF entries themselves and their float neighbours
    rs = (r.u64(20000) >> np.uint64(32)).astype(np.uint32)
    vals = [rs.astype(np.float32) / np.float32(4294967296.0), cdf[: n + 1]]
    vals += [np.nextafter(cdf[: n + 1], np.float32(2)), np.nextafter(cdf[: n + 1], np.float32(-1))]
    vals += [np.array([0.0, 1.0, 0.5, 1e-30, 0.99999994], np.float32)]
    values = np.concatenate(vals).astype(np.float32)
    if values.size > 60000:
        values = values[r.u64(60000) % np.uint64(values.size)]
    ref = reference_search(L, cdf, values, n)
    got = fast_search(cdf, values, n)
    bad = np.nonzero(ref != got)[0]
    assert bad.size == 0, (kind, values[bad[:5]], ref[bad[:5]], got[bad[:5]])
