"""er_math.h (the transcendental functions shared by the kernel and the oracle's er mode) against glibc libm."""
import ctypes as C

import numpy as np

from elevenrender_amd import scenes


def ulp_diff(a, b):
    ia = a.view(np.int32).astype(np.int64)
    ib = b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7FFFFFFF), ia)
    ib = np.where(ib < 0, -(ib & 0x7FFFFFFF), ib)
    d = np.abs(ia - ib)
    d[np.isnan(a) & np.isnan(b)] = 0
    return d


def run(L, kind, x, y=None):
    x = np.ascontiguousarray(x, np.float32)
    y = np.ascontiguousarray(y if y is not None else np.zeros_like(x), np.float32)
    out = [np.empty_like(x), np.empty_like(x)]
    P = C.POINTER(C.c_float)
    for mode in (0, 1):
        L.oracle_math(kind, mode, x.ctypes.data_as(P), y.ctypes.data_as(P), out[mode].ctypes.data_as(P), x.size)
    return out


def test_er_math_within_one_ulp_of_libm(oracle_mod):
    L = oracle_mod.lib()
    r = scenes.Rand(2024, 0)
    n = 200000
    cases = {
        "sin": (0, r.uniform(-7, 7, n), None), "sin_wide": (0, r.uniform(-2000, 2000, n), None),
        "cos": (1, r.uniform(-7, 7, n), None), "cos_wide": (1, r.uniform(-2000, 2000, n), None),
        "acos": (2, r.uniform(-1, 1, n), None),
        "log": (3, np.exp(r.uniform(-20, 20, n)).astype(np.float32), None),
        "pow2.2": (4, r.uniform(0, 1, n), np.full(n, 2.2, np.float32)),
        "pow2.2_big": (4, r.uniform(1, 50, n), np.full(n, 2.2, np.float32)),
        "atan2": (5, r.uniform(-2, 2, n), r.uniform(-2, 2, n)),
    }
    for name, (kind, x, y) in cases.items():
        libm, er = run(L, kind, x, y)
        d = ulp_diff(libm, er)
        assert d.max() <= 1, f"{name}: {d.max()} ulp at x={x[d.argmax()]}"
        if name in ("pow2.2", "log"):
            assert (d > 0).mean() < 0.02     # glibc's powf/logf are nearly correctly rounded: almost always identical


def test_er_math_special_values(oracle_mod):
    L = oracle_mod.lib()
    inf, nan = np.float32(np.inf), np.float32(np.nan)
    libm, er = run(L, 0, [0.0, -0.0, nan, inf])
    assert er[0] == 0 and er[1] == 0 and np.isnan(er[2]) and np.isnan(er[3])
    libm, er = run(L, 1, [0.0, nan, inf])
    assert er[0] == 1 and np.isnan(er[1]) and np.isnan(er[2])
    libm, er = run(L, 2, [1.0, -1.0, 0.0, 1.5, nan])
    assert er[0] == 0 and er[1] == np.float32(np.pi) and er[2] == np.float32(np.pi / 2) and np.isnan(er[3]) and np.isnan(er[4])
    assert (libm[:3] == er[:3]).all()
    libm, er = run(L, 3, [1.0, 0.0, -1.0, inf, 1e-40])
    assert er[0] == 0 and er[1] == -inf and np.isnan(er[2]) and er[3] == inf and er[4] == libm[4]
    libm, er = run(L, 4, [0.0, 1.0, -0.5, inf, 1e-30, 1e-20], [2.2] * 6)
    assert er[0] == 0 and er[1] == 1 and np.isnan(er[2]) and er[3] == inf and (er[4:] == libm[4:]).all()
    libm, er = run(L, 5, [0.0, -0.0, 0.0, 1.0, -1.0], [-1.0, -1.0, 0.0, 0.0, 0.0])   # atan2(y, x)
    assert (libm == er).all()
