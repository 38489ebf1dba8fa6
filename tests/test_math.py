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


def _neighbourhood(center, n=4096):
    """the n floats on either side of `center` (and center itself)"""
    c = np.float32(center)
    up = [c]
    dn = [c]
    for _ in range(n):
        up.append(np.nextafter(up[-1], np.float32(np.inf)))
        dn.append(np.nextafter(dn[-1], np.float32(-np.inf)))
    return np.array(dn[::-1] + up[1:], np.float32)


def test_er_math_dense_over_the_argument_ranges_of_the_path(oracle_mod):
    """Every er_math.h function over the argument range the per-sample path feeds it (reference call sites in brackets),
    4 M evenly spaced arguments plus the 8 k floats around every edge of the range; bar: 1 ulp of glibc everywhere.
      sin, cos   camera rotation in radians and the spherical mappings' angles [src/kernel.cpp:371-473, src/Texture.cpp:280-292]:
                 |x| <= 4 pi (rotations of up to +-720 degrees), edges 0, +-pi/2, +-pi, +-2 pi
      acos       acos(-y) of a unit direction [src/Texture.cpp:239-251]: [-1, 1], edges -1, 0, 1 (and just outside: NaN on both sides)
      atan2      atan2(-z, x) of a unit direction [src/Texture.cpp:239-251]: [-1, 1]^2 incl. the axes and signed zeros
      pow        pow(roughness, 2.2), pow(metallic, 2.2) [src/kernel.cpp:160-161]: base in [0, 1] (texture values may exceed 1: up to 16), exponent 2.2
      log        GTR1's log(a * a), a = lerp(0.1, 0.001, clearcoatGloss) [src/Disney.cpp:44-52]: [1e-6, 1e-2]; wider: [1e-30, 1e4]"""
    L = oracle_mod.lib()
    pi = np.pi
    n = 1 << 22

    def check(name, kind, x, y=None):
        libm, er = run(L, kind, x, y)
        d = ulp_diff(libm, er)
        both_nan = np.isnan(libm) & np.isnan(er)
        assert (np.isnan(libm) == np.isnan(er)).all(), f"{name}: NaN sets differ"
        d[both_nan] = 0
        assert d.max() <= 1, f"{name}: {d.max()} ulp at x={np.asarray(x)[d.argmax()]!r}"
        return float((d > 0).mean())

    lin = lambda a, b: np.linspace(a, b, n, dtype=np.float64).astype(np.float32)
    edges_trig = np.concatenate([_neighbourhood(v) for v in (0.0, pi / 2, -pi / 2, pi, -pi, 2 * pi, -2 * pi, 4 * pi)])
    shares = {}
    shares["sin"] = max(check("sin", 0, lin(-4 * pi, 4 * pi)), check("sin edges", 0, edges_trig))
    shares["cos"] = max(check("cos", 1, lin(-4 * pi, 4 * pi)), check("cos edges", 1, edges_trig))
    shares["acos"] = max(check("acos", 2, lin(-1, 1)), check("acos edges", 2, np.concatenate([_neighbourhood(v) for v in (-1.0, 0.0, 1.0)])))
    r = scenes.Rand(77, 0)
    ax, ay = r.uniform(-1, 1, n), r.uniform(-1, 1, n)
    shares["atan2"] = check("atan2", 5, ay, ax)
    zeros = np.array([0.0, -0.0, 1.0, -1.0, 1e-30, -1e-30, 0.5, -0.5], np.float32)
    gy, gx = np.meshgrid(zeros, zeros)
    check("atan2 axes", 5, gy.ravel(), gx.ravel())
    e22 = lambda x: np.full(len(x), 2.2, np.float32)
    xs = lin(0, 1)
    shares["pow2.2"] = max(check("pow 2.2 on [0,1]", 4, xs, e22(xs)), check("pow 2.2 on [1,16]", 4, lin(1, 16), e22(xs)))
    pe = np.concatenate([_neighbourhood(0.0)[4096:], _neighbourhood(1.0), np.array([1e-45, 1e-38, 1e-30, 1e-20], np.float32)])
    check("pow 2.2 edges", 4, pe, e22(pe))
    shares["log"] = max(check("log GTR1", 3, lin(1e-6, 1e-2)), check("log wide", 3, np.exp(np.linspace(-69, 9.2, n)).astype(np.float32)),
                        check("log edges", 3, np.concatenate([_neighbourhood(1.0), _neighbourhood(0.0)[4097:]])))
    print("share of arguments where er_math and glibc differ (by 1 ulp):", {k: round(v, 5) for k, v in shares.items()})
