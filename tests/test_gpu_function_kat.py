"""SURVEY.md section 4, level 1 on the GPU: every DEVICE function of the path (er_debug_eval, include/eleven_hip_debug.h)
against the oracle's function-level entry points on the same inputs, bit for bit -- a1 RNG, a3 camera ray (bokeh +
rotation), a5 the Tri::hit record (position, shading / geometric normal, tangent, bitangent, uv), a6-a8 texture fetch
(wrap, channels, bilinear) and the spherical mappings, a9 the HDRI search / pdf, a10-a12 Disney sample / eval / pdf, and
the six transcendentals.  (The traversal a4 and the whole bounce a14 are in test_gpu_function_level.py.)  Until round 2
these functions were pinned on the GPU only through image bit-exactness, which localises nothing when it breaks."""
import ctypes as C
import os

import numpy as np
import pytest

from elevenrender_amd import abi, render, scenes

pytestmark = pytest.mark.gpu

FN = dict(RNG=0, CAMERA_RAY=1, TRI_HIT=2, DISNEY_EVAL=3, DISNEY_PDF=4, DISNEY_SAMPLE=5, SPHERICAL=6, REV_SPHERICAL=7,
          TEXTURE=8, HDRI_SEARCH=9, HDRI_PDF=10, MATH=11)


def ibits(v):
    return np.asarray(v, np.int32).view(np.float32)


def same(a, b):
    a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    return (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))


@pytest.fixture(scope="module")
def rig(oracle_mod):
    sc = scenes.torture(600, 48, 36, seed=9, n_materials=4, tex_size=16, hdri_size=(64, 32), smooth=True, n_lights=0)
    cam = sc.camera
    cam.bokeh, cam.aperture, cam.focus_distance = 1, 1.8, 3.0
    cam.rotation = abi.ErVec3(7.0, -11.0, 4.0)
    # a 2-channel and a 1-channel BILINEAR texture next to the 3-channel NO_FILTER ones
    r = np.random.default_rng(4)
    sc.textures.append((abi._f32(r.random((8, 8, 2))), 8, 8, 2, 1))
    sc.textures.append((abi._f32(r.random((5, 7, 1))), 7, 5, 1, 1))
    sc._desc = None
    # (the library keeps a texture that materials use for scalar channels only with its first channel alone -- csrc/er_api.cpp; these
    # are known-answer tests of the FETCH on the textures as they came, so that is switched off for this scene:
    # tests/test_gpu_parity.py::test_scalar_only_textures_kept_with_one_channel_do_not_change_the_image covers the compaction)
    before = os.environ.get("ER_TEX_COMPACT")
    os.environ["ER_TEX_COMPACT"] = "0"
    try:
        rm = render.RenderingManager(render.RenderParameters(max_bounces=8))
        rm.start_rendering(sc)
    finally:
        if before is None:
            del os.environ["ER_TEX_COMPACT"]
        else:
            os.environ["ER_TEX_COMPACT"] = before
    yield sc, rm, oracle_mod
    rm.close()


def test_rng_streams(rig):
    sc, rm, orc = rig
    idx = np.array([0, 1, 2, 47, 48 * 36 - 1, 123456, 2**31 - 2], np.int64)
    out = rm.debug_eval(FN["RNG"], ibits(idx.astype(np.int32)).reshape(-1, 1), 32)
    for k, i in enumerate(idx):
        st = (C.c_uint32 * 16)()
        va = (C.c_float * 16)()
        orc.lib().oracle_rng_stream(int(i), 16, st, va)
        assert same(out[k, :16], np.array(va[:], np.float32)).all()
        assert (out[k, 16:].view(np.uint32) == np.array(st[:], np.uint32)).all()


def test_camera_rays_with_bokeh_and_rotation(rig):
    sc, rm, orc = rig
    r = np.random.default_rng(1)
    n = 500
    items = np.concatenate([r.integers(0, [sc.x_res, sc.y_res], (n, 2)).astype(np.float32), r.random((n, 5)).astype(np.float32)], 1)
    items[0, 2:] = 1.0                                           # next() can return exactly 1.0
    out = rm.debug_eval(FN["CAMERA_RAY"], items, 6)
    L = orc.lib()
    for k in range(n):
        o, d = (C.c_float * 3)(), (C.c_float * 3)()
        rr = (C.c_float * 5)(*items[k, 2:].tolist())
        L.oracle_camera_ray(C.byref(sc.camera), sc.x_res, sc.y_res, int(items[k, 0]), int(items[k, 1]), rr, orc.MATH_ER, o, d)
        assert same(out[k], np.array(list(o) + list(d), np.float32)).all(), k


def test_tri_hit_records(rig):
    sc, rm, orc = rig
    r = np.random.default_rng(2)
    n = 3000
    tri = r.integers(0, sc.tri_count, n)
    v = sc.vertices.reshape(-1, 3, 3)[tri]
    w = r.dirichlet([1, 1, 1], n).astype(np.float32)
    w[::10] = [1, 0, 0]                                          # through a vertex; [::11] along an edge
    w[::11, 2] = 0
    target = (v * w[:, :, None]).sum(1).astype(np.float32)
    origin = (target + r.normal(size=(n, 3)).astype(np.float32) * np.float32(0.7)).astype(np.float32)
    d = target - origin
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    items = np.concatenate([ibits(tri.astype(np.int32)).reshape(-1, 1), origin, d], 1)
    out = rm.debug_eval(FN["TRI_HIT"], items, 18)
    L = orc.lib()
    hits = 0
    fp = lambda a: np.ascontiguousarray(a, np.float32).ctypes.data_as(C.POINTER(C.c_float))
    for k in range(n):
        t = int(tri[k])
        ref = np.zeros(17, np.float32)
        ok = L.oracle_tri_hit(fp(sc.vertices.reshape(-1, 9)[t]), fp(sc.normals.reshape(-1, 9)[t]), fp(sc.tangents.reshape(-1, 9)[t]),
                              fp(sc.uvs.reshape(-1, 6)[t]), float(sc.tangent_sign[t]), fp(origin[k]), fp(d[k]), fp(ref))
        assert (out[k, 0] == 1.0) == bool(ok), k
        if ok:
            hits += 1
            assert same(out[k, 1:], ref).all(), (k, out[k, 1:], ref)
    assert hits > n // 2


def _hd(r, n):
    hd = r.random((n, 20)).astype(np.float32)
    hd[:, 5] = np.where(r.random(n) < 0.1, 1.0, hd[:, 5] * 0.5)                       # transmission (== 1 switches the BRDF off)
    t = r.normal(size=(n, 3)); b = r.normal(size=(n, 3))                                # un-normalised, non-orthogonal frame, as interpolated
    hd[:, 14:17] = t
    hd[:, 17:20] = b
    return hd


def _dirs(r, n):
    d = r.normal(size=(n, 3))
    return (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)


def test_disney_eval_pdf_sample(rig):
    sc, rm, orc = rig
    r = np.random.default_rng(3)
    n = 2000
    hd, V, N, Lv = _hd(r, n), _dirs(r, n), _dirs(r, n), _dirs(r, n)
    Lv[::7] = -N[::7]                                                                   # below the horizon: eval 0, pdf 1.0 (src/Disney.cpp:109-111)
    L = orc.lib()
    fp = lambda a: np.ascontiguousarray(a, np.float32).ctypes.data_as(C.POINTER(C.c_float))
    ev = rm.debug_eval(FN["DISNEY_EVAL"], np.concatenate([hd, V, N, Lv], 1), 3)
    pd = rm.debug_eval(FN["DISNEY_PDF"], np.concatenate([hd, V, N, Lv], 1), 1)
    rs = r.random((n, 3)).astype(np.float32)
    sm = rm.debug_eval(FN["DISNEY_SAMPLE"], np.concatenate([hd, V, N, rs], 1), 3)
    for k in range(n):
        o = np.zeros(3, np.float32)
        L.oracle_disney_eval(fp(hd[k]), fp(V[k]), fp(N[k]), fp(Lv[k]), orc.MATH_ER, fp(o))
        assert same(ev[k], o).all(), ("eval", k, ev[k], o)
        p = L.oracle_disney_pdf(fp(hd[k]), fp(V[k]), fp(N[k]), fp(Lv[k]), orc.MATH_ER)
        assert same(pd[k], np.float32(p)).all(), ("pdf", k, pd[k], p)
        L.oracle_disney_sample(fp(hd[k]), fp(V[k]), fp(N[k]), float(rs[k, 0]), float(rs[k, 1]), float(rs[k, 2]), orc.MATH_ER, fp(o))
        assert same(sm[k], o).all(), ("sample", k, sm[k], o)
    assert (pd[::7] == 1.0).all() and (ev[::7] == 0.0).all()


def test_spherical_mappings_and_textures(rig):
    sc, rm, orc = rig
    r = np.random.default_rng(5)
    L = orc.lib()
    fp = lambda a: np.ascontiguousarray(a, np.float32).ctypes.data_as(C.POINTER(C.c_float))
    p = _dirs(r, 1500)
    p[:6] = [[0, 1, 0], [0, -1, 0], [1, 0, 0], [-1, 0, 0], [0, 0, 1], [0, 0, -1]]        # poles and the seam
    uv = rm.debug_eval(FN["SPHERICAL"], p, 2)
    uvs = np.concatenate([r.random((1500, 2)).astype(np.float32), np.array([[0, 0], [1, 1], [0.5, 0], [0.5, 1]], np.float32)])
    rev = rm.debug_eval(FN["REV_SPHERICAL"], uvs, 3)
    for k in range(len(p)):
        u, v = C.c_float(), C.c_float()
        L.oracle_spherical_mapping(fp(p[k]), orc.MATH_ER, C.byref(u), C.byref(v))
        assert same(uv[k], np.array([u.value, v.value], np.float32)).all(), (k, p[k])
    for k in range(len(uvs)):
        o = np.zeros(3, np.float32)
        L.oracle_reverse_spherical_mapping(float(uvs[k, 0]), float(uvs[k, 1]), orc.MATH_ER, fp(o))
        assert same(rev[k], o).all(), k
    # textures: every scene texture (3-channel NO_FILTER, 2-channel and 1-channel BILINEAR) and the HDRI; u, v beyond [0,1] and negative
    n_tex = len(sc.textures)
    for tid in list(range(n_tex))[-3:] + [0, -1]:
        data, w, h, ch, flt = sc.hdri if tid < 0 else sc.textures[tid]
        tex = abi.ErTexture(w, h, ch, flt, abi._fptr(data))
        q = r.uniform(-2.5, 3.5, (400, 2)).astype(np.float32)
        for filtered in (0, 1):
            items = np.concatenate([np.full((400, 1), ibits(tid)), q, np.full((400, 1), np.float32(filtered))], 1)
            got = rm.debug_eval(FN["TEXTURE"], items, 3)
            for k in range(400):
                o = np.zeros(3, np.float32)
                L.oracle_texture_fetch(C.byref(tex), float(q[k, 0]), float(q[k, 1]), filtered, fp(o))
                assert same(got[k], o).all(), (tid, filtered, k, q[k], got[k], o)


def test_hdri_search_and_pdf(rig):
    sc, rm, orc = rig
    r = np.random.default_rng(6)
    L = orc.lib()
    data, w, h, ch, flt = sc.hdri
    tex = abi.ErTexture(w, h, ch, flt, abi._fptr(data))
    cdf = np.zeros(w * h + 1, np.float32)
    rsum = C.c_float()
    L.oracle_hdri_cdf(C.byref(tex), cdf.ctypes.data_as(C.POINTER(C.c_float)), C.byref(rsum))
    vals = np.concatenate([r.random(3000).astype(np.float32), cdf[r.integers(0, w * h + 1, 300)], np.array([0.0, 1.0], np.float32)])
    got = rm.debug_eval(FN["HDRI_SEARCH"], vals.reshape(-1, 1), 1).view(np.int32).reshape(-1)
    ref = np.array([L.oracle_hdri_binary_search(cdf.ctypes.data_as(C.POINTER(C.c_float)), float(v), w * h) for v in vals], np.int32)
    assert (got == ref).all()
    xy = np.stack([r.integers(0, w, 600), r.integers(0, h, 600)], 1).astype(np.int32)
    xy[:4, 1] = 0                                                # row 0: sin(theta) = 0 -> inf / nan, absorbed by the NaN gate
    got = rm.debug_eval(FN["HDRI_PDF"], ibits(xy), 1).reshape(-1)
    ref = np.array([L.oracle_hdri_pdf(C.byref(tex), rsum.value, int(x), int(y), orc.MATH_ER) for x, y in xy], np.float32)
    assert same(got, ref).all()


def test_transcendentals(rig):
    sc, rm, orc = rig
    r = np.random.default_rng(7)
    L = orc.lib()
    n = 4000
    for op, (lo, hi) in enumerate([(-20, 20), (-20, 20), (-1, 1), (1e-6, 50), (0, 4), (-3, 3)]):
        x = r.uniform(lo, hi, n).astype(np.float32)
        y = r.uniform(-3, 3, n).astype(np.float32) if op in (4, 5) else np.zeros(n, np.float32)
        x[:4] = [0.0, 1.0, -1.0 if lo < 0 else 1.0, lo if lo > 0 else 0.5]
        items = np.concatenate([np.full((n, 1), ibits(op)), x[:, None], y[:, None]], 1)
        got = rm.debug_eval(FN["MATH"], items, 1).reshape(-1)
        ref = np.zeros(n, np.float32)
        fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
        L.oracle_math(op, orc.MATH_ER, fp(x), fp(y), fp(ref), n)
        assert same(got, ref).all(), op


def test_texture_fetch_refuses_entries_the_library_keeps_compacted(oracle_mod):
    """With the compaction on (the default), a texture that materials use for scalar channels only lives on the device with its first
    channel alone, possibly already to the power 2.2: a debug fetch from that entry would not be a fetch from the scene's texture, so
    er_debug_eval(ER_FN_TEXTURE) refuses it (ER_ERR_STATE, with the way out in the message) instead of answering with other values
    (ADVICE r4); textures that are kept as they came -- an albedo texture, the HDRI -- are still served."""
    sc = scenes.torture(600, 48, 36, seed=9, n_materials=4, tex_size=16, hdri_size=(64, 32), smooth=True, n_lights=0)
    rm = render.RenderingManager(render.RenderParameters(max_bounces=8))
    rm.start_rendering(sc)
    try:
        rough = sc.materials[0].roughness_tex
        albedo = sc.materials[0].albedo_tex
        assert rough >= 0 and albedo >= 0 and rough != albedo
        def item(tid):
            row = np.array([[0.0, 0.3, 0.6, 0.0]], np.float32)
            row.view(np.int32)[0, 0] = tid          # (the id travels as float BITS)
            return row
        with pytest.raises(abi.ErError) as e:
            rm.debug_eval(FN["TEXTURE"], item(rough), 3)
        assert e.value.code == abi.ER_ERR_STATE and "ER_TEX_COMPACT=0" in str(e.value)
        assert np.isfinite(rm.debug_eval(FN["TEXTURE"], item(albedo), 3)).all()
        assert np.isfinite(rm.debug_eval(FN["TEXTURE"], item(-1), 3)).all()
    finally:
        rm.close()
