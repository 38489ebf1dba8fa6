"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on identical inputs.

Bar (DESIGN.md "Parity"): the oracle in er_math mode performs the same float operations as
the kernel, so images must be BIT-EXACT for every pixel whose paths never met two hits at
exactly equal distance (the only place where the two acceleration structures may pick a
different winner); such pixels must stay within the image tolerance of SURVEY 8c
(|d| <= 1e-3 + 1e-3|ref|) -- and must be rare.  Against the oracle in libm mode (what a CPU
build of the reference computes) the image tolerance alone applies.
"""
import os

import numpy as np
import pytest

from elevenrender_amd import abi, render, scenes

pytestmark = pytest.mark.gpu

PLANES = [abi.PASS_BEAUTY, abi.PASS_DENOISE, abi.PASS_NORMAL, abi.PASS_TANGENT, abi.PASS_BITANGENT]


def gpu_render(scene, spp, max_bounces=5, chunks=None, **kw):
    rm = render.RenderingManager(render.RenderParameters(max_bounces=max_bounces, **kw))
    rm.start_rendering(scene)
    for n in (chunks or [spp]):
        rm.render(n)
    out = {p: rm.get_pass(p) for p in ("beauty", "denoise", "normal", "tangent", "bitangent")}
    out["samples"] = rm.read_samples()
    out["rng"] = rm.read_rng()
    out["counters"] = rm.counters()
    out["info"] = rm.get_render_info().samples
    rm.close()
    return out


def oracle_render(oracle_mod, scene, spp, max_bounces=5, math_mode=1, traversal=0, threads=8):
    o = oracle_mod.Oracle(scene, math_mode=math_mode, max_bounces=max_bounces, traversal=traversal, threads=threads)
    o.render(spp)
    out = {name: o.read_pass(p) for name, p in abi.PASS_NAMES.items()}
    out["samples"] = o.read_samples()
    out["rng"] = o.read_rng()
    out["counters"] = o.counters()
    o.close()
    return out


def compare(g, o, min_exact=0.999, what=""):
    exact = np.ones(g["beauty"].shape[:2], bool)
    for p in ("beauty", "denoise", "normal", "tangent", "bitangent"):
        exact &= (g[p].view(np.uint32) == o[p].view(np.uint32)).all(-1)
    exact &= (g["rng"] == o["rng"]).reshape(exact.shape)
    exact &= (g["samples"] == o["samples"]).reshape(exact.shape)
    frac = exact.mean()
    d = np.abs(g["beauty"] - o["beauty"])
    tol = 1e-3 + 1e-3 * np.abs(o["beauty"])
    within = (d <= tol).all(-1).mean()
    print(f"{what}: bit-exact pixels {frac:.6f}, within tolerance {within:.6f}, max|d| {d.max():.3e}")
    assert frac >= min_exact, f"{what}: only {frac:.6f} of pixels bit-exact"
    assert within >= 0.995, f"{what}: only {within:.6f} of pixels within tolerance"
    gm, om = g["beauty"][..., :3].mean((0, 1)), o["beauty"][..., :3].mean((0, 1))
    assert np.allclose(gm, om, rtol=1e-4, atol=1e-6), f"{what}: image mean {gm} vs {om}"
    return frac


def test_c1_cornell_bit_exact(oracle_mod):
    """BASELINE config 1: Cornell 12 tris, 256x256, 16 spp (4 bounces in the config; 5 = the reference literal too)."""
    sc = scenes.cornell(256, 256)
    for mb in (5, 4):
        g = gpu_render(sc, 16, max_bounces=mb)
        o = oracle_render(oracle_mod, sc, 16, max_bounces=mb)
        compare(g, o, what=f"C1 max_bounces={mb}")
        assert g["counters"]["bounce_samples"] == o["counters"]["bounce_samples"]
        assert g["counters"]["paths"] == 256 * 256 * 16
        assert g["info"] == 17   # reference semantic: samples done + 1
        # DENOISE plane is never written: stays (0,0,0,1)
        assert (g["denoise"][..., :3] == 0).all() and (g["denoise"][..., 3] == 1).all()


def test_soup_small_bit_exact(oracle_mod):
    """2k-triangle soup + sky HDRI, 64x64, 8 spp, 8 bounces, reference-BVH oracle."""
    sc = scenes.soup(2000, 64, 64, seed=7, hdri_size=(64, 32))
    g = gpu_render(sc, 8, max_bounces=8)
    o = oracle_render(oracle_mod, sc, 8, max_bounces=8)
    compare(g, o, what="soup2k")
    assert g["counters"]["bounce_samples"] == o["counters"]["bounce_samples"]


def test_chunked_equals_single_launch(oracle_mod):
    """n samples in one launch == n launches of one sample (per-pixel RNG streams are sequential)."""
    sc = scenes.soup(500, 48, 40, seed=3, hdri_size=(32, 16))
    a = gpu_render(sc, 6, max_bounces=5)
    b = gpu_render(sc, 6, max_bounces=5, chunks=[1, 1, 1, 1, 1, 1])
    c = gpu_render(sc, 6, max_bounces=5, chunks=[2, 4])
    for p in ("beauty", "normal", "tangent", "bitangent"):
        assert (a[p].view(np.uint32) == b[p].view(np.uint32)).all()
        assert (a[p].view(np.uint32) == c[p].view(np.uint32)).all()
    assert (a["rng"] == b["rng"]).all() and (a["rng"] == c["rng"]).all()


def test_libm_oracle_within_tolerance(oracle_mod):
    """Against the libm-mode oracle (what the reference computes on a CPU) only the image tolerance holds."""
    sc = scenes.cornell(128, 128)
    g = gpu_render(sc, 16)
    o = oracle_render(oracle_mod, sc, 16, math_mode=0)
    compare(g, o, min_exact=0.90, what="C1 vs libm oracle")


@pytest.mark.parametrize("name", ["cornell_32x32_4spp", "torture_300tri_32x24_4spp"])
def test_gpu_reproduces_golden_fixtures(name):
    """Committed fixtures (tests/golden): inputs + the oracle's outputs.  Covers textures, normal map,
    opacity < 1, asl_shade placeholder, bokeh camera, rotated camera, smooth normals (lifted positions)."""
    from golden_util import load
    sc, spp, mb, z = load(name)
    g = gpu_render(sc, spp, max_bounces=mb)
    for p in ("beauty", "denoise", "normal", "tangent", "bitangent"):
        same = (g[p].view(np.uint32) == z[f"pass_{p}"].view(np.uint32)).all(-1)
        assert same.mean() >= 0.999, (p, same.mean())
    assert (g["rng"] == z["rng"]).mean() >= 0.999 and (g["samples"] == z["samples"]).mean() >= 0.999
    assert g["counters"]["bounce_samples"] == int(z["counters"][1])


def test_megakernel_schedule_equals_wavefront_schedule():
    """ER_FLAG_MEGAKERNEL (one fused kernel) and the default wavefront schedule are the same arithmetic."""
    sc = scenes.soup(3000, 96, 64, seed=11, hdri_size=(64, 32))
    a = gpu_render(sc, 5, max_bounces=8, flags=abi.FLAG_WAVEFRONT)
    b = gpu_render(sc, 5, max_bounces=8, flags=abi.FLAG_MEGAKERNEL)
    for p in ("beauty", "normal", "tangent", "bitangent"):
        assert (a[p].view(np.uint32) == b[p].view(np.uint32)).all()
    assert (a["rng"] == b["rng"]).all() and (a["samples"] == b["samples"]).all()
    assert a["counters"]["bounce_samples"] == b["counters"]["bounce_samples"]
    assert a["counters"]["paths"] == b["counters"]["paths"] == 96 * 64 * 5


def test_smooth_blobs_bit_exact(oracle_mod):
    """C4-style geometry (smooth vertex normals -> lifted hit positions, non-zero lift bounds): exercises the
    t-interval bookkeeping of er_wf_trace, its two-candidate exact resolve and the exact re-trace fallback."""
    sc = scenes.blob_instances(n_instances=60, tris_per_blob=300, x_res=96, y_res=64, grid=(5, 4, 3), spacing=0.45)
    g = gpu_render(sc, 6, max_bounces=8)
    o = oracle_render(oracle_mod, sc, 6, max_bounces=8)
    compare(g, o, what="smooth blobs")
    assert g["counters"]["bounce_samples"] == o["counters"]["bounce_samples"]
    m = gpu_render(sc, 6, max_bounces=8, flags=abi.FLAG_MEGAKERNEL)
    assert (m["beauty"].view(np.uint32) == g["beauty"].view(np.uint32)).all()


def test_c5_textured_materials_bit_exact(oracle_mod):
    """C5-style: textured materials (albedo/roughness/metallic maps), clearcoat/anisotropic/sheen, 16 bounces."""
    sc = scenes.torture(4000, 80, 60, seed=5, n_materials=16, tex_size=32, hdri_size=(128, 64))
    g = gpu_render(sc, 4, max_bounces=16)
    o = oracle_render(oracle_mod, sc, 4, max_bounces=16)
    compare(g, o, what="C5 small")
    assert g["counters"]["bounce_samples"] == o["counters"]["bounce_samples"]


def test_mid_size_soup_bit_exact(oracle_mod):
    """100k-triangle soup, 160x120, 3 spp, 8 bounces against the reference-BVH oracle (multi-threaded)."""
    sc = scenes.soup(100_000, 160, 120, seed=12345, hdri_size=(512, 256))
    g = gpu_render(sc, 3, max_bounces=8)
    o = oracle_render(oracle_mod, sc, 3, max_bounces=8, threads=16)
    compare(g, o, what="soup100k")
    assert g["counters"]["bounce_samples"] == o["counters"]["bounce_samples"]


def test_tile_sharding_invariance_on_one_gpu():
    """rank r of `world` renders only its tiles; stitched together (er_pack_owned/er_unpack_owned, the RCCL
    combine minus the wire) the image equals the single-GPU image bit for bit."""
    import ctypes as C
    lib = abi.load()
    hip = C.CDLL("libamdhip64.so")
    sc = scenes.soup(5000, 100, 76, seed=9, hdri_size=(64, 32))     # 100x76: partial tiles on both edges
    full = gpu_render(sc, 5, max_bounces=8)
    world = 3
    rms = []
    for r in range(world):
        rm = render.RenderingManager(render.RenderParameters(max_bounces=8, rank=r, world=world))
        rm.start_rendering(sc)
        rm.render(5)
        rms.append(rm)
    total_paths = sum(rm.counters()["paths"] for rm in rms)
    assert total_paths == 100 * 76 * 5
    for p in (abi.PASS_BEAUTY, abi.PASS_NORMAL):
        for r in range(1, world):
            n = rms[r].owned_count(r)
            buf = C.c_void_p()
            assert hip.hipMalloc(C.byref(buf), C.c_size_t(n * 16)) == 0
            rms[r].pack_owned(p, buf.value)
            rms[0].unpack_owned(p, r, buf.value)
            hip.hipFree(buf)
    stitched = rms[0].get_pass("beauty")
    assert (stitched.view(np.uint32) == full["beauty"].view(np.uint32)).all()
    assert (rms[0].get_pass("normal").view(np.uint32) == full["normal"].view(np.uint32)).all()
    for rm in rms:
        rm.close()


def test_full_size_properties():
    """BASELINE config 2 at full size (1M triangles, 1920x1080, 8 bounces): size-independent properties.
    (a) chunked == single call; (b) megakernel schedule == wavefront schedule on a window of tiles (rank 7 of 64);
    (c) counters: every path counted once, samples plane = calls + 1, image finite and inside the clamp."""
    sc = scenes.soup(1_000_000, 1920, 1080, seed=12345)
    a = gpu_render(sc, 3, max_bounces=8)
    b = gpu_render(sc, 3, max_bounces=8, chunks=[1, 2])
    assert (a["beauty"].view(np.uint32) == b["beauty"].view(np.uint32)).all()
    assert (a["rng"] == b["rng"]).all()
    assert a["counters"]["paths"] == 1920 * 1080 * 3
    assert (a["samples"] == 4).mean() > 0.999          # NaN-gated samples are the only exceptions
    assert np.isfinite(a["beauty"]).all() and a["beauty"][..., :3].min() >= 0 and a["beauty"][..., :3].max() <= 10
    assert a["info"] == 4
    # the whole frame, bit for bit, in the two schedules that matter at this size (the default here is the streaming schedule)
    full_w = gpu_render(sc, 3, max_bounces=8, flags=abi.FLAG_WAVEFRONT)
    full_s = gpu_render(sc, 3, max_bounces=8, flags=abi.FLAG_STREAM)
    for p in ("beauty", "normal", "tangent", "bitangent"):
        assert (a[p].view(np.uint32) == full_w[p].view(np.uint32)).all(), p
        assert (a[p].view(np.uint32) == full_s[p].view(np.uint32)).all(), p
    assert (a["rng"] == full_w["rng"]).all() and (a["rng"] == full_s["rng"]).all()
    for k in ("paths", "bounce_samples", "rays", "shaded_hits", "hdri_samples"):
        assert a["counters"][k] == full_w["counters"][k] == full_s["counters"][k], k
    w = gpu_render(sc, 3, max_bounces=8, rank=7, world=64, flags=abi.FLAG_WAVEFRONT)
    m = gpu_render(sc, 3, max_bounces=8, rank=7, world=64, flags=abi.FLAG_MEGAKERNEL)
    f = gpu_render(sc, 3, max_bounces=8, rank=7, world=64, flags=abi.FLAG_STREAM)
    assert (w["beauty"].view(np.uint32) == f["beauty"].view(np.uint32)).all()
    assert (w["beauty"].view(np.uint32) == m["beauty"].view(np.uint32)).all()
    assert w["counters"]["bounce_samples"] == m["counters"]["bounce_samples"]
    # the sharded window agrees with the full render on the pixels it owns
    from elevenrender_amd import dist as erdist
    idx = erdist.tile_pixel_index(erdist.owned_tiles(7, 64, 1920, 1080), 1920, 1080)
    idx = idx[idx >= 0]
    assert (w["beauty"].reshape(-1, 4)[idx].view(np.uint32) == a["beauty"].reshape(-1, 4)[idx].view(np.uint32)).all()


@pytest.mark.parametrize("scene_kind", ["soup", "blobs", "cornell"])
def test_streaming_schedule_equals_wavefront_schedule(scene_kind):
    """The CU-resident streaming schedule (the default at every frame size) computes the same arithmetic as the launch-per-bounce
    wavefront schedule, in one call or several; ER_FLAG_FUSED -- round 1's single kernel, removed in round 5 -- is still accepted and
    means the streaming schedule."""
    if scene_kind == "soup":
        sc = scenes.soup(20000, 136, 100, seed=4, hdri_size=(128, 64))
    elif scene_kind == "blobs":
        sc = scenes.blob_instances(n_instances=40, tris_per_blob=300, x_res=96, y_res=72, grid=(5, 4, 2), spacing=0.45)
    else:
        sc = scenes.cornell(100, 60)
    a = gpu_render(sc, 7, max_bounces=8, flags=abi.FLAG_WAVEFRONT)
    b = gpu_render(sc, 7, max_bounces=8)                                      # the automatic choice: the streaming schedule
    c = gpu_render(sc, 7, max_bounces=8, flags=abi.FLAG_FUSED, chunks=[3, 4])  # (the retired flag)
    d = gpu_render(sc, 7, max_bounces=8, flags=abi.FLAG_STREAM)
    e = gpu_render(sc, 7, max_bounces=8, flags=abi.FLAG_STREAM, chunks=[2, 5])
    for p in ("beauty", "normal", "tangent", "bitangent"):
        for other in (b, c, d, e):
            assert (a[p].view(np.uint32) == other[p].view(np.uint32)).all(), p
    for other in (b, d):
        assert (a["rng"] == other["rng"]).all() and (a["samples"] == other["samples"]).all()
        assert a["counters"]["bounce_samples"] == other["counters"]["bounce_samples"]
        assert a["counters"]["paths"] == other["counters"]["paths"]
        assert a["counters"]["rays"] == other["counters"]["rays"]


def test_streaming_schedule_long_calls_complete():
    """Regression: with ring positions reserved AHEAD of their producers a cell could be overwritten after the ring had wrapped
    (a lane whose wave did not poll for a few hundred microseconds), the ray was lost, its workgroup never finished and the
    watchdog ended the call -- first seen as `er_wait: ... watchdog status 1` on the C2 frame with a 4-sample call followed by
    a 64-sample one.  The same calls must complete, and count every path."""
    sc = scenes.soup(1_000_000, 1920, 1080, seed=12345)
    rm = render.RenderingManager(render.RenderParameters(max_bounces=8, flags=abi.FLAG_STREAM))
    rm.start_rendering(sc)
    rm.render(4)
    rm.render(64)          # raises if the library reports the watchdog
    c = rm.counters()
    samples = rm.read_samples()
    rm.close()
    assert c["paths"] == 1920 * 1080 * 68
    assert (samples == 69).mean() > 0.999


def test_streaming_schedule_small_frame_call_patterns():
    """Regression run for the two call patterns in which round 2's ring protocol failed on small frames (64 pixels per workgroup:
    the pixel ring turns over in microseconds): one call, and a short call followed by a longer one.  ONE render each -- the
    argument that the protocol is exact is not this test but the host-thread model of the same ring functions under
    ThreadSanitizer (tests/test_stream_protocol_cpu.py, csrc/er_ring.h); a GPU loop would only show absence of evidence."""
    sc = scenes.blob_instances(n_instances=40, tris_per_blob=300, x_res=96, y_res=72, grid=(5, 4, 2), spacing=0.45)
    ref = gpu_render(sc, 7, max_bounces=8, flags=abi.FLAG_WAVEFRONT)
    for chunks in (None, [2, 5]):
        g = gpu_render(sc, 7, max_bounces=8, flags=abi.FLAG_STREAM, chunks=chunks)
        assert (g["beauty"].view(np.uint32) == ref["beauty"].view(np.uint32)).all(), chunks
        assert (g["rng"] == ref["rng"]).all() and (g["samples"] == ref["samples"]).all(), chunks
        assert g["counters"]["paths"] == ref["counters"]["paths"], chunks


def test_streaming_schedule_many_turns_of_the_pixel_ring():
    """er_stream.hip: a workgroup holds 1024 of its pixels in slots and the others in a ring that every finished sample goes
    through.  1024x768 = 3072 pixels per workgroup: the ring turns over once per sample, 24 times here (its positions wrap
    its capacity several times), in one call and in uneven chunks -- bit for bit the wavefront schedule's frame."""
    sc = scenes.soup(50_000, 1024, 768, seed=21, hdri_size=(256, 128))
    w = gpu_render(sc, 24, max_bounces=6, flags=abi.FLAG_WAVEFRONT)
    s1 = gpu_render(sc, 24, max_bounces=6, flags=abi.FLAG_STREAM)
    s2 = gpu_render(sc, 24, max_bounces=6, flags=abi.FLAG_STREAM, chunks=[1, 7, 16])
    for other in (s1, s2):
        for p in ("beauty", "normal", "tangent", "bitangent"):
            assert (w[p].view(np.uint32) == other[p].view(np.uint32)).all(), p
        assert (w["rng"] == other["rng"]).all() and (w["samples"] == other["samples"]).all()
    for k in ("paths", "bounce_samples", "rays", "shaded_hits", "hdri_samples"):
        assert w["counters"][k] == s1["counters"][k] == s2["counters"][k], k


def _empty_scene(x_res, y_res):
    sc = scenes.soup(1, x_res, y_res, seed=3, hdri_size=(64, 32))
    keep = np.zeros(0, np.int64)
    for name in ("vertices", "normals", "tangents", "uvs", "tangent_sign", "material_id"):
        setattr(sc, name, np.ascontiguousarray(getattr(sc, name)[keep]))
    sc.tri_count = 0
    sc._desc = None
    return sc


@pytest.mark.parametrize("flags", [abi.FLAG_WAVEFRONT, abi.FLAG_MEGAKERNEL, abi.FLAG_STREAM])
def test_edge_cases_bit_exact(oracle_mod, flags):
    """Inputs at the edges of the domain, in every schedule: no triangles at all (every ray sees the HDRI), a frame
    smaller than one tile and frames with partial tiles on both edges, zero-area and duplicated triangles (exactly
    equal hit distances are the one documented source of non-bit-exact pixels, so duplicates are offset by 1e-3)."""
    sc = _empty_scene(17, 9)
    compare(gpu_render(sc, 3, max_bounces=8, flags=flags), oracle_render(oracle_mod, sc, 3, max_bounces=8), what="empty scene")
    sc = scenes.soup(1, 5, 3, seed=11, hdri_size=(64, 32))
    compare(gpu_render(sc, 4, max_bounces=8, flags=flags), oracle_render(oracle_mod, sc, 4, max_bounces=8), what="one triangle, 5x3")
    sc = scenes.soup(600, 29, 23, seed=21, hdri_size=(64, 32))
    v = sc.vertices.reshape(-1, 3, 3)
    v[::7, 2] = v[::7, 1]                   # zero-area: two equal corners
    v[1::11] = v[0::11][: len(v[1::11])] + np.float32(1e-3)   # near-duplicates of other triangles
    sc.vertices = np.ascontiguousarray(v.reshape(sc.vertices.shape))
    sc._desc = None
    compare(gpu_render(sc, 3, max_bounces=8, flags=flags), oracle_render(oracle_mod, sc, 3, max_bounces=8), what="degenerate soup")


def _window_schedules_agree(sc, spp, max_bounces, rank, world, mega_ties=0.0, extra_flags=0):
    """wavefront == streaming bit for bit (same traversal order); the megakernel walks the binary BVH, so on closed
    meshes it may resolve an exact distance tie on a shared edge the other way (`mega_ties` = tolerated pixel fraction;
    the two pixels found on C4 were checked against the oracle in both of its traversal orders: wavefront's answer)."""
    w = gpu_render(sc, spp, max_bounces=max_bounces, rank=rank, world=world, flags=abi.FLAG_WAVEFRONT | extra_flags)
    f = gpu_render(sc, spp, max_bounces=max_bounces, rank=rank, world=world, flags=extra_flags)      # the automatic choice
    m = gpu_render(sc, spp, max_bounces=max_bounces, rank=rank, world=world, flags=abi.FLAG_MEGAKERNEL | extra_flags)
    st = gpu_render(sc, spp, max_bounces=max_bounces, rank=rank, world=world, flags=abi.FLAG_STREAM | extra_flags)
    owned = w["samples"].reshape(w["beauty"].shape[:2]) > 1
    for p in ("beauty", "normal", "tangent", "bitangent"):
        assert (w[p].view(np.uint32) == f[p].view(np.uint32)).all(), p
        assert (w[p].view(np.uint32) == st[p].view(np.uint32)).all(), p
        differ = (w[p].view(np.uint32) != m[p].view(np.uint32)).any(-1)
        assert differ.sum() <= mega_ties * owned.sum(), (p, int(differ.sum()))
    assert (w["rng"] == f["rng"]).all() and (w["rng"] != m["rng"]).sum() <= mega_ties * owned.sum()
    assert w["counters"]["bounce_samples"] == f["counters"]["bounce_samples"]
    if mega_ties == 0.0:
        assert w["counters"]["bounce_samples"] == m["counters"]["bounce_samples"]
    return w


def test_c5_full_size_properties():
    """BASELINE config 5 at full size: 1M triangles, 64 materials with 192 value-noise textures of 256x256, clearcoat /
    anisotropic / sheen variants, 16 bounces, 1920x1080, reference behaviour (the descriptor's 256 point lights are ignored
    without ER_FLAG_POINT_LIGHTS; the lit configuration runs in tests/test_gpu_lights.py).
    The three schedules agree bit for bit on a window of tiles; chunked calls equal one call on that window."""
    sc = scenes.torture(1_000_000, 1920, 1080, seed=12345)
    w = _window_schedules_agree(sc, 2, 16, rank=5, world=48)
    c = gpu_render(sc, 2, max_bounces=16, rank=5, world=48, chunks=[1, 1])
    assert (w["beauty"].view(np.uint32) == c["beauty"].view(np.uint32)).all()
    assert np.isfinite(w["beauty"]).all() and w["beauty"][..., :3].max() <= 10
    assert w["counters"]["texel_fetches"] == 0 and w["counters"]["shaded_hits"] > 0      # texel counter needs ER_FLAG_COUNTERS


def test_c4_full_size_properties():
    """BASELINE config 4 at full size: 10M triangles (10 000 instances of a 1 000-triangle smooth-normal blob), 3840x2160.
    Whole frame once (finite, clamped, every path counted, sample plane = calls + 1), then the three schedules bit for
    bit on a window of tiles, which must also equal the whole-frame render on the pixels it owns."""
    sc = scenes.blob_instances()
    assert sc.tri_count == 10_000_000 and (sc.x_res, sc.y_res) == (3840, 2160)
    a = gpu_render(sc, 2, max_bounces=8)
    assert a["counters"]["paths"] == 3840 * 2160 * 2
    assert (a["samples"] == 3).mean() > 0.999
    assert np.isfinite(a["beauty"]).all() and a["beauty"][..., :3].min() >= 0 and a["beauty"][..., :3].max() <= 10
    w = _window_schedules_agree(sc, 2, 8, rank=11, world=96, mega_ties=1e-4)
    from elevenrender_amd import dist as erdist
    idx = erdist.tile_pixel_index(erdist.owned_tiles(11, 96, 3840, 2160), 3840, 2160)
    idx = idx[idx >= 0]
    assert (w["beauty"].reshape(-1, 4)[idx].view(np.uint32) == a["beauty"].reshape(-1, 4)[idx].view(np.uint32)).all()
    # the same window through a tree built on the device (er_gpu_build.hip): only exact-tie pixels may differ
    g = gpu_render(sc, 2, max_bounces=8, rank=11, world=96, flags=abi.FLAG_GPU_BUILD)
    assert (g["beauty"].view(np.uint32) != w["beauty"].view(np.uint32)).any(-1).sum() <= 1e-4 * len(idx)


def test_closest_hit_function_level(oracle_mod):
    """Function-level parity of a4/a5: the library's closest hit (include/eleven_hip_debug.h) against the
    oracle's throwRay on the same rays -- camera-like rays, rays leaving surface points, axis-parallel rays (a zero
    direction component) -- on a soup and on the Cornell box, whose coplanar wall triangles share edges.  Hit.position
    must agree bit for bit; on the box a ray through a seam hits two triangles at exactly the same distance, the one
    case where the winner depends on the traversal order, so a few rays in 10 000 may differ there."""
    rng = np.random.default_rng(7)
    for sc, tol in ((scenes.soup(6000, 64, 48, seed=13, hdri_size=(64, 32)), 0.0), (scenes.cornell(64, 48), 1e-3)):
        n = 20000
        o = np.tile(np.array([[0.01, 0.02, -0.5]], np.float32), (n, 1))
        d = rng.normal(size=(n, 3)).astype(np.float32)
        d[:, 2] = np.abs(d[:, 2]) + 0.5
        d[::50, 0] = 0.0                                   # axis-parallel components
        d[25::50, 1] = 0.0
        d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
        orc = oracle_mod.Oracle(sc, math_mode=oracle_mod.MATH_ER, max_bounces=8, threads=1)
        rm = render.RenderingManager(render.RenderParameters(max_bounces=8))
        rm.start_rendering(sc)
        for generation in range(3):
            tri, pos, dist = rm.debug_closest_hit(o, d)
            otri, opos = orc.closest_hit(o, d)
            same = (pos.view(np.uint32) == opos.view(np.uint32)).all(-1) & ((tri < 0) == (otri < 0))
            print(f"{sc.tri_count} triangles, generation {generation}: {int((tri >= 0).sum())} hits, {int((~same).sum())} rays differ")
            assert (~same).mean() <= tol, int((~same).sum())
            hit = (tri >= 0) & same
            # next generation: rays leaving the hit points in random directions (reference: position + dir * 0.001)
            nd = rng.normal(size=(n, 3)).astype(np.float32)
            nd /= np.linalg.norm(nd, axis=1, keepdims=True).astype(np.float32)
            o = np.where(hit[:, None], (pos + nd * np.float32(0.001)).astype(np.float32), o)
            d = np.where(hit[:, None], nd, d)
        rm.close()
        orc.close()


@pytest.mark.parametrize("flags", [abi.FLAG_WAVEFRONT, abi.FLAG_STREAM])
def test_read_back_during_asynchronous_rendering_is_a_sample_boundary_snapshot(flags):
    """The reference reads passes on a second queue while the render thread enqueues samples, unsynchronised
    (src/Managers.cpp:287-302: torn reads).  Here er_read_pass is ordered after everything enqueued so far -- including
    the wavefront schedule's slot pools on their own streams -- so a read issued right after an asynchronous
    er_render_samples returns exactly the state after those samples."""
    sc = scenes.soup(30000, 200, 120, seed=19, hdri_size=(128, 64))
    want3 = gpu_render(sc, 3, max_bounces=8, flags=flags)
    want8 = gpu_render(sc, 8, max_bounces=8, flags=flags)
    rm = render.RenderingManager(render.RenderParameters(max_bounces=8, flags=flags))
    rm.start_rendering(sc)
    rm.render(3, blocking=False)
    got3 = rm.get_pass("beauty")                  # no er_wait in between
    assert rm.get_render_info().samples == 4
    rm.render(5, blocking=False)
    got8 = rm.get_pass("beauty")
    rm.wait()
    rm.close()
    assert (got3.view(np.uint32) == want3["beauty"].view(np.uint32)).all()
    assert (got8.view(np.uint32) == want8["beauty"].view(np.uint32)).all()


def test_two_threads_reading_two_passes_of_one_scene_each_get_their_own():
    """er_read_pass gathers the requested plane into the scene's ONE staging buffer and then copies it out; with
    the lock released in between, two threads could interleave as gather A, gather B, copy, copy and the first silently
    received the other pass.  The lock now spans gather + copy: 40 concurrent pairs of reads, every one its own plane."""
    import threading
    sc = scenes.soup(20000, 320, 200, seed=23, hdri_size=(128, 64))
    rm = render.RenderingManager(render.RenderParameters(max_bounces=8))
    rm.start_rendering(sc)
    rm.render(3)
    want = {p: rm.get_pass(p) for p in ("beauty", "normal")}
    assert (want["beauty"].view(np.uint32) != want["normal"].view(np.uint32)).any()
    wrong = []

    def reader(name):
        for _ in range(40):
            got = rm.get_pass(name)            # ctypes releases the GIL for the call
            if not (got.view(np.uint32) == want[name].view(np.uint32)).all():
                wrong.append(name)
    threads = [threading.Thread(target=reader, args=(n,)) for n in ("beauty", "normal")]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    rm.close()
    assert not wrong, wrong


def test_adaptive_tracer_shader_split_does_not_change_the_image(monkeypatch, capfd):
    """After a completed call the library may turn one tracer wave of the streaming kernel into a shader wave (er_stream_adapt in
    csrc/er_api.cpp: two consecutive calls whose tracer lanes were under 0.85 full).  Whatever it decides, the planes are those of a
    fixed split: a frame of 1.06 M pixels in three calls with the adaptation on == the same calls at a fixed 12 + 4 and at a fixed
    10 + 6.  The lane occupancy itself is a measurement that depends on clocks, so nothing is asserted on it: the mechanism is DRIVEN
    with a forced reading (ER_STREAM_FORCE_BUSY=0.70) and the sequence of splits it must produce is exact -- no change after the first
    low reading, one wave moved after the second, none after the third; the natural readings are printed."""
    sc = scenes.soup(60_000, 1280, 832, seed=31, hdri_size=(256, 128))
    monkeypatch.setenv("ER_STREAM_TRACERS", "12")
    fixed12 = gpu_render(sc, 6, max_bounces=8, flags=abi.FLAG_STREAM, chunks=[2, 2, 2])
    monkeypatch.setenv("ER_STREAM_TRACERS", "10")
    fixed10 = gpu_render(sc, 6, max_bounces=8, flags=abi.FLAG_STREAM, chunks=[2, 2, 2])
    monkeypatch.delenv("ER_STREAM_TRACERS")
    monkeypatch.setenv("ER_STREAM_ADAPT", "1")
    monkeypatch.setenv("ER_STREAM_VERBOSE", "1")
    monkeypatch.setenv("ER_STREAM_FORCE_BUSY", "0.70")
    capfd.readouterr()
    adaptive = gpu_render(sc, 6, max_bounces=8, flags=abi.FLAG_STREAM, chunks=[2, 2, 2])
    err = capfd.readouterr().err
    for other in (fixed10, adaptive):
        for p in ("beauty", "normal", "tangent", "bitangent"):
            assert (fixed12[p].view(np.uint32) == other[p].view(np.uint32)).all(), p
        assert (fixed12["rng"] == other["rng"]).all() and (fixed12["samples"] == other["samples"]).all()
    lines = [l for l in err.splitlines() if l.startswith("[er_stream] tracer lanes") and "->" in l]
    assert len(lines) == 3 and all("forced reading" in l for l in lines), err
    start = 9 if os.environ.get("ER_STREAM_WAVES") == "12" else 13      # (tools/knob_corners.sh runs this file at 12 waves per CU too)
    assert [int(l.split("->")[1].split("+")[0]) for l in lines] == [start, start - 1, start - 1], lines
    # a forced HIGH reading after that gives the one step back that a render is allowed, and only one
    monkeypatch.setenv("ER_STREAM_FORCE_BUSY", "0.70,0.70,0.95,0.95,0.70,0.70,0.95")
    capfd.readouterr()
    again = gpu_render(sc, 7, max_bounces=8, flags=abi.FLAG_STREAM, chunks=[1, 1, 1, 1, 1, 1, 1])
    err = capfd.readouterr().err
    lines = [l for l in err.splitlines() if l.startswith("[er_stream] tracer lanes") and "->" in l]
    assert [int(l.split("->")[1].split("+")[0]) for l in lines] == [start, start - 1, start, start, start, start - 1, start - 1], lines
    ref7 = gpu_render(sc, 7, max_bounces=8, flags=abi.FLAG_STREAM)
    assert (again["beauty"].view(np.uint32) == ref7["beauty"].view(np.uint32)).all() and (again["rng"] == ref7["rng"]).all()
    monkeypatch.delenv("ER_STREAM_FORCE_BUSY")
    capfd.readouterr()
    gpu_render(sc, 6, max_bounces=8, flags=abi.FLAG_STREAM, chunks=[2, 2, 2])
    print("natural readings:", [l for l in capfd.readouterr().err.splitlines() if l.startswith("[er_stream] tracer lanes")])


def test_host_precomputed_constants_equal_the_device_evaluation(monkeypatch):
    """Round 4 moved three per-frame / per-material evaluations from every sample to the host (csrc/er_api.cpp), computed with the
    same er_math.h functions the device calls: the camera's six rotation sines / cosines (src/kernel.cpp:371-473), pow(roughness, 2.2)
    and pow(metallic, 2.2) of untextured channels (src/kernel.cpp:152-153) and the logarithm GTR1 takes (src/Disney.cpp:40-46); and
    the texture wrap |x % w| became a mask when every texture has power-of-two sides.  Each has a knob that makes the device do the
    work itself: the images must not differ in a single bit (rotated thin-lens camera, textured + constant materials, clearcoat)."""
    sc = scenes.torture(3000, 96, 64, seed=5, n_materials=6, tex_size=16, hdri_size=(64, 32), n_lights=0)
    sc.materials[1].roughness_tex = -1
    sc.materials[1].metallic_tex = -1          # constant channels beside textured ones
    sc.materials[1].roughness, sc.materials[1].metallic = 0.37, 0.62
    sc.materials[2].clearcoat = 1.0
    sc.materials[2].clearcoat_gloss = 0.3
    sc.camera.bokeh = 1
    sc.camera.focus_distance = 3.0
    sc.camera.rotation = abi.ErVec3(3.0, -5.0, 2.0)
    sc._desc = None
    ref = gpu_render(sc, 5, max_bounces=8, flags=abi.FLAG_STREAM)
    for knob in ("ER_CAM_TRIG_ON_DEVICE", "ER_MAT_PRE_ON_DEVICE", "ER_TEX_POW2"):
        monkeypatch.setenv(knob, "0" if knob == "ER_TEX_POW2" else "1")
        for flags in (abi.FLAG_STREAM, abi.FLAG_WAVEFRONT):
            other = gpu_render(sc, 5, max_bounces=8, flags=flags)
            for p in ("beauty", "normal", "tangent", "bitangent"):
                assert (ref[p].view(np.uint32) == other[p].view(np.uint32)).all(), (knob, flags, p)
            assert (ref["rng"] == other["rng"]).all(), (knob, flags)
        monkeypatch.delenv(knob)


def test_textures_and_hdri_without_power_of_two_sides_bit_exact(oracle_mod):
    """The general texture wrap (a signed `%` by the run-time side, src/Texture.cpp:176-180) is what runs when ANY texture or the
    HDRI has a side that is not a power of two (the mask form is only taken when all are): 12x10 textures, a 24x12 HDRI."""
    sc = scenes.torture(2500, 80, 60, seed=9, n_materials=5, tex_size=16, hdri_size=(64, 32), n_lights=0)
    r = scenes.Rand(77, 3)
    sc.textures = [(abi._f32(r.u01(10, 12, 3)), 12, 10, 3, i % 2) for i in range(len(sc.textures))]      # (data [h, w, c], w, h, channels, filter)
    sc.hdri = (abi._f32(0.2 + 2.0 * r.u01(12, 24, 3)), 24, 12, 3, 0)
    sc._desc = None
    g = gpu_render(sc, 4, max_bounces=8)
    o = oracle_render(oracle_mod, sc, 4, max_bounces=8)
    compare(g, o, what="non-power-of-two textures")
    assert g["counters"]["bounce_samples"] == o["counters"]["bounce_samples"]


def test_frame_wider_than_the_streaming_schedules_packed_pixel(oracle_mod):
    """The streaming schedule carries a pixel as x | y << 16 (round 4), so it takes frames up to 65 535 x 65 535.  A 65 600 x 3 frame
    (196 800 pixels: a size at which flags = 0 would otherwise choose it) must be rendered by the automatic choice with another
    schedule, bit-exact against the oracle like every other frame, and forcing ER_FLAG_STREAM on it must be refused with
    ER_ERR_INVALID_ARG rather than wrap a coordinate."""
    sc = scenes.soup(300, 65600, 3, seed=31, hdri_size=(64, 32))
    g = gpu_render(sc, 2, max_bounces=4)
    o = oracle_render(oracle_mod, sc, 2, max_bounces=4)
    compare(g, o, what="65600 x 3 frame, automatic schedule")
    rm = render.RenderingManager(render.RenderParameters(max_bounces=4, flags=abi.FLAG_STREAM))
    with pytest.raises(abi.ErError) as e:
        rm.start_rendering(sc)
    assert e.value.code == abi.ER_ERR_INVALID_ARG and "65535" in str(e.value)
    rm.close()


def test_deal_is_decided_by_counted_work_and_the_image_does_not_change(monkeypatch, capfd):
    """The streaming schedule starts a render on the default deal of 8 x 8-tile screen regions per XCD and keeps a deal of 16 x 16-tile
    regions (better L2 locality on frames of even cost) beside it.  During the first call the kernel adds every finished path's length to
    its tile's sum, and after it the library takes the large regions iff the XCDs' shares of that COUNTED work under them are within
    ER_STREAM_COST_SPREAD_MAX of each other (csrc/er_api.cpp er_stream_adapt, csrc/er_stream.h): a decision made from counts -- the
    same on every run of the same frame, unlike round 4's, which hung on the XCDs' measured finish times.  A soup seen from far away --
    geometry in the middle of the frame, sky around it -- is a frame of uneven cost: the verbose line must say that the default deal
    stays, with the same figure on a second run; with the limit raised the large regions are taken after the first call; and the planes
    equal those of both fixed deals bit for bit (any deal renders the same pixels)."""
    sc = scenes.soup(60_000, 1280, 832, seed=31, hdri_size=(256, 128))
    sc.camera.position = abi.ErVec3(0.01, 0.02, -3.0)
    sc._desc = None
    monkeypatch.setenv("ER_STREAM_SUPER_TILE", "8")
    fixed8 = gpu_render(sc, 12, max_bounces=8, flags=abi.FLAG_STREAM, chunks=[8, 2, 2])
    monkeypatch.setenv("ER_STREAM_SUPER_TILE", "16")
    fixed16 = gpu_render(sc, 12, max_bounces=8, flags=abi.FLAG_STREAM, chunks=[8, 2, 2])
    monkeypatch.delenv("ER_STREAM_SUPER_TILE")
    monkeypatch.setenv("ER_STREAM_VERBOSE", "1")
    outs, figures = [], []
    for limit in (None, None, "1e9"):
        if limit is not None:
            monkeypatch.setenv("ER_STREAM_COST_SPREAD_MAX", limit)
        capfd.readouterr()
        outs.append(gpu_render(sc, 12, max_bounces=8, flags=abi.FLAG_STREAM, chunks=[8, 2, 2]))
        lines = [l for l in capfd.readouterr().err.splitlines() if l.startswith("[er_stream] counted work")]
        assert len(lines) == 1, lines                      # decided once, after the first call
        figures.append(float(lines[0].split(":")[1].split("of the mean")[0]))
        assert ("-> large regions" in lines[0]) == (limit is not None), lines
        assert ("the default deal stays" in lines[0]) == (limit is None), lines
    assert figures[0] == figures[1] == figures[2] and figures[0] > 0.1, figures      # counted, not timed: the same figure every time
    # Round 6: a render that is ONE call decides too -- the call's first sample is a launch of its own, the decision
    # follows it (the same figure: the work counted is that first pass's in both cases) and the other samples run on the deal decided.
    for limit in (None, "1e9"):
        if limit is None:
            monkeypatch.delenv("ER_STREAM_COST_SPREAD_MAX", raising=False)
        else:
            monkeypatch.setenv("ER_STREAM_COST_SPREAD_MAX", limit)
        capfd.readouterr()
        outs.append(gpu_render(sc, 12, max_bounces=8, flags=abi.FLAG_STREAM, chunks=[12]))
        err = capfd.readouterr().err.splitlines()
        lines = [l for l in err if l.startswith("[er_stream] counted work")]
        assert len(lines) == 1 and float(lines[0].split(":")[1].split("of the mean")[0]) == figures[0], lines
        assert ("-> large regions" in lines[0]) == (limit is not None), lines
    for other in [fixed16] + outs:
        for p in ("beauty", "normal", "tangent", "bitangent"):
            assert (fixed8[p].view(np.uint32) == other[p].view(np.uint32)).all(), p
        assert (fixed8["rng"] == other["rng"]).all() and (fixed8["samples"] == other["samples"]).all()


def test_scalar_only_textures_kept_with_one_channel_do_not_change_the_image(oracle_mod, monkeypatch):
    """A texture that materials use for scalar channels only -- opacity, roughness, metallic, transmission take `.x` of the fetched value
    (src/kernel.cpp:100-150) -- is kept on the device with its first channel alone (csrc/er_api.cpp; C5's texture pool: 151 -> 84 MB).
    The image must be that of the textures as they came (ER_TEX_COMPACT=0) and the oracle's, bit for bit, on a scene that has all three
    kinds: textures used as colours only, as scalars only (3- and 2-channel, filtered and not), and as both (kept whole)."""
    sc = scenes.torture(3000, 96, 64, seed=5, n_materials=6, tex_size=16, hdri_size=(64, 32), n_lights=0)
    r = scenes.Rand(91, 2)
    sc.textures.append((abi._f32(r.u01(8, 8, 2)), 8, 8, 2, 1))          # a 2-channel bilinear texture, used as opacity below
    sc.materials[0].opacity_tex = len(sc.textures) - 1
    sc.materials[1].roughness_tex = sc.materials[1].albedo_tex             # one texture as a colour AND as a scalar: stays whole
    sc.materials[2].transmission_tex = sc.materials[3].metallic_tex        # a scalar-only texture shared by two materials: as metallic AND as transmission
    d, w, h, ch, _ = sc.textures[sc.materials[4].roughness_tex]
    sc.textures[sc.materials[4].roughness_tex] = (d, w, h, ch, 1)           # a BILINEAR roughness texture beside an unfiltered albedo: material 4 is not fused
    for tid in (sc.materials[5].albedo_tex, sc.materials[5].roughness_tex, sc.materials[5].metallic_tex):
        d, w, h, ch, _ = sc.textures[tid]
        sc.textures[tid] = (d, w, h, ch, 1)                                 # material 5: all three BILINEAR -> fused, filtered on five channels at once
    sc._desc = None
    # (materials whose albedo / roughness / metallic textures share size and filter are fetched from ONE fused texel record, DevFused in
    # csrc/er_device.h: here materials 0, 2, 3 unfiltered, 1 with one texture in two roles, 5 bilinear; ER_TEX_COMPACT=0 switches it off too)
    # (the unfiltered textures that are read only as roughness or metallic are also held to the power 2.2 that generateHitData takes of
    # every fetch, src/kernel.cpp:152-153 -- not the bilinear one, whose power is of the FILTERED value, nor the one that is also a
    # transmission, which takes none; ER_TEX_COMPACT=0 switches both off)
    compact = gpu_render(sc, 5, max_bounces=8, flags=abi.FLAG_STREAM)
    monkeypatch.setenv("ER_TEX_COMPACT", "0")
    as_is = gpu_render(sc, 5, max_bounces=8, flags=abi.FLAG_STREAM)
    monkeypatch.delenv("ER_TEX_COMPACT")
    for p in ("beauty", "normal", "tangent", "bitangent"):
        assert (compact[p].view(np.uint32) == as_is[p].view(np.uint32)).all(), p
    assert (compact["rng"] == as_is["rng"]).all()
    compare(compact, oracle_render(oracle_mod, sc, 5, max_bounces=8), what="scalar-only textures with one channel")


@pytest.mark.parametrize("waves", ["12", "16"])
@pytest.mark.parametrize("ext", [0, abi.FLAG_POINT_LIGHTS | abi.FLAG_MIS])
def test_speculative_samples_change_neither_the_image_nor_any_count(waves, ext, monkeypatch):
    """Round 6: on shares of few pixels per slot the streaming kernel starts a pixel's NEXT sample beside the one in flight, from the RNG
    state that one will leave if it draws as many numbers as the pixel's samples have been drawing (csrc/er_stream.hip, ST_DRAWS_MASK).  A
    speculative sample is accumulated only after its predecessor, and only if the predecessor left exactly the guessed state; otherwise
    it is dropped and the pixel's next sample starts from the true state.  So the planes, the sample counts, the RNG states AND every
    event counter -- paths, bounce-loop iterations (the metric's unit), rays, shaded hits, HDRI samples, and with ER_FLAG_COUNTERS node
    visits and triangle tests -- must equal those of the wavefront schedule, which knows no speculation; and the guesses must have
    happened: both right and wrong ones (a frame of 40 x 30 tiles on 256 workgroups: 300 pixels per CU, 724 free slots each)."""
    sc = scenes.soup(30000, 320, 240, seed=17, hdri_size=(128, 64))
    if ext:
        sc.point_lights = scenes.point_lights(12, seed=4)
        sc._desc = None
    monkeypatch.setenv("ER_STREAM_WAVES", waves)
    monkeypatch.setenv("ER_STREAM_SPEC_FORM", "1")
    for count in (0, abi.FLAG_COUNTERS):
        w = gpu_render(sc, 24, max_bounces=8, flags=abi.FLAG_WAVEFRONT | ext | count)
        rm = render.RenderingManager(render.RenderParameters(max_bounces=8, flags=abi.FLAG_STREAM | ext | count))
        rm.start_rendering(sc)
        for n in (5, 19):
            rm.render(n)
        si = rm.stream_info()
        s = {p: rm.get_pass(p) for p in ("beauty", "normal", "tangent", "bitangent")}
        s["rng"], s["samples"], s["counters"] = rm.read_rng(), rm.read_samples(), rm.counters()
        rm.close()
        assert si["waves"] == int(waves)
        print(f"waves {waves} ext {ext} counters {count}: speculative samples started {si['spec_started']}, right {si['spec_right']}, wrong {si['spec_wrong']}")
        assert si["spec_right"] > 1000 and si["spec_wrong"] > 1000 and si["spec_started"] >= si["spec_right"] + si["spec_wrong"]
        for p in ("beauty", "normal", "tangent", "bitangent"):
            assert (w[p].view(np.uint32) == s[p].view(np.uint32)).all(), p
        assert (w["rng"] == s["rng"]).all() and (w["samples"] == s["samples"]).all()
        keys = ["paths", "bounce_samples", "rays", "shaded_hits", "hdri_samples"] + (["node_visits", "tri_tests", "texel_fetches"] if count else [])
        for k in keys:
            assert w["counters"][k] == s["counters"][k], (k, w["counters"][k], s["counters"][k])


@pytest.mark.parametrize("ext", [0, abi.FLAG_POINT_LIGHTS | abi.FLAG_MIS])
def test_pixels_behind_keep_their_slots_and_the_image_does_not_change(ext, monkeypatch):
    """Round 6: where a workgroup's pixels outnumber its slots (the kernel's forms 1 and 2 at 16 waves: a half ... a sixth of a 1080p frame per GPU, a 720p frame) a pixel
    that is two samples behind the workgroup's most advanced one goes on in the slot it has instead of queueing in the pixel ring
    (csrc/er_stream.hip s_front), so that the expensive pixels are not what the launch ends on.  Which slot renders a sample and when is
    nobody's business: planes, sample counts, RNG states and every event counter (with ER_FLAG_COUNTERS also node visits, triangle tests and
    texel fetches) equal the wavefront schedule's, in both forms, with the rule on (the product), off, and with a slack of one sample -- on a
    frame of 2 560 pixels per workgroup (2.5 per slot), where the ring is never empty until the launch's end."""
    sc = scenes.soup(30000, 1024, 640, seed=23, hdri_size=(128, 64))
    if ext:
        sc.point_lights = scenes.point_lights(12, seed=4)
        sc._desc = None
    for form, keep, count in ((1, "2", 0), (1, "2", abi.FLAG_COUNTERS), (1, "1", 0), (2, "2", 0), (2, "0", 0), (2, "1", abi.FLAG_COUNTERS)):
        # (form 1 is what a share of this size gets by itself; form 2 -- the rule beside speculative samples -- is forced here: a share of 2 304 pixels per
        # CU or fewer gets it by itself)
        w = gpu_render(sc, 13, max_bounces=8, flags=abi.FLAG_WAVEFRONT | ext | count)
        if form == 2:
            monkeypatch.setenv("ER_STREAM_SPEC_FORM", "1")
        else:
            monkeypatch.delenv("ER_STREAM_SPEC_FORM", raising=False)
        monkeypatch.setenv("ER_STREAM_SPEC_KEEP", keep)
        rm = render.RenderingManager(render.RenderParameters(max_bounces=8, flags=abi.FLAG_STREAM | ext | count))
        rm.start_rendering(sc)
        for n in (4, 9):
            rm.render(n)
        si = rm.stream_info()
        s = {p: rm.get_pass(p) for p in ("beauty", "normal", "tangent", "bitangent")}
        s["rng"], s["samples"], s["counters"] = rm.read_rng(), rm.read_samples(), rm.counters()
        rm.close()
        assert si["waves"] == 16 and si["pixels_per_cu"] > 2304 and si["form"] == form, si
        for p in ("beauty", "normal", "tangent", "bitangent"):
            assert (w[p].view(np.uint32) == s[p].view(np.uint32)).all(), (form, keep, p)
        assert (w["rng"] == s["rng"]).all() and (w["samples"] == s["samples"]).all()
        for k in ["paths", "bounce_samples", "rays", "shaded_hits", "hdri_samples"] + (["node_visits", "tri_tests", "texel_fetches"] if count else []):
            assert w["counters"][k] == s["counters"][k], (form, keep, k, w["counters"][k], s["counters"][k])


def test_a_share_of_a_1080p_frame_gets_the_form_made_for_its_size():
    """Round 6 (csrc/er_api.cpp, er_stream.h): the streaming kernel has three forms and two wave counts, picked by a rank's owned pixels per CU --
    the whole frame (8 100): form 0, 16 waves, 13 tracers (round 5's code); a half (4 050): form 1 (pixels that are behind keep their slots); a
    quarter (2 025): form 2 (that and speculative samples), 16 waves; an eighth (1 012): form 2 in 12 waves, 10 of them tracers; a sixteenth
    (506): 9 tracers.  A scene of fewer than 1 000 triangles starts no speculative samples whatever its share."""
    sc = scenes.soup(20000, 1920, 1080, seed=3, hdri_size=(64, 32))
    want = {1: (0, 16, 13), 2: (1, 16, 13), 4: (2, 16, 13), 8: (2, 12, 10), 16: (2, 12, 9)}
    for world, (form, waves, tracers) in want.items():
        rm = render.RenderingManager(render.RenderParameters(max_bounces=3, flags=abi.FLAG_STREAM, rank=0, world=world))
        rm.start_rendering(sc)
        rm.render(2)
        si = rm.stream_info()
        rm.close()
        assert (si["form"], si["waves"], si["tracers"]) == (form, waves, tracers), (world, si)
    rm = render.RenderingManager(render.RenderParameters(max_bounces=3, flags=abi.FLAG_STREAM, rank=0, world=8))
    rm.start_rendering(scenes.cornell(1920, 1080))
    rm.render(2)
    si = rm.stream_info()
    rm.close()
    assert si["form"] == 0 and si["spec_started"] == 0, si
