"""ER_FLAG_POINT_LIGHTS / ER_FLAG_MIS on the GPU against the oracle's mirror of the same rules (csrc/er_shade.h).

Both are build-defined extensions (the reference's pointLight() has no caller and defines no
result; its MIS weights are computed and never applied) -- PARITY UNPINNED against the reference by construction.
What these tests pin is: GPU == oracle bit for bit in every schedule, flags off == no lights at all, and the three
schedules against each other at BASELINE config 5's full size with its 256 lights and MIS on.
"""
import numpy as np
import pytest

from elevenrender_amd import abi, scenes
from test_gpu_parity import compare, gpu_render, oracle_render

pytestmark = pytest.mark.gpu

SCHEDULES = [abi.FLAG_WAVEFRONT, abi.FLAG_MEGAKERNEL, abi.FLAG_STREAM]
EXT = [abi.FLAG_POINT_LIGHTS, abi.FLAG_MIS, abi.FLAG_POINT_LIGHTS | abi.FLAG_MIS]


def lit_cornell(x_res=64, y_res=48):
    sc = scenes.cornell(x_res, y_res)
    sc.point_lights = scenes.point_lights(5, seed=3, lo=(-0.8, -0.8, 2.2), hi=(0.8, 0.8, 3.8))
    sc._desc = None
    return sc


def lit_soup(n=3000, x_res=72, y_res=56, n_lights=7):
    sc = scenes.soup(n, x_res, y_res, seed=17, hdri_size=(64, 32))
    sc.point_lights = scenes.point_lights(n_lights, seed=5)
    sc._desc = None
    return sc


@pytest.mark.parametrize("ext", EXT)
@pytest.mark.parametrize("sched", SCHEDULES)
def test_lit_scenes_bit_exact_against_the_oracle(oracle_mod, sched, ext):
    for name, sc, spp, mb in (("cornell", lit_cornell(), 6, 5), ("soup", lit_soup(), 5, 8),
                              ("textured", scenes.torture(3000, 64, 48, seed=5, n_materials=8, tex_size=16, hdri_size=(128, 64), n_lights=16), 3, 16),
                              ("blobs", None, 4, 8)):
        if sc is None:
            sc = scenes.blob_instances(n_instances=30, tris_per_blob=300, x_res=64, y_res=48, grid=(5, 3, 2), spacing=0.45)
            sc.point_lights = scenes.point_lights(6, seed=9, lo=(-1.2, -0.8, 1.5), hi=(1.2, 0.8, 3.0))
            sc._desc = None
        g = gpu_render(sc, spp, max_bounces=mb, flags=sched | ext)
        o = oracle_mod.Oracle(sc, math_mode=1, max_bounces=mb, threads=8, flags=ext)
        o.render(spp)
        ref = {n: o.read_pass(p) for n, p in abi.PASS_NAMES.items()}
        ref["samples"], ref["rng"], ref["counters"] = o.read_samples(), o.read_rng(), o.counters()
        o.close()
        compare(g, ref, what=f"{name} sched={sched} ext={ext}")
        assert g["counters"]["bounce_samples"] == ref["counters"]["bounce_samples"]


@pytest.mark.parametrize("sched", SCHEDULES)
def test_lights_change_the_image_only_with_the_flag(sched):
    """Lights in the descriptor but no flag == no lights (reference behaviour: they are copied and ignored,
    src/SYCLCopy.cpp:64-66); the flag with an empty light list == no flag; with lights the image gets brighter."""
    sc0 = scenes.soup(3000, 72, 56, seed=17, hdri_size=(64, 32))
    sc1 = lit_soup()
    a = gpu_render(sc0, 4, max_bounces=8, flags=sched)
    b = gpu_render(sc1, 4, max_bounces=8, flags=sched)
    c = gpu_render(sc0, 4, max_bounces=8, flags=sched | abi.FLAG_POINT_LIGHTS)
    d = gpu_render(sc1, 4, max_bounces=8, flags=sched | abi.FLAG_POINT_LIGHTS)
    for p in ("beauty", "normal", "tangent", "bitangent"):
        assert (a[p].view(np.uint32) == b[p].view(np.uint32)).all()
        assert (a[p].view(np.uint32) == c[p].view(np.uint32)).all()
    assert (a["rng"] == b["rng"]).all() and (a["rng"] == c["rng"]).all()
    assert (d["rng"] != a["rng"]).mean() > 0.5            # one extra draw per opaque bounce
    assert d["beauty"][..., :3].mean() > a["beauty"][..., :3].mean()
    assert d["counters"]["rays"] > a["counters"]["rays"]


def test_c5_full_size_with_its_256_lights_and_mis():
    """BASELINE config 5 in full: 1M triangles, 64 textured materials, 256 point lights, MIS on, 16 bounces, 1920x1080.
    Size-independent properties: the three schedules bit for bit on a window of tiles, chunked == one call."""
    from test_gpu_parity import _window_schedules_agree
    sc = scenes.torture(1_000_000, 1920, 1080, seed=12345)
    assert len(sc.point_lights) == 256
    ext = abi.FLAG_POINT_LIGHTS | abi.FLAG_MIS
    w = _window_schedules_agree(sc, 2, 16, rank=5, world=48, extra_flags=ext)
    c = gpu_render(sc, 2, max_bounces=16, rank=5, world=48, chunks=[1, 1], flags=ext)
    assert (w["beauty"].view(np.uint32) == c["beauty"].view(np.uint32)).all()
    assert np.isfinite(w["beauty"]).all() and w["beauty"][..., :3].max() <= 10
    plain = gpu_render(sc, 2, max_bounces=16, rank=5, world=48)
    assert w["counters"]["rays"] > plain["counters"]["rays"]           # the light queries
    owned = plain["samples"].reshape(plain["beauty"].shape[:2]) > 1
    assert (w["beauty"].view(np.uint32) != plain["beauty"].view(np.uint32)).any(-1)[owned].mean() > 0.2
