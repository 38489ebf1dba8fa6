"""Checkpoint / resume (include/eleven_hip.h: er_state_size / er_state_export / er_state_import).
The reference keeps the progressive estimator only in device memory; passes + sample counts + RNG state are the whole
resumable state, so a render continued from a snapshot must equal the uninterrupted one bit for bit -- in another scene
object, in another schedule, and when the continuing run is tile-sharded."""
import ctypes as C

import numpy as np
import pytest

from elevenrender_amd import abi, render, scenes
from test_gpu_parity import gpu_render

pytestmark = pytest.mark.gpu


def test_resume_from_a_snapshot_equals_the_uninterrupted_render():
    sc = scenes.soup(4000, 88, 60, seed=31, hdri_size=(64, 32))
    whole = gpu_render(sc, 7, max_bounces=8, flags=abi.FLAG_WAVEFRONT)
    a = render.RenderingManager(render.RenderParameters(max_bounces=8, flags=abi.FLAG_WAVEFRONT))
    a.start_rendering(sc)
    a.render(3, blocking=False)            # no wait: the export is ordered after the samples
    snap = a.state_export()
    a.close()
    assert snap[:8].tobytes() == b"ERSTATE2" and snap.size == 64 + 88 * 60 * (5 * 16 + 8)
    for flags in (abi.FLAG_WAVEFRONT, abi.FLAG_MEGAKERNEL, abi.FLAG_STREAM):       # resume in any schedule
        b = render.RenderingManager(render.RenderParameters(max_bounces=8, flags=flags))
        b.start_rendering(sc)
        b.state_import(snap)
        assert b.get_render_info().samples == 4
        b.render(4)
        for p in ("beauty", "normal", "tangent", "bitangent"):
            assert (b.get_pass(p).view(np.uint32) == whole[p].view(np.uint32)).all(), (flags, p)
        assert (b.read_rng() == whole["rng"]).all() and (b.read_samples() == whole["samples"]).all()
        b.close()
    # a tile-sharded rank continues its own pixels from the same snapshot
    r = render.RenderingManager(render.RenderParameters(max_bounces=8, rank=1, world=3))
    r.start_rendering(sc)
    r.state_import(snap)
    r.render(4)
    owned = r.read_samples().reshape(60, 88) == 8
    assert owned.sum() > 0 and (r.get_pass("beauty")[owned].view(np.uint32) == whole["beauty"][owned].view(np.uint32)).all()
    r.close()


def test_snapshot_validation():
    lib = abi.load()
    sc = scenes.cornell(32, 24)
    rm = render.RenderingManager(render.RenderParameters())
    rm.start_rendering(sc)
    rm.render(1)
    snap = rm.state_export()
    bad = snap.copy()
    bad[0] = 0
    with pytest.raises(abi.ErError) as e:
        rm.state_import(bad)
    assert e.value.code == abi.ER_ERR_INVALID_ARG and "magic" in str(e.value)
    with pytest.raises(abi.ErError):
        rm.state_import(snap[:100])
    other = render.RenderingManager(render.RenderParameters())
    other.start_rendering(scenes.cornell(16, 24))
    with pytest.raises(abi.ErError) as e:
        other.state_import(snap)
    assert "32x24" in str(e.value)
    n = C.c_uint64()
    assert lib.er_state_export(rm.handle, snap.ctypes.data_as(C.c_void_p), 10) == abi.ER_ERR_INVALID_ARG
    other.close()
    rm.close()
