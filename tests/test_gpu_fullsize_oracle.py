"""The HIP path against the oracle ON THE BASELINE SCENES THEMSELVES.

Every other oracle comparison runs on scenes of at most 131 072 triangles; the reference's fixed-depth-18 tree
(src/BVH.h:47, src/BVH.cpp:63-120) is a different regime at 1 M triangles (3.8 triangles per leaf) and at 10 M (38), and the
product's own tree, its LDS-resident top levels and its pixel rings have their real sizes only here.  Per configuration:

  (a) the WHOLE frame at the real resolution, default schedule (what bench.py times), 2 spp; a window of 8 whole rows spread
      over the frame (15 360 pixels at 1080p, 30 720 at 4K) must equal the oracle's render of those rows bit for bit in all
      five planes, the sample counts and the RNG states (src/kernel.cpp:477-646);
  (b) one rank's share of a 128-way tile split (about 16 k pixels, 8x8 tiles spread over the whole frame) in the streaming
      schedule: planes as above AND the event counters -- paths, bounce_samples (the metric's unit), shaded hits, HDRI
      samples -- equal to the oracle's over exactly those pixels.

C4 also checks the window against tests/golden/c4_fullsize_rows.npz, generated in the build container by
tests/golden/make_golden_fullsize.py (the oracle's output: parity unpinned like every vector here).
"""
import os

import numpy as np
import pytest

from elevenrender_amd import abi, render
from elevenrender_amd import dist as erdist
from fullsize_util import PLANE_NAMES, SPP, OracleSession, config_scene, scene_digest, window_rows

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _gpu(sc, max_bounces, flags, rank=0, world=1):
    rm = render.RenderingManager(render.RenderParameters(max_bounces=max_bounces, flags=flags, rank=rank, world=world))
    rm.start_rendering(sc)
    rm.render(SPP)
    out = {p: rm.get_pass(p) for p in PLANE_NAMES}
    out["samples"] = rm.read_samples().reshape(sc.y_res, sc.x_res)
    out["rng"] = rm.read_rng().reshape(sc.y_res, sc.x_res)
    out["counters"] = rm.counters()
    out["schedule"] = rm.profile()["schedule"]
    rm.close()
    return out


def _differing(g, o, sel):
    """number of selected pixels that differ in any plane, the sample count or the RNG state; g indexed by `sel`."""
    bad = np.zeros(o["samples"].shape, bool)
    for p in PLANE_NAMES:
        bad |= (sel(g[p]).view(np.uint32) != o[p].view(np.uint32)).any(-1)
    bad |= sel(g["samples"]) != o["samples"]
    bad |= sel(g["rng"]) != o["rng"]
    return int(bad.sum()), bad.size


def _check_config(oracle_mod, name, golden=None):
    sc, mb, ext = config_scene(name)
    W = sc.x_res
    rows = window_rows(sc.y_res)
    row_idx = (np.array(rows, np.int64)[:, None] * W + np.arange(W, dtype=np.int64)[None, :])
    rank, world = 37, 128
    tile_idx = erdist.tile_pixel_index(erdist.owned_tiles(rank, world, sc.x_res, sc.y_res), sc.x_res, sc.y_res)
    tile_idx = np.sort(tile_idx[tile_idx >= 0])
    # the oracle: ONE reference-style build; the shard's pixels first (their counters), then the rest of the rows
    ses = OracleSession(oracle_mod, sc, mb, ext)
    oc = ses.render(tile_idx)
    ses.render(row_idx)
    o, ot = ses.read(row_idx), ses.read(tile_idx)
    ses.close()
    flat = lambda a: a.reshape((-1,) + a.shape[2:])      # [H, W, ...] -> [H*W, ...]
    # (a) whole frame, default schedule, against the oracle on whole rows
    g = _gpu(sc, mb, ext)
    assert g["schedule"] == abi.FLAG_STREAM, "the default schedule at this size is the streaming one"
    assert g["counters"]["paths"] == sc.x_res * sc.y_res * SPP
    n_bad, n = _differing(g, o, lambda a: flat(a)[row_idx])
    lit = (o["beauty"][..., :3] > 0).any(-1).mean()
    print(f"{name}: rows {rows}: {n - n_bad} of {n} pixels bit-exact in 5 planes + samples + rng; {lit:.3f} of them lit; "
          f"oracle build {ses.build_seconds:.1f} s; in the reference tree {oc['node_visits'] / max(1, oc['rays']):.0f} node visits + "
          f"{oc['tri_tests'] / max(1, oc['rays']):.0f} triangle tests per ray")
    assert n_bad == 0, f"{name}: {n_bad} of {n} window pixels differ from the oracle"
    assert (o["samples"] == SPP + 1).mean() > 0.999 and lit > 0.2
    if golden is not None:
        z = np.load(golden)
        assert str(z["scene_sha256"]) == scene_digest(sc), "the scene generator produced other inputs than the golden file was made for"
        assert list(z["rows"]) == rows and int(z["spp"]) == SPP and int(z["max_bounces"]) == mb
        zo = {p: z[p] for p in PLANE_NAMES}
        zo["samples"], zo["rng"] = z["samples"], z["rng"]
        n_bad_golden, _ = _differing(g, zo, lambda a: flat(a)[row_idx])
        assert n_bad_golden == 0, f"{name}: {n_bad_golden} window pixels differ from the committed golden rows"
        for p in PLANE_NAMES:      # the oracle on this box == the oracle in the build container
            assert (zo[p].view(np.uint32) == o[p].view(np.uint32)).all(), p
    # (b) one rank's tiles of a 128-way split, streaming schedule: planes and event counters
    w = _gpu(sc, mb, ext | abi.FLAG_STREAM, rank=rank, world=world)
    n_bad, n = _differing(w, ot, lambda a: flat(a)[tile_idx])
    assert n_bad == 0, f"{name}: {n_bad} of {n} pixels of rank {rank}/{world} differ from the oracle"
    for p in PLANE_NAMES:      # ... and the whole-frame render agrees with the shard on those pixels
        assert (flat(g[p])[tile_idx].view(np.uint32) == flat(w[p])[tile_idx].view(np.uint32)).all(), p
    # (not `rays`: the oracle traces every shadow ray the reference traces, src/kernel.cpp:555-562; the product skips a shadow query whose
    # contribution is exactly zero -- DisneyEval = 0 below the horizon -- because its verdict cannot change the sum)
    keys = ("paths", "bounce_samples", "shaded_hits", "hdri_samples")
    for k in keys:
        assert w["counters"][k] == oc[k], (name, k, w["counters"][k], oc[k])
    print(f"{name}: rank {rank}/{world}: {n} pixels bit-exact; counters equal: " + ", ".join(f"{k} {oc[k]}" for k in keys))


def test_c2_full_size_against_the_oracle(oracle_mod):
    """BASELINE config 2: 1M-triangle soup + 2048x1024 HDRI, 1920x1080, 8 bounces."""
    _check_config(oracle_mod, "C2")


def test_c5_full_size_against_the_oracle_reference_behaviour(oracle_mod):
    """BASELINE config 5, reference behaviour (point lights ignored, no MIS): 64 textured materials, 16 bounces."""
    _check_config(oracle_mod, "C5")


def test_c5_full_size_against_the_oracle_lights_and_mis(oracle_mod):
    """BASELINE config 5 in full: 256 point lights + MIS (build-defined extensions: the oracle side is their mirror)."""
    _check_config(oracle_mod, "C5lit")


def test_c4_full_size_against_the_oracle_and_golden_rows(oracle_mod):
    """BASELINE config 4: 10M triangles (10 000 x 1 000-triangle smooth blobs), 3840x2160, 8 bounces."""
    _check_config(oracle_mod, "C4", golden=os.path.join(GOLDEN, "c4_fullsize_rows.npz"))
