"""Shared by tests/test_gpu_fullsize_oracle.py and tests/golden/make_golden_fullsize.py: the BASELINE scenes at their
real size, the window of whole rows that is compared, and the oracle render of that window."""
import hashlib

import numpy as np

from elevenrender_amd import abi, scenes

PLANE_NAMES = ("beauty", "denoise", "normal", "tangent", "bitangent")
SPP = 2
N_ROWS = 8


def window_rows(y_res, n_rows=N_ROWS):
    """`n_rows` whole rows spread over the frame (the centres of n_rows equal bands)."""
    return [int((2 * k + 1) * y_res // (2 * n_rows)) for k in range(n_rows)]


def config_scene(name):
    """-> (scene, max_bounces, extension flags) of BASELINE.json configs[1] / [3] / [4] at full size."""
    if name == "C2":
        return scenes.soup(1_000_000, 1920, 1080, seed=12345), 8, 0
    if name == "C4":
        return scenes.blob_instances(), 8, 0
    if name == "C5":
        return scenes.torture(1_000_000, 1920, 1080, seed=12345), 16, 0
    if name == "C5lit":
        return scenes.torture(1_000_000, 1920, 1080, seed=12345), 16, abi.FLAG_POINT_LIGHTS | abi.FLAG_MIS
    raise KeyError(name)


def scene_digest(sc):
    """sha256 over the geometry arrays: a golden window is only meaningful for bit-identical inputs."""
    h = hashlib.sha256()
    for a in (sc.vertices, sc.normals, sc.tangents, sc.uvs, sc.tangent_sign, sc.material_id):
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def oracle_window(oracle_mod, sc, max_bounces, flags, rows, spp=SPP, threads=16):
    """The oracle (er_math mode, reference-style fixed-depth-18 tree) on whole rows `rows` of the frame for `spp` samples.
    -> dict of [n_rows, W, 4] planes + samples / rng [n_rows, W] + counters + build seconds."""
    W = sc.x_res
    o = oracle_mod.Oracle(sc, math_mode=oracle_mod.MATH_ER, max_bounces=max_bounces, threads=threads, flags=flags)
    for y in rows:
        o.render(spp, y * W, (y + 1) * W)
    out = {name: np.ascontiguousarray(o.read_pass(abi.PASS_NAMES[name])[rows]) for name in PLANE_NAMES}
    out["samples"] = o.read_samples().reshape(sc.y_res, W)[rows].copy()
    out["rng"] = o.read_rng().reshape(sc.y_res, W)[rows].copy()
    out["counters"] = o.counters()
    out["build_seconds"] = o.build_seconds
    o.close()
    return out


class OracleSession:
    """One oracle (one reference-style build) that renders disjoint pixel sets one after the other."""

    def __init__(self, oracle_mod, sc, max_bounces, flags, threads=16):
        self.sc = sc
        self.o = oracle_mod.Oracle(sc, math_mode=oracle_mod.MATH_ER, max_bounces=max_bounces, threads=threads, flags=flags)
        self.done = np.zeros(sc.x_res * sc.y_res, bool)
        self.build_seconds = self.o.build_seconds

    def render(self, pixel_idx, spp=SPP):
        """renders the pixels of `pixel_idx` not rendered before (runs of consecutive indices); -> the counters' increase"""
        idx = np.unique(np.asarray(pixel_idx, np.int64))
        idx = idx[~self.done[idx]]
        c0 = self.o.counters()
        if idx.size:
            for run in np.split(idx, np.nonzero(np.diff(idx) != 1)[0] + 1):
                self.o.render(spp, int(run[0]), int(run[-1]) + 1)
        self.done[idx] = True
        c1 = self.o.counters()
        return {k: c1[k] - c0[k] for k in c1}

    def read(self, pixel_idx):
        """per-pixel planes / samples / rng at `pixel_idx` (any shape of index array)"""
        idx = np.asarray(pixel_idx, np.int64)
        assert self.done[idx].all()
        out = {name: self.o.read_pass(abi.PASS_NAMES[name]).reshape(-1, 4)[idx] for name in PLANE_NAMES}
        out["samples"] = self.o.read_samples()[idx]
        out["rng"] = self.o.read_rng()[idx]
        return out

    def close(self):
        self.o.close()
