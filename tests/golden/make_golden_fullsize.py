#!/usr/bin/env python3
"""Generates tests/golden/c4_fullsize_rows.npz: the ORACLE's render (er_math mode, reference-style fixed-depth-18 tree over
all 10 M triangles) of 8 whole rows of BASELINE config 4 at 3840x2160, 2 spp, 8 bounces -- the window that
tests/test_gpu_fullsize_oracle.py compares the HIP path with on the GPU box.

A regression vector of this repository's oracle, not a reference output (the reference cannot be built here and ships no
vectors: DESIGN.md 1, parity unpinned).  The scene is NOT stored (1.4 GB of arrays); its generator is committed
(elevenrender_amd/scenes.py blob_instances) and the file carries the sha256 of the arrays it was made for.
Usage: python tests/golden/make_golden_fullsize.py      (about a minute: one 16 s reference-style build + 8 rows)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle  # noqa: E402
from fullsize_util import PLANE_NAMES, SPP, config_scene, oracle_window, scene_digest, window_rows  # noqa: E402


def main():
    sc, mb, ext = config_scene("C4")
    rows = window_rows(sc.y_res)
    o = oracle_window(oracle, sc, mb, ext, rows, threads=os.cpu_count() or 8)
    out = {p: o[p] for p in PLANE_NAMES}
    out.update(samples=o["samples"], rng=o["rng"], rows=np.array(rows, np.int64), spp=np.int64(SPP), max_bounces=np.int64(mb),
               scene_sha256=np.array(scene_digest(sc)), tri_count=np.int64(sc.tri_count),
               bounce_samples=np.int64(o["counters"]["bounce_samples"]), rays=np.int64(o["counters"]["rays"]))
    path = os.path.join(HERE, "c4_fullsize_rows.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes;", o["counters"])


if __name__ == "__main__":
    main()
