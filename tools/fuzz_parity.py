"""Randomised parity run on an MI355X: random small scenes (soup / textured soup / smooth blobs / Cornell), random
frame sizes, bounce limits and sample counts; every schedule (the streaming one included) and both acceleration-structure builders against the
oracle, bit for bit; half of the textured scenes with textures and an HDRI whose sides are not powers of two.  Usage: python tools/fuzz_parity.py [cases] [seed].  Prints MISMATCH lines and a summary.
Known source of single-pixel mismatches: an exact distance tie between two triangles, e.g. a ray
through a seam of the Cornell box."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np

import oracle
from elevenrender_amd import abi, scenes
from test_gpu_parity import gpu_render, oracle_render

FLAG_SETS = (abi.FLAG_WAVEFRONT, abi.FLAG_MEGAKERNEL, abi.FLAG_STREAM, abi.FLAG_WAVEFRONT | abi.FLAG_GPU_BUILD, abi.FLAG_STREAM | abi.FLAG_GPU_BUILD,
             abi.FLAG_STREAM | abi.FLAG_HOST_BUILD, abi.FLAG_WAVEFRONT | abi.FLAG_HOST_BUILD)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
bad, t0 = 0, time.time()
for it in range(cases):
    kind, seed = it % 4, int(rng.integers(1, 1 << 30))
    w, h = int(rng.integers(9, 90)), int(rng.integers(7, 70))
    mb, spp = int(rng.choice([1, 2, 5, 8, 16])), int(rng.integers(1, 6))
    if kind == 0:
        sc = scenes.soup(int(rng.integers(1, 40000)), w, h, seed=seed, hdri_size=(int(rng.choice([16, 64, 256])), int(rng.choice([8, 32, 128]))))
    elif kind == 1:
        sc = scenes.torture(int(rng.integers(50, 8000)), w, h, seed=seed, n_materials=int(rng.integers(1, 12)), tex_size=int(rng.choice([4, 16, 32])), hdri_size=(64, 32))
        if rng.integers(0, 2):      # textures and HDRI without power-of-two sides: the general wrap (a signed % by the run-time side) instead of the mask
            r = scenes.Rand(seed, 5)
            tw, th = int(rng.integers(3, 40)), int(rng.integers(3, 40))
            sc.textures = [(abi._f32(r.u01(th, tw, 3)), tw, th, 3, int(rng.integers(0, 2))) for _ in sc.textures]
            hw, hh = int(rng.integers(5, 90)), int(rng.integers(3, 50))
            sc.hdri = (abi._f32(0.2 + 2.0 * r.u01(hh, hw, 3)), hw, hh, 3, 0)
            sc._desc = None
    elif kind == 2:
        sc = scenes.blob_instances(n_instances=int(rng.integers(1, 30)), tris_per_blob=int(rng.choice([8, 72, 200, 512])), x_res=w, y_res=h, grid=(5, 3, 2),
                                   spacing=float(rng.choice([0.2, 0.45])))
    else:
        sc = scenes.cornell(w, h)
    o = oracle_render(oracle, sc, spp, max_bounces=mb, threads=16)
    for flags in FLAG_SETS:
        g = gpu_render(sc, spp, max_bounces=mb, flags=flags)
        exact = np.ones(g["beauty"].shape[:2], bool)
        for p in ("beauty", "denoise", "normal", "tangent", "bitangent"):
            exact &= (g[p].view(np.uint32) == o[p].view(np.uint32)).all(-1)
        exact &= (g["rng"] == o["rng"]).reshape(exact.shape) & (g["samples"] == o["samples"]).reshape(exact.shape)
        if not (exact.all() and g["counters"]["bounce_samples"] == o["counters"]["bounce_samples"]):
            bad += 1
            print("MISMATCH case", it, "kind", kind, "seed", seed, (w, h), "bounces", mb, "spp", spp, "tris", sc.tri_count, "flags", flags,
                  "pixels differing", int((~exact).sum()), flush=True)
    print("case", it, "kind", kind, "tris", sc.tri_count, (w, h), "bounces", mb, "spp", spp, "| mismatching runs so far:", bad, f"{time.time() - t0:.1f} s", flush=True)
print("DONE: mismatching runs =", bad, "of", cases * len(FLAG_SETS))
