"""Stress for timing-dependent faults of the streaming schedule: renders small and awkwardly sized scenes many times and compares
every frame with the wavefront schedule's, bit for bit (planes, RNG plane, sample counts, path count).
GPU box, repo root: python tools/stress_small.py [repeats]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from elevenrender_amd import abi, scenes
from test_gpu_parity import gpu_render

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
LM = abi.FLAG_POINT_LIGHTS | abi.FLAG_MIS
cases = [
    ("blobs 96x72 (64 px per workgroup)", scenes.blob_instances(n_instances=40, tris_per_blob=300, x_res=96, y_res=72, grid=(5, 4, 2), spacing=0.45), 7, 8, 0),
    ("textured + lights + MIS 64x48", scenes.torture(3000, 64, 48, seed=5, n_materials=8, tex_size=16, hdri_size=(128, 64), n_lights=16), 5, 16, LM),
    ("soup 640x448 (1120 px per workgroup: just above the 1024 slots)", scenes.soup(20000, 640, 448, seed=8, hdri_size=(128, 64)), 6, 8, 0),
    ("soup 200x120 (partial tiles, idle workgroups)", scenes.soup(5000, 200, 120, seed=2, hdri_size=(64, 32)), 9, 8, 0),
]
total_bad = 0
for name, sc, spp, mb, ext in cases:
    ref = gpu_render(sc, spp, max_bounces=mb, flags=abi.FLAG_WAVEFRONT | ext)
    bad = 0
    for i in range(reps):
        chunks = None if i % 3 == 0 else ([1] * spp if i % 3 == 1 else [2, spp - 2])
        g = gpu_render(sc, spp, max_bounces=mb, flags=abi.FLAG_STREAM | ext, chunks=chunks)
        d = (g["beauty"].view(np.uint32) != ref["beauty"].view(np.uint32)).any(-1)
        if d.any() or (g["rng"] != ref["rng"]).any() or (g["samples"] != ref["samples"]).any() or g["counters"]["paths"] != ref["counters"]["paths"]:
            bad += 1
            print(f"  run {i} (chunks {chunks}): {int(d.sum())} pixels differ, samples differ {(g['samples'] != ref['samples']).sum()}, "
                  f"paths {g['counters']['paths']} vs {ref['counters']['paths']}")
    print(f"{name}: {bad} of {reps} runs differ")
    total_bad += bad
sys.exit(1 if total_bad else 0)
