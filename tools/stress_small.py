"""Stress for timing-dependent faults on small frames: renders one small scene many times per schedule and compares every
frame with the wavefront schedule's, bit for bit.  GPU box, repo root: python tools/stress_small.py [repeats]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from elevenrender_amd import abi, scenes
from test_gpu_parity import gpu_render

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
sc = scenes.blob_instances(n_instances=40, tris_per_blob=300, x_res=96, y_res=72, grid=(5, 4, 2), spacing=0.45)
ref = gpu_render(sc, 7, max_bounces=8, flags=abi.FLAG_WAVEFRONT)
for name, flag, chunks in (("stream", abi.FLAG_STREAM, None), ("stream chunks", abi.FLAG_STREAM, [2, 5]), ("fused", abi.FLAG_FUSED, None),
                           ("wavefront", abi.FLAG_WAVEFRONT, None)):
    bad = 0
    for i in range(reps):
        g = gpu_render(sc, 7, max_bounces=8, flags=flag, chunks=chunks)
        d = (g["beauty"].view(np.uint32) != ref["beauty"].view(np.uint32)).any(-1)
        if d.any() or (g["rng"] != ref["rng"]).any():
            bad += 1
            ys, xs = np.nonzero(d)
            print(f"  {name} run {i}: {int(d.sum())} pixels differ, first at ({xs[0] if len(xs) else -1}, {ys[0] if len(ys) else -1}); "
                  f"samples differ {(g['samples'] != ref['samples']).sum()}, rng differ {(g['rng'] != ref['rng']).sum()}, "
                  f"bounce_samples {g['counters']['bounce_samples']} vs {ref['counters']['bounce_samples']}, paths {g['counters']['paths']} vs {ref['counters']['paths']}")
    print(f"{name}: {bad} of {reps} runs differ")
