"""Soak of the streaming schedule on the GPU box: long renders in uneven calls, compared with the wavefront schedule bit for bit.
Usage (repo root): python tools/soak_stream.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from elevenrender_amd import abi, render, scenes
from test_gpu_parity import gpu_render
t0 = time.time()
sc = scenes.soup(200_000, 1280, 720, seed=5, hdri_size=(512, 256))
w = gpu_render(sc, 160, max_bounces=8, flags=abi.FLAG_WAVEFRONT)
s = gpu_render(sc, 160, max_bounces=8, flags=abi.FLAG_STREAM, chunks=[1, 2, 3, 50, 104])
for p in ("beauty", "normal", "tangent", "bitangent"):
    assert (w[p].view(np.uint32) == s[p].view(np.uint32)).all(), p
assert (w["rng"] == s["rng"]).all()
print("soak 1: 1280x720 x 160 spp, stream (5 calls) == wavefront bit for bit,", round(time.time() - t0, 1), "s")
t0 = time.time()
sc = scenes.torture(300_000, 1024, 576, seed=3, hdri_size=(512, 256), n_lights=256)
if sc is not None:
    fl = abi.FLAG_POINT_LIGHTS | abi.FLAG_MIS
    w = gpu_render(sc, 48, max_bounces=16, flags=abi.FLAG_WAVEFRONT | fl)
    s = gpu_render(sc, 48, max_bounces=16, flags=abi.FLAG_STREAM | fl)
    for p in ("beauty", "normal", "tangent", "bitangent"):
        assert (w[p].view(np.uint32) == s[p].view(np.uint32)).all(), p
    print("soak 2: torture scene with lights + MIS, 48 spp x 16 bounces, stream == wavefront,", round(time.time() - t0, 1), "s")
