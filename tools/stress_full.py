"""The C2 frame (1M-triangle soup, 1920x1080) through the streaming schedule again and again, each run compared with the wavefront
schedule's frame bit for bit.  GPU box, repo root: python tools/stress_full.py [repeats] [spp]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from elevenrender_amd import abi, render, scenes

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 6
sc = scenes.soup(1_000_000, 1920, 1080, seed=12345)


def frame(flags, chunks):
    rm = render.RenderingManager(render.RenderParameters(max_bounces=8, flags=flags))
    rm.start_rendering(sc)
    out = []
    for n in chunks:
        rm.render(n)
    r = (rm.get_pass("beauty"), rm.read_rng(), rm.read_samples(), rm.counters()["paths"])
    rm.close()
    return r


ref = frame(abi.FLAG_WAVEFRONT, [spp])
bad = 0
t0 = time.time()
rm = render.RenderingManager(render.RenderParameters(max_bounces=8, flags=abi.FLAG_STREAM))
rm.start_rendering(sc)
for i in range(reps):
    rm.start_rendering(sc)          # restart the same scene (er_render_begin again)
    for n in ([spp] if i % 2 == 0 else [1, spp - 1]):
        rm.render(n)
    b, r, s, p = rm.get_pass("beauty"), rm.read_rng(), rm.read_samples(), rm.counters()["paths"]
    ok = (b.view(np.uint32) == ref[0].view(np.uint32)).all() and (r == ref[1]).all() and (s == ref[2]).all() and p == ref[3]
    if not ok:
        bad += 1
        print(f"  run {i}: differs ({int((b.view(np.uint32) != ref[0].view(np.uint32)).any(-1).sum())} pixels, paths {p} vs {ref[3]})")
rm.close()
print(f"C2 frame x {spp} spp: {bad} of {reps} runs differ ({time.time() - t0:.0f} s)")
sys.exit(1 if bad else 0)
