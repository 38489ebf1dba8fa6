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
# soak 3 / 4: BASELINE's configurations in full -- C2 at its 256 spp, C4 at its 1 024 spp -- streaming schedule in uneven calls against the
# wavefront schedule in one, bit for bit (planes, RNG state, sample counts), with the wall time of the streaming render
for name, sc, spp, chunks, mb in (("C2 1920x1080 x 256 spp", scenes.soup(1_000_000, 1920, 1080, seed=12345), 256, [1, 7, 56, 192], 8),
                                 ("C4 3840x2160 x 1024 spp", scenes.blob_instances(), 1024, [3, 61, 960], 8)):
    if name.startswith("C4") and os.environ.get("ER_SOAK_C4", "1") == "0":
        continue
    t0 = time.time()
    w = gpu_render(sc, spp, max_bounces=mb, flags=abi.FLAG_WAVEFRONT)
    t1 = time.time()
    s = gpu_render(sc, spp, max_bounces=mb, chunks=chunks)          # flags = 0: the library's own choice (the streaming schedule)
    t2 = time.time()
    for p in ("beauty", "normal", "tangent", "bitangent"):
        assert (w[p].view(np.uint32) == s[p].view(np.uint32)).all(), p
    assert (w["rng"] == s["rng"]).all() and (w["samples"] == s["samples"]).all() and (s["samples"] == spp + 1).mean() > 0.999          # (a sample whose light is NaN is not accumulated: the only exceptions)
    print(f"soak: {name}: default schedule in {len(chunks)} calls == wavefront bit for bit; wall time incl. scene set-up and read-back {t2 - t1:.1f} s (wavefront {t1 - t0:.1f} s), "
          f"{s['counters']['bounce_samples'] / 1e6:.0f} M samples", flush=True)
    del w, s
# soak 5 (round 6): one GPU's share of the C2 frame at every size the kernel has a form for -- 1/2 and 1/3 (form 1: pixels that are behind keep their
# slots), 1/4 and 1/6 (form 2 at 16 waves: that and speculative samples), 1/8 and 1/16 (form 2 at 12 waves) -- streaming schedule in uneven calls
# against the wavefront schedule in one: planes, RNG states, sample counts and the event counters of exactly that share
sc = scenes.soup(1_000_000, 1920, 1080, seed=12345)
for world in (2, 3, 4, 6, 8, 16):
    t0 = time.time()
    w = gpu_render(sc, 40, max_bounces=8, flags=abi.FLAG_WAVEFRONT, rank=world - 1, world=world)
    rm = render.RenderingManager(render.RenderParameters(max_bounces=8, flags=abi.FLAG_STREAM, rank=world - 1, world=world))
    rm.start_rendering(sc)
    for n in (1, 4, 15, 20):
        rm.render(n)
    si = rm.stream_info()
    s = {p: rm.get_pass(p) for p in ("beauty", "normal", "tangent", "bitangent")}
    s["rng"], s["samples"], s["counters"] = rm.read_rng(), rm.read_samples(), rm.counters()
    rm.close()
    for p in ("beauty", "normal", "tangent", "bitangent"):
        assert (w[p].view(np.uint32) == s[p].view(np.uint32)).all(), (world, p)
    assert (w["rng"] == s["rng"]).all() and (w["samples"] == s["samples"]).all()
    for k in ("paths", "bounce_samples", "rays", "shaded_hits", "hdri_samples"):
        assert w["counters"][k] == s["counters"][k], (world, k)
    print(f"soak: rank {world - 1} of {world} of the C2 frame x 40 spp in 4 calls: form {si['form']}, {si['waves']} waves ({si['tracers']} tracers), {si['pixels_per_cu']} pixels per CU, "
          f"speculative samples {si['spec_started']} started / {si['spec_right']} right: == wavefront bit for bit, counters equal ({s['counters']['bounce_samples'] / 1e6:.0f} M samples), {time.time() - t0:.1f} s", flush=True)
