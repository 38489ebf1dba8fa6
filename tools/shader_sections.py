#!/usr/bin/env python3
"""Where a shader wave of the streaming schedule spends its time, by section of the shading step.

Needs the diagnostic build:  make -C elevenrender_amd/csrc BUILD=build_tp OUT=../libeleven_tp.so EXTRA=-DER_TIME_PROBE
and runs on the GPU box:     ELEVEN_HIP_LIB=$PWD/elevenrender_amd/libeleven_tp.so python3 tools/shader_sections.py [C2|C4|C5] [passes] [world]
(world > 1: rank 0's share of a world-way tile split of the frame -- what one GPU of an N-GPU node renders; the wave counts printed
are then those the library chose for that share: 9 + 3 of 12 waves for shares of <= 1 152 pixels per CU)
In that build the shader waves stamp s_memtime at section boundaries (ER_TPS / ER_TP in er_stream.hip, er_bounce.inc) and the
event counters carry the summed cycles / 16 per section instead of events."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elevenrender_amd import abi, render, scenes

NAMES = ["1 slot record, shadow verdicts, resolve closest hit (and the whole miss path)", "2 full hit record, material, textures (generate_hit_data)",
         "3 opacity draw + HDRI CDF search", "4 DisneySample, HDRI direction / texel / pdf, DisneyEval, shadow-query set-up, lights",
         "5 DisneyPdf + DisneyEval of the sampled direction, new ray", "6 bounce bookkeeping, slot stores, finished paths: accumulate into the planes",
         "7 pixel ring exchange + first camera ray of the next sample", "8 publish rays, retire", "9 between steps: loop top, waiting for a batch, take"]
KEYS = ["node_visits", "tri_tests", "shaded_hits", "texel_fetches", "hdri_samples", "trace_wave_steps", "trace_busy_lanes", "trace_node_lanes", "trace_tri_lanes"]


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    world = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    if cfg == "C5":
        sc, mb, fl = scenes.torture(1_000_000, 1920, 1080, seed=12345), 16, abi.FLAG_POINT_LIGHTS | abi.FLAG_MIS
    elif cfg == "C4":
        sc, mb, fl = scenes.blob_instances(), 8, 0
    else:
        sc, mb, fl = scenes.soup(1_000_000, 1920, 1080, seed=12345), 8, 0
    rm = render.RenderingManager(render.RenderParameters(max_bounces=mb, rank=0, world=world, flags=abi.FLAG_STREAM | abi.FLAG_COUNTERS | fl))
    rm.start_rendering(sc)
    t0 = time.perf_counter()
    rm.render(n)
    wall = time.perf_counter() - t0
    c = rm.counters()
    rm_info = {"cus": render.list_devices()[0]["max_compute_units"]}
    rm.close()
    tot = sum(c[k] for k in KEYS)
    px_per_cu = sc.x_res * sc.y_res // world // max(1, rm_info["cus"])
    waves = int(os.environ.get("ER_STREAM_WAVES", "12" if px_per_cu <= 1152 else "16"))
    tracers = int(os.environ.get("ER_STREAM_TRACERS", "9" if waves == 12 else ("12" if cfg in ("C5", "C4") else "13")))
    info = rm_info
    cus = info["cus"]
    steps, slots, tr_cyc = c["paths"], c["bounce_samples"], c["rays"] * 16
    print(f"{cfg}{'' if world == 1 else f' (rank 0 of {world}: {px_per_cu} pixels per CU)'}: {n} passes in {wall * 1e3:.1f} ms on {cus} CUs, {tracers} tracer + {waves - tracers} shader waves each")
    print(f"  a shading step: {slots / max(steps, 1):.1f} slots, {(tot - c[KEYS[-1]]) * 16 / max(steps, 1):.0f} cycles; a shader wave works {tot * 16 / (cus * (waves - tracers)) / 1e6:.2f} M cycles, "
          f"a tracer wave {tr_cyc / (cus * tracers) / 1e6:.2f} M cycles in its iterations (the call: ~{wall * 2.1e3:.1f} M cycles at 2.1 GHz)")
    print("  shader-wave cycles by section (share of all stamped cycles):")
    for name, k in zip(NAMES, KEYS):
        print(f"  {c[k] / tot:6.3f}  {name}")


if __name__ == "__main__":
    main()
