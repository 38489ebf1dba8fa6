#!/usr/bin/env python3
"""Where a bounce of a path spends its time in the streaming schedule, stage by stage (closest-hit rays only).

Needs the diagnostic build:  make -C elevenrender_amd/csrc BUILD=build_sp OUT=../libeleven_sp.so EXTRA=-DER_STAGE_PROBE
and runs on the GPU box:     ELEVEN_HIP_LIB=$PWD/elevenrender_amd/libeleven_sp.so python3 tools/stage_probe.py [passes] [world] [C2|C4|C5|C5nl]
In that build every closest-hit ray is stamped (10-ns ticks) when a shader wave queues it (A), when a tracer lane takes it (B), when
its traversal ends (C), when the tracer publishes it (D), when a shader wave starts the slot's next step (E) and when that step has
queued the next ray (F); the event counters carry the sums."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))      # (the repo root: elevenrender_amd, bench)
from elevenrender_amd import abi, render, scenes


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    world = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    cfg = sys.argv[3] if len(sys.argv) > 3 else "C2"      # C2 | C4 | C5 (with lights + MIS) | C5nl: bench.py's scenes
    import argparse
    import bench
    sc, ext, _ = bench.make_scene(argparse.Namespace(config=cfg[:2], width=0, height=0, tris=1_000_000, blob_tris=1000, no_lights=cfg == "C5nl"), scenes, abi)
    rm = render.RenderingManager(render.RenderParameters(max_bounces=bench.CONFIGS[cfg[:2]][1], rank=0, world=world, flags=abi.FLAG_STREAM | ext))
    rm.start_rendering(sc)
    rm.render(2)
    c0 = rm.counters()
    t0 = time.perf_counter()
    rm.render(n)
    wall = time.perf_counter() - t0
    c1 = rm.counters()
    rm.close()
    keys = ["node_visits", "tri_tests", "shaded_hits", "texel_fetches", "hdri_samples", "trace_wave_steps", "trace_busy_lanes", "trace_node_lanes", "trace_tri_lanes", "paths"]
    d = [c1[k] - c0[k] for k in keys]
    rays = max(1, d[5])
    names = ["A->B  queued by a shader wave -> taken by a tracer lane (ray ring)", "B->C  traversal", "C->D  finished -> published (the wave's next ring visit)",
             "D->E  published -> the slot's next step starts (its other rays, the shade / finish ring, a free shader wave)", "E->F  the step, until the next ray is queued"]
    px = sc.x_res * sc.y_res // world // 256
    print(f"{cfg}{'' if world == 1 else f', rank 0 of {world}'} ({px} pixels per CU): {n} passes in {wall * 1e3:.1f} ms ({wall * 1e3 / n:.3f} ms per pass); {rays} closest-hit rays stamped")
    tot = 0.0
    for nm, v in zip(names, d[:5]):
        us = v / rays * 0.01
        tot += us
        print(f"  {us:8.2f} us  {nm}")
    if d[7]:
        print(f"  {d[6] / d[7] * 0.01:8.2f} us  G->E' a path that ended in a shading step: handed to the finish ring -> its finishing step starts ({d[7]} such paths, {d[7] / max(1, n) / (sc.x_res * sc.y_res // world):.2f} per pixel and pass)")
    if d[9]:
        print(f"  {d[8] / d[9] * 0.01:8.2f} us  a whole sample in its slot: first camera ray queued -> its finishing step reaches the accumulate ({d[9]} samples)")
    print(f"  {tot:8.2f} us  per bounce in all; x {rays / max(1, n) / (sc.x_res * sc.y_res // world):.2f} closest rays per pixel and pass = {tot * rays / max(1, n) / (sc.x_res * sc.y_res // world):.0f} us per sample of a pixel")


if __name__ == "__main__":
    main()
