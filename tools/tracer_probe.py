#!/usr/bin/env python3
"""Where an iteration of a tracer wave of the streaming schedule spends its time, part by part.

Needs the diagnostic build:  make -C elevenrender_amd/csrc BUILD=build_trp OUT=../libeleven_trp.so EXTRA=-DER_TRACER_PROBE
and runs on the GPU box:     ELEVEN_HIP_LIB=$PWD/elevenrender_amd/libeleven_trp.so python3 tools/tracer_probe.py [passes] [world] [C2|C4]
(world > 1: rank 0's share of a world-way tile split of the frame.)  In that build every tracer wave stamps s_memtime (100 MHz on
gfx950: the constant reference clock, not shader cycles) at the boundaries of its loop's parts; the event counters carry the sums."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elevenrender_amd import abi, render, scenes

PARTS = ["ring visit (publish finished rays, take new ones) + loop top", "choose the step (pop / nearest child / push, triangle pair)",
         "issue the 11 loads, wait for the triangle pieces", "triangle block", "wait for the node pieces (+ top of the tree from LDS)",
         "node block + end of the iteration", "idle polls (no ray in the wave)"]
KEYS = ["node_visits", "tri_tests", "shaded_hits", "texel_fetches", "hdri_samples", "trace_wave_steps", "trace_busy_lanes", "trace_node_lanes", "trace_tri_lanes"]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    world = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    cfg = sys.argv[3] if len(sys.argv) > 3 else "C2"
    sc = scenes.blob_instances() if cfg == "C4" else scenes.soup(1_000_000, 1920, 1080, seed=12345)
    rm = render.RenderingManager(render.RenderParameters(max_bounces=8, rank=0, world=world, flags=abi.FLAG_STREAM))
    rm.start_rendering(sc)
    rm.render(2)
    c0 = rm.counters()
    t0 = time.perf_counter()
    rm.render(n)
    wall = time.perf_counter() - t0
    c1 = rm.counters()
    rm.close()
    d = {k: c1[k] - c0[k] for k in KEYS + ["paths", "bounce_samples", "rays"]}
    iters, busy, rays = max(1, d["paths"]), d["bounce_samples"], max(1, d["rays"])
    parts = [d[k] for k in KEYS[:7]]
    pubs, takes = d[KEYS[7]], d[KEYS[8]]
    px = sc.x_res * sc.y_res // world // 256
    waves = int(os.environ.get("ER_STREAM_WAVES", "12" if px <= 1152 else "16"))
    tracers = int(os.environ.get("ER_STREAM_TRACERS", "9" if waves == 12 else ("12" if cfg == "C4" else "13")))
    # the stamps' unit, from the run itself: every tracer wave stamps from the start of its loop to its end, i.e. for the whole launch
    tick_ns = float(os.environ.get("ER_MEMTIME_NS", "0")) or wall * 1e9 * 256 * tracers / max(1, sum(parts))
    print(f"{cfg}{'' if world == 1 else f', rank 0 of {world}'} ({px} pixels per CU): {n} passes in {wall * 1e3:.1f} ms ({wall * 1e3 / n:.3f} ms per pass); "
          f"{iters} tracer-wave iterations, {busy / iters:.1f} lanes with a ray in each, {rays} rays = {busy / rays:.1f} iterations per ray")
    work = sum(parts[:6])
    print(f"  (one s_memtime tick = {tick_ns:.2f} ns, from the launch's wall time and {tracers} tracer waves on 256 CUs)")
    print(f"  an iteration lasts {work / iters * tick_ns / 1e3:.3f} us  (publishes in {pubs / iters:.3f} of the iterations, refills in {takes / iters:.3f})")
    for nm, v in zip(PARTS[:6], parts[:6]):
        print(f"  {v / iters * tick_ns / 1e3:7.3f} us  {v / work:6.3f}  {nm}")
    print(f"  a ray's traversal = {busy / rays:.1f} iterations x {work / iters * tick_ns / 1e3:.3f} us = {busy / rays * work / iters * tick_ns / 1e3:.1f} us;  idle polls: {parts[6] / max(1, work + parts[6]):.3f} of the tracer waves' time")


if __name__ == "__main__":
    main()
