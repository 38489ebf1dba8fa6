#!/usr/bin/env python3
"""Path-length statistics of the C2 frame, per pixel and per sample, from the CPU oracle (analysis helper, not a test).

What a strong-scaling schedule needs to know about a frame: how unevenly the bounce-loop iterations are spread over the
pixels (a pixel's samples are ONE RNG stream, reference src/kernel.cpp:483-485,645, so a pixel's samples run one after
the other and a launch ends on its most expensive pixels), and how well the number of RNG draws of a pixel's next sample
is predicted by its last one (what a speculative start of sample k + 1 beside sample k rests on).

    python3 tests/analysis_path_stats.py [grid_x grid_y samples]      -> prints the table; ~1 min on 8 cores
"""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle  # noqa: E402
from elevenrender_amd import scenes  # noqa: E402


def main():
    gx = int(sys.argv[1]) if len(sys.argv) > 1 else 48
    gy = int(sys.argv[2]) if len(sys.argv) > 2 else 27
    ns = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    sc = scenes.soup(1_000_000, 1920, 1080, seed=12345)
    o = oracle.Oracle(sc, math_mode=oracle.MATH_ER, max_bounces=8, threads=8)
    xs = (np.arange(gx) + 0.5) * sc.x_res / gx
    ys = (np.arange(gy) + 0.5) * sc.y_res / gy
    idx = [int(y) * sc.x_res + int(x) for y in ys for x in xs]
    its = np.zeros((len(idx), ns), np.int32)      # bounce-loop iterations (the metric's unit) per sample
    hits = np.zeros((len(idx), ns), np.int32)     # iterations that hit a triangle (5 RNG draws each when opaque)

    def run(k):
        for s in range(ns):
            recs = o.trace_pixel(idx[k])
            its[k, s] = len(recs)
            hits[k, s] = sum(1 for r in recs if r.tri >= 0)

    with ThreadPoolExecutor(8) as ex:
        list(ex.map(run, range(len(idx))))
    o.close()
    per_px = its.sum(1)
    print(f"C2, {gx} x {gy} pixels on a grid, {ns} samples each: mean iterations per sample {its.mean():.3f}")
    print("iterations per sample, share of samples:", " ".join(f"{k}:{(its == k).mean():.3f}" for k in range(1, 10)))
    q = np.percentile(per_px / ns, [0, 10, 25, 50, 75, 90, 99, 100])
    print("a pixel's mean iterations per sample, percentiles 0/10/25/50/75/90/99/100:", " ".join(f"{v:.2f}" for v in q))
    print(f"most expensive pixel / mean pixel = {per_px.max() / per_px.mean():.3f}")
    same = hits[:, 1:] == hits[:, :-1]
    w = its[:, 1:]
    print(f"next sample draws as many numbers as the last one: {same.mean():.3f} of samples, {(same * w).sum() / w.sum():.3f} weighted by iterations")
    for lo in (0, 4, 6, 7, 7.5):
        m = per_px / ns >= lo
        if m.sum():
            print(f"  pixels with mean >= {lo}: {m.mean():.3f} of pixels, {per_px[m].sum() / per_px.sum():.3f} of the work, predicted {same[m].mean():.3f}")
    np.save("/tmp/path_its.npy", its)
    np.save("/tmp/path_hits.npy", hits)


if __name__ == "__main__":
    main()
