#!/usr/bin/env python3
"""How much would finer triangle REFERENCES buy on the C2 soup?  Upper-bound probe for a spatial-split builder: every triangle is
replaced by its four midpoint sub-triangles (same surfaces, same materials; NOT the same image bits -- hit records differ -- so this
is a measurement tool, not a product path).  Prints node visits / triangle tests per ray and the time per sample pass.
GPU box: python3 tools/presplit_potential.py [levels]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elevenrender_amd import abi, render, scenes


def subdivide(sc):
    v = sc.vertices.reshape(-1, 3, 3)
    a, b, c = v[:, 0], v[:, 1], v[:, 2]
    ab, bc, ca = (a + b) * np.float32(0.5), (b + c) * np.float32(0.5), (c + a) * np.float32(0.5)
    nv = np.stack([np.stack([a, ab, ca], 1), np.stack([ab, b, bc], 1), np.stack([ca, bc, c], 1), np.stack([ab, bc, ca], 1)], 1).reshape(-1, 3, 3)
    def rep(x):
        return np.ascontiguousarray(np.repeat(np.asarray(x), 4, axis=0))
    n = v.shape[0]
    return abi.SceneData(np.ascontiguousarray(nv.astype(np.float32)), rep(np.asarray(sc.normals).reshape(n, 3, 3)), rep(np.asarray(sc.tangents).reshape(n, 3, 3)),
                         rep(np.asarray(sc.uvs).reshape(n, 3, 2)), rep(np.asarray(sc.tangent_sign).reshape(n)), rep(np.asarray(sc.material_id).reshape(n)),
                         sc.materials, hdri=sc.hdri, camera=sc.camera, x_res=sc.x_res, y_res=sc.y_res)


def run(sc, label, passes=8):
    rm = render.RenderingManager(render.RenderParameters(max_bounces=8, flags=abi.FLAG_STREAM | abi.FLAG_COUNTERS))
    rm.start_rendering(sc)
    rm.render(2)
    c = rm.counters()
    rm.close()
    rm = render.RenderingManager(render.RenderParameters(max_bounces=8, flags=abi.FLAG_STREAM))
    rm.start_rendering(sc)
    rm.render(2)
    t0 = time.perf_counter()
    rm.render(passes)
    dt = time.perf_counter() - t0
    c2 = rm.counters()
    rm.close()
    print(f"{label}: {sc.tri_count} triangles, node visits per ray {c['node_visits'] / c['rays']:.2f}, triangle tests per ray {c['tri_tests'] / c['rays']:.2f}, "
          f"{dt / passes * 1e3:.2f} ms per pass, {c2['bounce_samples'] / (passes + 2) / (dt / passes) / 1e6:.0f} Msamples/s", flush=True)


def main():
    levels = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    sc = scenes.soup(1_000_000, 1920, 1080, seed=12345)
    run(sc, "C2 soup")
    for l in range(levels):
        sc = subdivide(sc)
        run(sc, f"subdivided x{4 ** (l + 1)}")


if __name__ == "__main__":
    main()
