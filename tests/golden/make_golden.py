#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz from the CPU oracle (er_math mode).

These are REGRESSION vectors of this repository's oracle, not reference outputs: the
reference cannot be built or run in this image (DESIGN.md "Oracle"), and it ships no
vectors of its own.  Each file stores the complete inputs next to the expected outputs,
so the tests never depend on regenerating inputs bit for bit.
Usage: python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import oracle  # noqa: E402
from elevenrender_amd import abi, scenes  # noqa: E402

MAT_FIELDS = [n for n, _ in abi.ErMaterial._fields_]


def mat_to_row(m):
    row = []
    for n in MAT_FIELDS:
        v = getattr(m, n)
        row += [v.x, v.y, v.z] if isinstance(v, abi.ErVec3) else [float(v)]
    return row


def cam_to_row(c):
    return [c.focal_length, c.sensor_width, c.sensor_height, c.aperture, c.focus_distance,
            c.rotation.x, c.rotation.y, c.rotation.z, float(c.bokeh), c.position.x, c.position.y, c.position.z]


def dump(name, sc, spp, max_bounces, trace_pixels):
    o = oracle.Oracle(sc, math_mode=oracle.MATH_ER, max_bounces=max_bounces)
    o.render(spp)
    out = {
        "vertices": sc.vertices, "normals": sc.normals, "tangents": sc.tangents, "uvs": sc.uvs,
        "tangent_sign": sc.tangent_sign, "material_id": sc.material_id,
        "materials": np.array([mat_to_row(m) for m in sc.materials], np.float64),
        "camera": np.array(cam_to_row(sc.camera), np.float64),
        "hdri": sc.hdri[0], "hdri_meta": np.array(sc.hdri[1:], np.int64),
        "res": np.array([sc.x_res, sc.y_res, spp, max_bounces], np.int64),
        "samples": o.read_samples(), "rng": o.read_rng(),
    }
    for i, (d, w, h, ch, flt) in enumerate(sc.textures):
        out[f"tex{i}"] = d
        out[f"tex{i}_meta"] = np.array([w, h, ch, flt], np.int64)
    for pname, p in abi.PASS_NAMES.items():
        out[f"pass_{pname}"] = o.read_pass(p)
    c = o.counters()
    out["counters"] = np.array([c["paths"], c["bounce_samples"], c["rays"], c["shaded_hits"], c["hdri_samples"]], np.int64)
    # per-bounce traces of the NEXT sample of a few pixels
    recs = []
    for px in trace_pixels:
        for r in o.trace_pixel(px):
            recs.append([px, r.bounce, r.tri, r.shadow_tri, r.opaque] + list(r.position) + list(r.wi) + list(r.light) + list(r.reduction))
    out["trace"] = np.array(recs, np.float64)
    o.close()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "paths", c["paths"], "bounce_samples", c["bounce_samples"], "trace recs", len(recs))


def main():
    L = oracle.lib()
    import ctypes as C
    n = 16
    states = np.zeros((16, n), np.uint32)
    vals = np.zeros((16, n), np.float32)
    for px in range(16):
        L.oracle_rng_stream(px, n, states[px].ctypes.data_as(C.POINTER(C.c_uint32)), vals[px].ctypes.data_as(C.POINTER(C.c_float)))
    np.savez_compressed(os.path.join(HERE, "rng_streams.npz"), states=states, values=vals)

    dump("cornell_32x32_4spp", scenes.cornell(32, 32), 4, 5, trace_pixels=[0, 135, 500, 528, 1023])
    sc = scenes.torture(300, 32, 24, seed=21, n_materials=4, tex_size=8, hdri_size=(16, 8), smooth=True)
    sc.materials[1].opacity = 0.5          # exercise the transparent branch
    sc.materials[2].albedo_shader_id = 1   # asl_shade placeholder
    sc.materials[3].normal_tex = 0         # tangent-space normal map path
    sc.camera.bokeh = 1
    sc.camera.focus_distance = 3.0
    sc.camera.rotation = abi.ErVec3(2.0, -3.0, 1.0)
    sc._desc = None
    dump("torture_300tri_32x24_4spp", sc, 4, 8, trace_pixels=[5, 100, 400, 767])


if __name__ == "__main__":
    main()
