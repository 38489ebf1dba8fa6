"""Loads tests/golden/*.npz (inputs + expected outputs; see tests/golden/make_golden.py)."""
import os

import numpy as np

from elevenrender_amd import abi

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MAT_FIELDS = [(n, t) for n, t in abi.ErMaterial._fields_]


def load(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    mats = []
    for row in z["materials"]:
        m = abi.ErMaterial()
        k = 0
        for n, t in MAT_FIELDS:
            if t is abi.ErVec3:
                setattr(m, n, abi.ErVec3(*row[k:k + 3]))
                k += 3
            else:
                setattr(m, n, int(row[k]) if "int" in t.__name__ else float(row[k]))
                k += 1
        mats.append(m)
    c = z["camera"]
    cam = abi.ErCamera()
    (cam.focal_length, cam.sensor_width, cam.sensor_height, cam.aperture, cam.focus_distance) = (float(v) for v in c[:5])
    cam.rotation = abi.ErVec3(*c[5:8])
    cam.bokeh = int(c[8])
    cam.position = abi.ErVec3(*c[9:12])
    textures = []
    i = 0
    while f"tex{i}" in z:
        w, h, ch, flt = (int(v) for v in z[f"tex{i}_meta"])
        textures.append((z[f"tex{i}"], w, h, ch, flt))
        i += 1
    hw, hh, hch, hflt = (int(v) for v in z["hdri_meta"])
    x_res, y_res, spp, max_bounces = (int(v) for v in z["res"])
    sc = abi.SceneData(z["vertices"], z["normals"], z["tangents"], z["uvs"], z["tangent_sign"], z["material_id"], mats,
                       textures=textures, hdri=(z["hdri"], hw, hh, hch, hflt), camera=cam, x_res=x_res, y_res=y_res)
    return sc, spp, max_bounces, z
