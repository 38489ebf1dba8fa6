"""OBJ ingest of the C++ host mirror (elevenrender_amd/host/eleven_obj.hpp), host only.

The Cornell box of BASELINE config 1 is written out as OBJ text (z negated, because the loader flips z exactly as
the reference does, src/ObjLoader.cpp:116-117) and read back: positions, normals, uvs and material names must come
back exactly; the generated tangents must be unit, orthogonal to the normals and equal to the direction of
increasing u, which for this scene is the generator's own tangent."""
import os
import subprocess

import numpy as np

from elevenrender_amd import scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NATIVE = os.path.join(ROOT, "tests", "native")


def build():
    exe, src = os.path.join(NATIVE, "obj_dump"), os.path.join(NATIVE, "obj_dump.cpp")
    hdr = os.path.join(ROOT, "elevenrender_amd", "host", "eleven_obj.hpp")
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O1", "-std=c++17", src, "-o", exe])
    return exe


def write_obj(path, sc, names=None):
    v = sc.vertices.reshape(-1, 3, 3)
    n = sc.normals.reshape(-1, 3, 3)
    uv = sc.uvs.reshape(-1, 3, 2)
    with open(path, "w") as f:
        f.write("# cornell box\no box\n")
        for t in range(len(v)):
            for j in range(3):
                f.write("v %.9g %.9g %.9g\n" % (v[t, j, 0], v[t, j, 1], -v[t, j, 2]))
                f.write("vn %.9g %.9g %.9g\n" % (n[t, j, 0], n[t, j, 1], -n[t, j, 2]))
                f.write("vt %.9g %.9g\n" % (uv[t, j, 0], uv[t, j, 1]))
        for t in range(len(v)):
            f.write("usemtl %s\n" % (names[int(sc.material_id[t])] if names else "m%d" % int(sc.material_id[t])))
            a = 3 * t + 1
            if t % 2 == 0:
                f.write("f %d/%d/%d %d/%d/%d %d/%d/%d\n" % (a, a, a, a + 1, a + 1, a + 1, a + 2, a + 2, a + 2))
            else:       # relative indices, counted from the end of the arrays
                total = 3 * len(v)
                r = [a - total - 1, a + 1 - total - 1, a + 2 - total - 1]
                f.write("f %d/%d/%d %d/%d/%d %d/%d/%d\n" % (r[0], r[0], r[0], r[1], r[1], r[1], r[2], r[2], r[2]))


def parse(out):
    corners, signs, mats = [], [], []
    for line in out.splitlines():
        p = line.split()
        if p[0] == "c":
            corners.append([float(x) for x in p[1:]])
        elif p[0] == "t":
            signs.append(float(p[1]))
            mats.append(p[2])
    return np.array(corners, np.float32).reshape(-1, 3, 11), np.array(signs, np.float32), mats


def test_cornell_obj_round_trip(tmp_path):
    sc = scenes.cornell(64, 64)
    path = str(tmp_path / "cornell.obj")
    write_obj(path, sc)
    c, signs, mats = parse(subprocess.check_output([build(), path], text=True))
    assert c.shape[0] == sc.tri_count == 12
    assert (c[:, :, 0:3] == sc.vertices.reshape(-1, 3, 3)).all()
    assert np.allclose(c[:, :, 3:6], sc.normals.reshape(-1, 3, 3), atol=1e-6)
    assert (c[:, :, 6:8] == sc.uvs.reshape(-1, 3, 2)).all()
    assert mats == ["m%d" % int(m) for m in sc.material_id]
    tang = c[:, :, 8:11]
    assert np.allclose(np.linalg.norm(tang, axis=-1), 1.0, atol=1e-5)
    assert np.abs((tang * c[:, :, 3:6]).sum(-1)).max() < 1e-5
    assert np.allclose(tang, sc.tangents.reshape(-1, 3, 3), atol=1e-5)      # uv (0,0),(1,0),(0,1): dP/du is the first edge
    assert (np.abs(signs) == 1).all()


def test_polygons_missing_attributes_and_recomputed_normals(tmp_path):
    path = str(tmp_path / "quad.obj")
    with open(path, "w") as f:
        f.write("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nv 0 0 -1\ng a\nusemtl red\nf 1 2 3 4\ng b\nf 1//1 2 5\n")
    c, signs, mats = parse(subprocess.check_output([build(), path, "1"], text=True))
    assert c.shape[0] == 3 and mats == ["red", "red", "red"]          # the quad fans into two triangles
    assert (c[0, :, 2] == 0).all() and c[2, 2, 2] == 1.0               # z flipped on load
    n = c[:, :, 3:6]
    assert np.allclose(np.linalg.norm(n, axis=-1), 1.0, atol=1e-6)     # recomputed, face-weighted per position
    assert np.allclose(n[0, 2], [0, 0, -1]) or np.allclose(n[0, 2], [0, 0, 1])
