"""OBJ ingest of the C++ host mirror (elevenrender_amd/host/eleven_obj.hpp), host only.

The Cornell box of BASELINE config 1 is written out as OBJ text (z negated, because the loader flips z exactly as
the reference does, src/ObjLoader.cpp:116-117) and read back: positions, normals, uvs and material names must come
back exactly; the generated tangents must be unit, orthogonal to the normals and equal to the direction of
increasing u, which for this scene is the generator's own tangent."""
import os
import subprocess

import numpy as np
import pytest

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
    # every triangle has uv (0,0),(1,0),(0,1): dP/du is its first edge.  The two triangles of a wall share corner 0 with the SAME
    # position, normal and uv (one welded vertex), but the diagonal they share carries different uvs on its two sides, so they
    # are not neighbours across an edge of equal welded indices: mikktspace.c keeps them in separate groups (round 2 grouped by
    # attribute equality alone and blended the two first edges at that corner)
    ref = sc.tangents.reshape(-1, 3, 3)
    assert np.allclose(tang, ref, atol=1e-5)
    assert (signs == 1).all()


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


def test_mikktspace_style_tangents_on_a_uv_mapped_cube(tmp_path):
    """Hand-computable case: a cube whose faces carry the unit uv square.  On every face the tangent of every corner is
    the direction of increasing u (exactly: both triangles of a face agree, so the angle-weighted blend is that
    direction), it is orthogonal to the face normal, and the sign is +1 where (T, dP/dv, N) is right-handed in uv
    orientation and -1 on the faces whose uv square is mirrored."""
    path = str(tmp_path / "cube.obj")
    faces = []   # (origin, u axis, v axis): corner = o + a*U + b*V, uv = (a, b)
    for axis in range(3):
        for sgn in (0.0, 1.0):
            o = np.zeros(3); o[axis] = sgn
            U = np.zeros(3); U[(axis + 1) % 3] = 1
            V = np.zeros(3); V[(axis + 2) % 3] = 1
            faces.append((o, U, V))
    with open(path, "w") as f:
        f.write("o cube\n")
        for k, (o, U, V) in enumerate(faces):
            mirrored = k % 2 == 1
            for a, b in ((0, 0), (1, 0), (1, 1), (0, 1)):
                p = o + a * U + b * V
                f.write("v %g %g %g\n" % (p[0], p[1], -p[2]))                # the loader flips z back
                f.write("vt %g %g\n" % ((1 - a) if mirrored else a, b))
            n = np.cross(U, V)
            f.write("vn %g %g %g\n" % (n[0], n[1], -n[2]))
            q = 4 * k
            f.write("f %d/%d/%d %d/%d/%d %d/%d/%d %d/%d/%d\n" % (q + 1, q + 1, k + 1, q + 2, q + 2, k + 1, q + 3, q + 3, k + 1, q + 4, q + 4, k + 1))
    c, signs, mats = parse(subprocess.check_output([build(), path], text=True))
    assert c.shape[0] == 12
    for k, (o, U, V) in enumerate(faces):
        mirrored = k % 2 == 1
        want = -U if mirrored else U
        for t in (2 * k, 2 * k + 1):
            assert np.allclose(c[t, :, 8:11], want, atol=1e-6), (k, c[t, :, 8:11])
            assert signs[t] == (-1.0 if mirrored else 1.0)
            assert np.abs((c[t, :, 8:11] * c[t, :, 3:6]).sum(-1)).max() < 1e-6


def test_mikktspace_style_tangents_on_a_smooth_sphere(tmp_path):
    """A latitude-longitude sphere with smooth normals and u = longitude: away from the poles every vertex's tangent is
    the unit east vector d/d(longitude), shared by all (six) triangles that meet there, orthogonal to the normal."""
    path = str(tmp_path / "sphere.obj")
    nu, nv = 24, 12
    with open(path, "w") as f:
        f.write("o sphere\n")
        for j in range(nv + 1):
            th = np.pi * j / nv
            for i in range(nu + 1):                     # the seam column is duplicated so uvs stay monotone
                ph = 2 * np.pi * i / nu
                p = np.array([np.sin(th) * np.cos(ph), np.cos(th), np.sin(th) * np.sin(ph)])
                f.write("v %.9g %.9g %.9g\nvn %.9g %.9g %.9g\nvt %.9g %.9g\n" % (p[0], p[1], -p[2], p[0], p[1], -p[2], i / nu, j / nv))
        idx = lambda i, j: j * (nu + 1) + i + 1
        for j in range(1, nv - 1):                      # skip the polar caps (degenerate uv there)
            for i in range(nu):
                a, b, c_, d = idx(i, j), idx(i + 1, j), idx(i + 1, j + 1), idx(i, j + 1)
                f.write("f %d/%d/%d %d/%d/%d %d/%d/%d\n" % (a, a, a, b, b, b, c_, c_, c_))
                f.write("f %d/%d/%d %d/%d/%d %d/%d/%d\n" % (a, a, a, c_, c_, c_, d, d, d))
    c, signs, mats = parse(subprocess.check_output([build(), path], text=True))
    pos, nrm, tang = c[:, :, 0:3].reshape(-1, 3), c[:, :, 3:6].reshape(-1, 3), c[:, :, 8:11].reshape(-1, 3)
    east = np.stack([-pos[:, 2], np.zeros(len(pos)), pos[:, 0]], -1)        # d/d(longitude) of (sin th cos ph, cos th, sin th sin ph)
    east /= np.linalg.norm(east, axis=-1, keepdims=True)
    interior = (np.abs(pos[:, 1]) < np.cos(np.pi * 1.5 / nv))               # vertices whose whole fan is in the mesh
    seam = (np.abs(pos[:, 2]) < 1e-6) & (pos[:, 0] > 0)                     # longitude 0 = 2 pi: two uv values, so two half-fan groups
    assert (interior & ~seam).sum() > 1000
    assert (np.abs((tang * nrm).sum(-1)) < 1e-5).all()
    assert ((tang[interior & ~seam] * east[interior & ~seam]).sum(-1) > 0.9999).all()
    assert ((tang[interior & seam] * east[interior & seam]).sum(-1) > 0.99).all()      # half a fan: biased by half a segment
    assert len(set(signs.tolist())) == 1                                    # one orientation everywhere
    # the same vertex gets the same tangent in every triangle that uses it
    key = np.round(c[:, :, 0:8].reshape(-1, 8), 6)
    _, inv = np.unique(key, axis=0, return_inverse=True)
    for g in np.unique(inv.reshape(-1))[:200]:
        m = inv.reshape(-1) == g
        assert np.abs(tang[m] - tang[m][0]).max() < 1e-6


def test_mtl_reader_key_handling(tmp_path):
    """parse_mtl against the reference's ObjLoader::parseMtl rules (src/ObjLoader.cpp:10-50)."""
    path = str(tmp_path / "m.mtl")
    with open(path, "w") as f:
        f.write("# comment\nnewmtl red\nNs 96\nKa 1 1 1\nKd 0.8 0.1 0.1\nKs 0.25 0.5 0.75\nKe 0 0 0\nNi 1.45\nd 0.5\nillum 2\n"
                "map_Kd wood.png\nmap_Bump bump.png\n\nnewmtl lamp\nKe 5 4 3\nrefl env.hdr\nmap_Ns gloss.png\n")
    out = subprocess.check_output([build(), "--mtl", path], text=True).splitlines()
    assert out[0].split()[:2] == ["m", "red"]
    red = dict(zip(out[0].split()[2::2], out[0].split()[3::2]))
    assert [float(v) for v in red["Kd"].split(",")] == pytest.approx([0.8, 0.1, 0.1])
    assert float(red["specular"]) == pytest.approx(0.25) and float(red["eta"]) == pytest.approx(1.45) and float(red["opacity"]) == pytest.approx(0.5)
    assert red["map_Kd"] == "wood.png" and red["map_Bump"] == "bump.png"
    lamp = dict(zip(out[1].split()[2::2], out[1].split()[3::2]))
    assert [float(v) for v in lamp["Ke"].split(",")] == pytest.approx([5, 4, 3]) and lamp["refl"] == "env.hdr" and lamp["map_Ns"] == "gloss.png"
    assert float(lamp["opacity"]) == 1.0 and float(lamp["specular"]) == 0.5      # Material defaults (src/Material.h:20-47)


def _dump(tmp_path, name, text):
    path = str(tmp_path / name)
    open(path, "w").write(text)
    return parse(subprocess.check_output([build(), path], text=True))


def _tri(f, corners, n=(0, 0, 1)):
    """append one triangle (loaded-space positions, uv) to OBJ text f; returns nothing (positions are written z-negated)"""
    for (p, uv) in corners:
        f.append("v %.9g %.9g %.9g\nvt %.9g %.9g\nvn %.9g %.9g %.9g" % (p[0], p[1], -p[2], uv[0], uv[1], n[0], n[1], -n[2]))
    k = len([l for l in f if l.startswith("v ")])
    f.append("f %d/%d/%d %d/%d/%d %d/%d/%d" % (k - 2, k - 2, k - 2, k - 1, k - 1, k - 1, k, k, k))


def test_mikktspace_mirrored_uv_quads_do_not_blend_across_the_shared_edge(tmp_path):
    """Two quads in the plane z = 0 share the edge x = 0 with identical position, normal AND uv there, but the right quad's uv
    square is mirrored (u = 1 - x).  mikktspace.c: neighbours across that edge, yet of different orientation, so the flood
    (AssignRecur) stops: the left quad's corners get +x with sign +1, the right quad's -x (the direction of increasing u) with
    sign -1 -- also at the two shared vertices."""
    f = ["o quads"]
    L = [((-1, 0, 0), (0, 0)), ((0, 0, 0), (1, 0)), ((0, 1, 0), (1, 1)), ((-1, 1, 0), (0, 1))]
    R = [((0, 0, 0), (1, 0)), ((1, 0, 0), (0, 0)), ((1, 1, 0), (0, 1)), ((0, 1, 0), (1, 1))]
    for q in (L, R):
        _tri(f, [q[0], q[1], q[2]])
        _tri(f, [q[0], q[2], q[3]])
    c, signs, _ = _dump(tmp_path, "mirror.obj", "\n".join(f) + "\n")
    assert np.allclose(c[0:2, :, 8:11], [1, 0, 0], atol=1e-6) and (signs[0:2] == 1).all()
    assert np.allclose(c[2:4, :, 8:11], [-1, 0, 0], atol=1e-6) and (signs[2:4] == -1).all()


def test_mikktspace_groups_need_edge_connectivity_not_just_equal_attributes(tmp_path):
    """A bow tie: two triangles that share ONE vertex (same position, normal, uv, same orientation) and no edge.  mikktspace.c
    grows groups across shared edges only, so each triangle keeps its own dP/du at that vertex: (1,0,0) and (0,-1,0) -- an
    attribute-equality grouping (round 2) would have blended them to the diagonal."""
    f = ["o bowtie"]
    _tri(f, [((0, 0, 0), (0, 0)), ((1, 0, 0), (1, 0)), ((0, 1, 0), (0, 1))])
    _tri(f, [((0, 0, 0), (0, 0)), ((0, -1, 0), (1, 0)), ((1, -1, 0), (0, 1))])
    c, signs, _ = _dump(tmp_path, "bowtie.obj", "\n".join(f) + "\n")
    assert np.allclose(c[0, :, 8:11], [1, 0, 0], atol=1e-6)
    assert np.allclose(c[1, :, 8:11], [0, -1, 0], atol=1e-6)
    assert (signs == 1).all()


def test_mikktspace_degenerate_and_zero_uv_area_triangles(tmp_path):
    """Hand-derived from mikktspace.c's rules:
      G  good triangle, dP/du = (0,1,0): every corner (0,1,0), sign +1;
      A  zero uv area (GROUP_WITH_ANY), neighbour of G across G's edge 1->2 with opposite winding: its two corners on that edge
         are pulled into G's groups and get G's tangent (A itself contributes nothing); its third corner is never assigned and
         keeps the initial space (1,0,0) / not orientation preserving -- and the triangle's ONE sign is what the last corner
         delivered: -1;
      D  two equal positions (degenerate): corners 0 and 1 coincide with G's vertex 1 in position, normal and uv and borrow its
         tangent (0,1,0); corner 2 finds no good corner and keeps (1,0,0); sign of the last corner: -1;
      Z  an isolated zero-uv-area triangle: (1,0,0) three times, sign -1."""
    f = ["o mixed"]
    _tri(f, [((0, 0, 0), (0, 0)), ((0, 1, 0), (1, 0)), ((-1, 0, 0), (0, 1))])                      # G
    _tri(f, [((-1, 0, 0), (0, 1)), ((0, 1, 0), (1, 0)), ((-1, 1, 0), (0.5, 0.5))])                 # A
    _tri(f, [((0, 1, 0), (1, 0)), ((0, 1, 0), (1, 0)), ((5, 5, 0), (3, 3))])                       # D
    _tri(f, [((10, 0, 0), (0, 0)), ((11, 0, 0), (0.25, 0.25)), ((10, 1, 0), (0.5, 0.5))])          # Z
    c, signs, _ = _dump(tmp_path, "mixed.obj", "\n".join(f) + "\n")
    T = c[:, :, 8:11]
    assert np.allclose(T[0], [[0, 1, 0]] * 3, atol=1e-6) and signs[0] == 1
    assert np.allclose(T[1], [[0, 1, 0], [0, 1, 0], [1, 0, 0]], atol=1e-6) and signs[1] == -1
    assert np.allclose(T[2], [[0, 1, 0], [0, 1, 0], [1, 0, 0]], atol=1e-6) and signs[2] == -1
    assert np.allclose(T[3], [[1, 0, 0]] * 3, atol=1e-6) and signs[3] == -1


def test_mikktspace_uv_seamed_cylinder(tmp_path):
    """A cylinder around the y axis, smooth radial normals, u = longitude / 2 pi with the seam column duplicated (u = 0 and
    u = 1 at the same position: different uv, so two vertices and no edge across the seam).  Away from the seam every vertex's
    six triangles form one group and their angle-weighted dP/du is the unit east vector; at the seam each side is a half fan
    whose tangent leans by at most half a segment."""
    nu, nv = 20, 4
    f = ["o cylinder"]
    for j in range(nv + 1):
        for i in range(nu + 1):
            ph = 2 * np.pi * i / nu
            p = (np.cos(ph), j / nv, np.sin(ph))
            f.append("v %.9g %.9g %.9g\nvn %.9g %.9g %.9g\nvt %.9g %.9g" % (p[0], p[1], -p[2], p[0], 0.0, -p[2], i / nu, j / nv))
    idx = lambda i, j: j * (nu + 1) + i + 1
    for j in range(nv):
        for i in range(nu):
            a, b, c_, d = idx(i, j), idx(i + 1, j), idx(i + 1, j + 1), idx(i, j + 1)
            f.append("f %d/%d/%d %d/%d/%d %d/%d/%d" % (a, a, a, b, b, b, c_, c_, c_))
            f.append("f %d/%d/%d %d/%d/%d %d/%d/%d" % (a, a, a, c_, c_, c_, d, d, d))
    c, signs, _ = _dump(tmp_path, "cyl.obj", "\n".join(f) + "\n")
    pos, nrm, tang, uv = c[:, :, 0:3].reshape(-1, 3), c[:, :, 3:6].reshape(-1, 3), c[:, :, 8:11].reshape(-1, 3), c[:, :, 6:8].reshape(-1, 2)
    east = np.stack([-pos[:, 2], np.zeros(len(pos)), pos[:, 0]], -1)
    inner = (uv[:, 1] > 0.01) & (uv[:, 1] < 0.99)
    seam = (uv[:, 0] < 1e-6) | (uv[:, 0] > 1 - 1e-6)
    assert (np.abs((tang * nrm).sum(-1)) < 1e-5).all() and np.allclose(np.linalg.norm(tang, axis=-1), 1, atol=1e-5)
    assert ((tang[inner & ~seam] * east[inner & ~seam]).sum(-1) > 0.99999).all()
    assert ((tang[seam] * east[seam]).sum(-1) > np.cos(np.pi / nu) - 1e-4).all()
    assert len(set(signs.tolist())) == 1
