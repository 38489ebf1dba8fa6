"""Seeded synthetic scenes for the BASELINE.json configurations.

All randomness comes from `Rand` below (a counter-based splitmix64 stream), never from
numpy/`random` generators whose sequences are version dependent.  Scenes enter at the
triangle-array level (explicit normals/tangents), as SURVEY.md section 8c prescribes.
"""
import numpy as np

from . import abi

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


class Rand:
    """splitmix64 evaluated at counters seed*2^32+stream.. : u01(n) -> float32 in [0,1)."""

    def __init__(self, seed, stream=0):
        self.base = np.uint64((int(seed) * 0x9E3779B97F4A7C15 + int(stream) * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF)
        self.ctr = np.uint64(0)

    def u64(self, n):
        with np.errstate(over="ignore"):
            i = np.arange(n, dtype=np.uint64) + self.ctr
            self.ctr = self.ctr + np.uint64(n)
            z = (self.base + (i + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)) & _M64
            z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
            z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
            return z ^ (z >> np.uint64(31))

    def u01(self, *shape):
        n = int(np.prod(shape)) if shape else 1
        v = (self.u64(n) >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / (1 << 24))
        return v.reshape(shape) if shape else v[0]

    def uniform(self, lo, hi, *shape):
        return (np.float32(lo) + (np.float32(hi) - np.float32(lo)) * self.u01(*shape)).astype(np.float32)


def _normalize(v):
    l = np.sqrt((v * v).sum(-1, keepdims=True, dtype=np.float32)).astype(np.float32)
    l = np.where(l == 0, np.float32(1), l)
    return (v / l).astype(np.float32)


def face_frame(vertices):
    """face normal on all three corners, tangent = normalised (v1 - v0), sign +1."""
    v = np.asarray(vertices, np.float32)
    e1, e2 = v[:, 1] - v[:, 0], v[:, 2] - v[:, 0]
    n = _normalize(np.cross(e1, e2).astype(np.float32))
    t = _normalize(e1)
    normals = np.repeat(n[:, None, :], 3, axis=1)
    tangents = np.repeat(t[:, None, :], 3, axis=1)
    return normals.astype(np.float32), tangents.astype(np.float32)


def sky_hdri(width=2048, height=1024):
    """procedural sky of SURVEY 8d: L(u,v) = 0.2 + 4 exp(-((u-.25)^2+(v-.35)^2)/0.002), tint (1,.9,.8)."""
    u = ((np.arange(width, dtype=np.float32) + np.float32(0.5)) / np.float32(width))[None, :]
    v = ((np.arange(height, dtype=np.float32) + np.float32(0.5)) / np.float32(height))[:, None]
    L = np.float32(0.2) + np.float32(4.0) * np.exp(-((u - np.float32(0.25)) ** 2 + (v - np.float32(0.35)) ** 2) / np.float32(0.002))
    tint = np.array([1.0, 0.9, 0.8], np.float32)
    data = (L[:, :, None].astype(np.float32) * tint[None, None, :]).astype(np.float32)
    return (data, width, height, 3, 0)


def cornell(x_res=256, y_res=256):
    """C1: Cornell-style box, 12 triangles (5 walls + ceiling light), box [-1,1]^2 x [2,4],
    camera at (0,0,-1.5), grey / red / green / emissive(5,5,5) materials, default 1x1 HDRI."""
    def quad(p0, p1, p2, p3):   # two triangles (p0,p1,p2),(p0,p2,p3)
        return [[p0, p1, p2], [p0, p2, p3]]
    x0, x1, y0, y1, z0, z1 = -1.0, 1.0, -1.0, 1.0, 2.0, 4.0
    tris, mats = [], []
    # winding chosen so cross(e1,e2) faces the interior
    tris += quad([x0, y0, z0], [x0, y0, z1], [x1, y0, z1], [x1, y0, z0]); mats += [0, 0]   # floor, normal +y
    tris += quad([x0, y1, z0], [x1, y1, z0], [x1, y1, z1], [x0, y1, z1]); mats += [0, 0]   # ceiling, normal -y
    tris += quad([x0, y0, z1], [x0, y1, z1], [x1, y1, z1], [x1, y0, z1]); mats += [0, 0]   # back wall, normal -z
    tris += quad([x0, y0, z0], [x0, y1, z0], [x0, y1, z1], [x0, y0, z1]); mats += [1, 1]   # left wall (red), normal +x
    tris += quad([x1, y0, z0], [x1, y0, z1], [x1, y1, z1], [x1, y1, z0]); mats += [2, 2]   # right wall (green), normal -x
    l, yl = 0.4, 0.995
    tris += quad([-l, yl, 3 - l], [l, yl, 3 - l], [l, yl, 3 + l], [-l, yl, 3 + l]); mats += [3, 3]   # light, normal -y
    v = np.array(tris, np.float32)
    normals, tangents = face_frame(v)
    uvs = np.tile(np.array([[0, 0], [1, 0], [0, 1]], np.float32), (len(v), 1, 1))
    materials = [abi.default_material(),
                 abi.default_material(albedo=(0.8, 0.1, 0.1)),
                 abi.default_material(albedo=(0.1, 0.8, 0.1)),
                 abi.default_material(emission=(5.0, 5.0, 5.0))]
    cam = abi.default_camera()
    cam.position = abi.ErVec3(0.0, 0.0, -1.5)
    return abi.SceneData(v, normals, tangents, uvs, np.ones(len(v), np.float32), np.array(mats, np.int32),
                         materials, camera=cam, x_res=x_res, y_res=y_res)


def soup_geometry(n_tris, seed=12345):
    """C2 distribution: centroid c ~ U([-1,1]^2 x [2,4]); v0 = c, v1,v2 = c + U([-1,1]^3) * e, e = 2/cbrt(N)."""
    r = Rand(seed, 1)
    c = r.uniform(-1, 1, n_tris, 3)
    c[:, 2] = c[:, 2] + np.float32(3.0)
    e = np.float32(2.0 / np.cbrt(float(max(n_tris, 1))))
    d1 = r.uniform(-1, 1, n_tris, 3) * e
    d2 = r.uniform(-1, 1, n_tris, 3) * e
    v = np.stack([c, c + d1, c + d2], axis=1).astype(np.float32)
    return v


def soup(n_tris=1_000_000, x_res=1920, y_res=1080, seed=12345, hdri_size=(2048, 1024)):
    """C2/C3: random-triangle soup + procedural sky HDRI, camera (0.01,0.02,-0.5), one default material."""
    v = soup_geometry(n_tris, seed)
    normals, tangents = face_frame(v)
    uvs = np.tile(np.array([[0, 0], [1, 0], [0, 1]], np.float32), (n_tris, 1, 1))
    cam = abi.default_camera()
    cam.position = abi.ErVec3(0.01, 0.02, -0.5)
    return abi.SceneData(v, normals, tangents, uvs, np.ones(n_tris, np.float32), np.zeros(n_tris, np.int32),
                         [abi.default_material()], hdri=sky_hdri(*hdri_size), camera=cam, x_res=x_res, y_res=y_res)


def value_noise_texture(size, seed, channels=3):
    """LINEAR value-noise texture in [0,1] (C5): 16x16 lattice, bilinear upsampled."""
    r = Rand(seed, 7)
    g = 16
    lat = r.u01(g + 1, g + 1, channels)
    lat[g, :, :] = lat[0, :, :]
    lat[:, g, :] = lat[:, 0, :]
    t = (np.arange(size, dtype=np.float32) + np.float32(0.5)) * np.float32(g / size)
    i0 = np.floor(t).astype(np.int64)
    f = (t - i0).astype(np.float32)
    i0 = np.clip(i0, 0, g - 1)
    a = lat[i0][:, i0] * ((1 - f)[:, None, None] * (1 - f)[None, :, None])
    a += lat[i0 + 1][:, i0] * (f[:, None, None] * (1 - f)[None, :, None])
    a += lat[i0][:, i0 + 1] * ((1 - f)[:, None, None] * f[None, :, None])
    a += lat[i0 + 1][:, i0 + 1] * (f[:, None, None] * f[None, :, None])
    return (np.clip(a, 0, 1).astype(np.float32), size, size, channels, 0)


def point_lights(n=256, seed=12345, lo=(-1.0, -1.0, 2.0), hi=(1.0, 1.0, 4.0)):
    """`n` point lights at U([-1,1]^2 x [2,4]) with radiance U(0.5,2)^3 (reference PointLight,
    src/PointLight.h:4-16).  They only take effect with ER_FLAG_POINT_LIGHTS (a build-defined extension, a15)."""
    r = Rand(seed, 7)
    u = r.u01(n, 3)
    pos = (np.asarray(lo, np.float32) + (np.asarray(hi, np.float32) - np.asarray(lo, np.float32)) * u).astype(np.float32)
    rad = r.uniform(0.5, 2.0, n, 3)
    return [abi.ErPointLight(abi.ErVec3(*map(float, pos[i])), abi.ErVec3(*map(float, rad[i]))) for i in range(n)]


def torture(n_tris=1_000_000, x_res=1920, y_res=1080, seed=12345, n_materials=64, tex_size=256,
            hdri_size=(2048, 1024), smooth=False, n_lights=256):
    """C5: C2 geometry, `n_materials` textured materials (albedo + roughness + metallic noise textures),
    clearcoat/anisotropic/sheen varied per material, `n_lights` point lights (evaluated only with
    ER_FLAG_POINT_LIGHTS: the extension path of SURVEY 8 a15; without the flag they are ignored like the
    reference ignores them)."""
    sc = soup(n_tris, x_res, y_res, seed, hdri_size)
    sc.material_id = (np.arange(n_tris, dtype=np.int64) % n_materials).astype(np.int32)
    textures, materials = [], []
    for m in range(n_materials):
        base = len(textures)
        textures.append(value_noise_texture(tex_size, 1000 + 3 * m))
        textures.append(value_noise_texture(tex_size, 1001 + 3 * m))
        textures.append(value_noise_texture(tex_size, 1002 + 3 * m))
        materials.append(abi.default_material(albedo_tex=base, roughness_tex=base + 1, metallic_tex=base + 2,
                                              clearcoat=(m % 4) / 3.0, anisotropic=(m % 5) / 5.0, sheen=(m % 3) / 2.0))
    sc.textures = [(abi._f32(d), w, h, ch, flt) for (d, w, h, ch, flt) in textures]
    sc.materials = materials
    sc.point_lights = point_lights(n_lights, seed)
    if smooth:
        r = Rand(seed, 3)
        jitter = r.uniform(-0.35, 0.35, n_tris, 3, 3)
        sc.normals = _normalize(sc.normals + jitter)
    sc._desc = None
    return sc


def blob_instances(n_instances=10000, tris_per_blob=1000, x_res=3840, y_res=2160, seed=12345, grid=(25, 20, 20), spacing=0.1):
    """C4: instances of a 1000-triangle smooth blob (subdivided octahedron with radial noise, SMOOTH vertex
    normals) flattened into world-space triangles on a jittered grid (the reference has no instancing:
    MeshObject is a triangle range, src/MeshObject.hpp:14-22)."""
    # unit blob: octahedron with every face cut into f x f triangles (8 f^2 faces, the largest f with 8 f^2 <= tris_per_blob:
    # 968 for 1000), vertices pushed onto the unit sphere and shared between faces so the normals come out smooth; the
    # remaining (tris_per_blob - 8 f^2) / 2 pairs of triangles come from splitting that many edges at their midpoints (a split
    # edge turns its two triangles into four: +2), edges spread evenly over the face list and no triangle split twice --
    # 16 splits make the 1 000 triangles of BASELINE config 4 exactly, the mesh stays closed and welded
    f = max(1, int(np.floor(np.sqrt(max(tris_per_blob, 8) / 8.0))))
    octa = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], np.float64)
    octa_faces = [[0, 2, 4], [2, 1, 4], [1, 3, 4], [3, 0, 4], [2, 0, 5], [1, 2, 5], [3, 1, 5], [0, 3, 5]]
    pts, tri = [], []
    for a, b, c in octa_faces:
        base = len(pts)
        index = {}
        for i in range(f + 1):
            for j in range(f + 1 - i):
                p = (octa[a] * (f - i - j) + octa[b] * i + octa[c] * j) / f
                index[(i, j)] = base + len(index)
                pts.append(p / np.linalg.norm(p))
        for i in range(f):
            for j in range(f - i):
                tri.append([index[(i, j)], index[(i + 1, j)], index[(i, j + 1)]])
                if i + j < f - 1:
                    tri.append([index[(i + 1, j)], index[(i + 1, j + 1)], index[(i, j + 1)]])
    pts = np.array(pts, np.float64)
    _, first, inverse = np.unique(np.round(pts, 9), axis=0, return_index=True, return_inverse=True)
    verts = pts[first]
    faces = inverse.reshape(-1)[np.array(tri, np.int64)]
    n_split = max(0, (tris_per_blob - len(faces)) // 2)
    if n_split:
        faces = [list(map(int, t)) for t in faces]
        edge_faces = {}
        for fi, t in enumerate(faces):
            for k in range(3):
                edge_faces.setdefault((min(t[k], t[(k + 1) % 3]), max(t[k], t[(k + 1) % 3])), []).append(fi)
        used, extra_verts, done = set(), [], 0
        n_faces0 = len(faces)
        for fi in ((k * n_faces0) // n_split for k in range(n_split)):
            # the first face at or after fi with an edge whose two triangles are both still whole
            for fj in list(range(fi, n_faces0)) + list(range(0, fi)):
                if fj in used:
                    continue
                t = faces[fj]
                pick = None
                for k in range(3):
                    a, b = t[k], t[(k + 1) % 3]
                    other = [g for g in edge_faces[(min(a, b), max(a, b))] if g != fj]
                    if len(other) == 1 and other[0] not in used:
                        pick = (a, b, t[(k + 2) % 3], other[0])
                        break
                if pick:
                    break
            a, b, c, fo = pick
            m = verts[a] + verts[b]
            extra_verts.append(m / np.linalg.norm(m))
            mi = len(verts) + len(extra_verts) - 1
            to = faces[fo]
            ko = [k for k in range(3) if to[k] == b and to[(k + 1) % 3] == a][0]     # the neighbour runs the edge the other way
            d = to[(ko + 2) % 3]
            faces[fj] = [a, mi, c]
            faces.append([mi, b, c])
            faces[fo] = [b, mi, d]
            faces.append([mi, a, d])
            used.update((fj, fo))
            done += 1
        verts = np.concatenate([verts, np.array(extra_verts, np.float64)])
        faces = np.array(faces, np.int64)
    bump = 1.0 + 0.15 * np.sin(5 * verts[:, 0]) * np.sin(4 * verts[:, 1] + 1.0) * np.sin(3 * verts[:, 2] + 2.0)
    verts = verts * bump[:, None]
    # smooth vertex normals: area-weighted face normals
    fn = np.cross(verts[faces[:, 1]] - verts[faces[:, 0]], verts[faces[:, 2]] - verts[faces[:, 0]])
    vn = np.zeros_like(verts)
    for k in range(3):
        np.add.at(vn, faces[:, k], fn)
    vn /= np.linalg.norm(vn, axis=1, keepdims=True)
    r = Rand(seed, 5)
    gx, gy, gz = grid
    n_instances = min(n_instances, gx * gy * gz)
    ii = np.arange(n_instances)
    cx = (ii % gx - (gx - 1) / 2) * spacing
    cy = ((ii // gx) % gy - (gy - 1) / 2) * spacing
    cz = (ii // (gx * gy)) * spacing + 2.0
    ang = r.u01(n_instances).astype(np.float64) * 2 * np.pi
    scale = 0.42 * spacing
    ca, sa = np.cos(ang), np.sin(ang)
    def rot(p):   # rotate about y, per instance: p [F,3,3] -> [I,F,3,3]
        x = ca[:, None, None] * p[None, :, :, 0] + sa[:, None, None] * p[None, :, :, 2]
        z = -sa[:, None, None] * p[None, :, :, 0] + ca[:, None, None] * p[None, :, :, 2]
        y = np.broadcast_to(p[None, :, :, 1], x.shape)
        return np.stack([x, y, z], -1)
    fv, fnrm = verts[faces], vn[faces]
    wv = rot(fv) * scale + np.stack([cx, cy, cz], -1)[:, None, None, :]
    wn = rot(fnrm)
    v = wv.reshape(-1, 3, 3).astype(np.float32)
    normals = _normalize(wn.reshape(-1, 3, 3).astype(np.float32))
    tangents = np.repeat(_normalize(v[:, 1] - v[:, 0])[:, None, :], 3, axis=1)
    n = len(v)
    uvs = np.tile(np.array([[0, 0], [1, 0], [0, 1]], np.float32), (n, 1, 1))
    cam = abi.default_camera()
    cam.position = abi.ErVec3(0.01, 0.02, -0.5)
    return abi.SceneData(v, normals, tangents, uvs, np.ones(n, np.float32), np.zeros(n, np.int32),
                         [abi.default_material()], hdri=sky_hdri(), camera=cam, x_res=x_res, y_res=y_res)
