"""The C-ABI library on a machine without a GPU: it loads, exports every declared symbol,
validates arguments, refuses to compute (no CPU fallback), and its host-side BVH builder is sound."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from elevenrender_amd import abi, render, scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(er_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = abi.load()
    names = declared_symbols("eleven_hip.h")
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/eleven_hip.h but not exported"
        assert n in abi.SYMBOLS, f"{n} has no ctypes prototype in abi.py"
    for n in declared_symbols("eleven_hip_debug.h"):
        assert hasattr(lib, n)
    assert lib.er_abi_version() == 2


def test_struct_layouts_match_the_header():
    # sizes the C compiler gives the PODs (checked against a tiny C program's output at build time would be
    # circular; these are the natural-alignment sizes of the declared fields)
    assert C.sizeof(abi.ErVec3) == 12
    assert C.sizeof(abi.ErCamera) == 5 * 4 + 12 + 4 + 12
    assert C.sizeof(abi.ErMaterial) == 8 * 4 + 24 + 15 * 4
    assert C.sizeof(abi.ErTexture) == 24
    assert C.sizeof(abi.ErCounters) == 96
    assert C.sizeof(abi.ErRenderParams) == 28


def test_argument_validation_and_error_text():
    lib = abi.load()
    h = C.c_void_p()
    assert lib.er_scene_create(None, C.byref(h)) == abi.ER_ERR_INVALID_ARG
    assert b"NULL" in lib.er_last_error()
    sc = scenes.cornell(16, 16)
    d = sc.desc()
    d.x_res = 0
    assert lib.er_scene_create(C.byref(d), C.byref(h)) == abi.ER_ERR_INVALID_ARG
    d.x_res = 16
    sc.material_id[3] = 99
    assert lib.er_scene_create(C.byref(d), C.byref(h)) == abi.ER_ERR_INVALID_ARG
    assert b"material_id" in lib.er_last_error()
    sc.material_id[3] = 0
    assert lib.er_scene_create(C.byref(d), C.byref(h)) == abi.ER_OK
    # calls before er_render_begin are state errors, never crashes
    buf = np.zeros(16 * 16 * 4, np.float32)
    assert lib.er_render_samples(h, 1) == abi.ER_ERR_STATE
    assert lib.er_read_pass(h, 0, buf.ctypes.data_as(C.POINTER(C.c_float))) == abi.ER_ERR_STATE
    assert lib.er_read_pass(h, 7, buf.ctypes.data_as(C.POINTER(C.c_float))) == abi.ER_ERR_INVALID_ARG
    n = C.c_uint64()
    assert lib.er_state_size(h, C.byref(n)) == abi.ER_OK and n.value == 64 + 16 * 16 * (5 * 16 + 8)
    assert lib.er_state_export(h, buf.ctypes.data_as(C.c_void_p), n.value) == abi.ER_ERR_STATE
    p = abi.ErRenderParams(16, 8, 5, 0, 3, 2, 0)          # rank >= world
    assert lib.er_render_begin(h, C.byref(p)) == abi.ER_ERR_INVALID_ARG
    lib.er_scene_destroy(h)
    lib.er_scene_destroy(None)


def test_no_cpu_fallback_without_a_device():
    lib = abi.load()
    if lib.er_device_count() > 0:
        pytest.skip("a HIP device is present")
    rm = render.RenderingManager()
    with pytest.raises(abi.ErError) as e:
        rm.start_rendering(scenes.cornell(16, 16))
    assert e.value.code == abi.ER_ERR_NO_DEVICE
    assert render.list_devices() == []


class ErBvhCheck(C.Structure):
    _fields_ = [("node_count", C.c_uint32), ("leaf_count", C.c_uint32), ("max_depth", C.c_uint32),
                ("max_leaf_size", C.c_uint32), ("tris_in_leaves", C.c_uint32), ("duplicate_tris", C.c_uint32),
                ("uncontained", C.c_uint32), ("unreachable_nodes", C.c_uint32), ("lift_bound", C.c_float),
                ("build_ms", C.c_float), ("sah_cost", C.c_double)]


def bvh_check(verts, normals, threads=0):
    lib = abi.load()
    lib.er_debug_bvh_check.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_uint32, C.c_int, C.POINTER(ErBvhCheck)]
    out = ErBvhCheck()
    v = np.ascontiguousarray(verts, np.float32)
    n = np.ascontiguousarray(normals, np.float32)
    P = C.POINTER(C.c_float)
    assert lib.er_debug_bvh_check(v.ctypes.data_as(P), n.ctypes.data_as(P), len(v), threads, C.byref(out)) == 0
    return out


@pytest.mark.parametrize("n_tris", [0, 1, 3, 4, 5, 12, 1000, 100000])
def test_bvh_builder_invariants(n_tris):
    if n_tris == 12:
        sc = scenes.cornell(8, 8)
        v, nn = sc.vertices, sc.normals
    else:
        v = scenes.soup_geometry(n_tris, seed=n_tris + 1)
        nn, _ = scenes.face_frame(v) if n_tris else (np.zeros((0, 3, 3), np.float32), None)
    c = bvh_check(v, nn)
    assert c.tris_in_leaves == n_tris and c.duplicate_tris == 0 and c.uncontained == 0 and c.unreachable_nodes == 0
    assert c.max_leaf_size <= 4 and c.max_depth <= 32
    if n_tris > 4:
        assert c.node_count == c.leaf_count - 1
    if n_tris >= 1000:
        assert c.lift_bound < 1e-4          # flat-shaded soup: shading position == geometric position up to rounding


def test_bvh_builder_degenerate_inputs_stay_bounded():
    # all triangles identical (every centroid coincides): SAH cannot split, median fallback must
    n = 5000
    one = np.array([[0, 0, 3], [1, 0, 3], [0, 1, 3]], np.float32)
    v = np.repeat(one[None], n, axis=0)
    nn, _ = scenes.face_frame(v)
    c = bvh_check(v, nn)
    assert c.tris_in_leaves == n and c.duplicate_tris == 0 and c.max_depth <= 32
    # a long thin line of triangles (worst case for unbalanced SAH splits)
    x = np.cumsum(np.geomspace(1e-6, 1.0, n)).astype(np.float32)
    v = one[None] * 1e-3 + np.stack([x, np.zeros(n, np.float32), np.zeros(n, np.float32)], -1)[:, None, :]
    c = bvh_check(v.astype(np.float32), nn)
    assert c.tris_in_leaves == n and c.max_depth <= 32 and c.uncontained == 0
    # smooth normals: the lift bound is positive and bounded by the triangle size
    sc = scenes.blob_instances(n_instances=2, tris_per_blob=200, x_res=8, y_res=8)
    c = bvh_check(sc.vertices, sc.normals)
    assert 0 < c.lift_bound < 0.05 and c.uncontained == 0


def test_bvh_builder_threaded_equals_serial():
    v = scenes.soup_geometry(60000, seed=5)
    nn, _ = scenes.face_frame(v)
    a, b = bvh_check(v, nn, threads=1), bvh_check(v, nn, threads=8)
    assert (a.node_count, a.leaf_count, a.max_depth) == (b.node_count, b.leaf_count, b.max_depth)
    assert abs(a.sah_cost - b.sah_cost) < 1e-9 * max(1.0, a.sah_cost)


def test_out_of_memory_comes_back_as_a_status_code_not_an_exception():
    """The header promises "never throws": a std::bad_alloc inside an entry point must surface as ER_ERR_OOM.  The
    allocator limit of include/eleven_hip_debug.h makes the scene copy of a valid, small description fail."""
    lib = abi.load()
    lib.er_debug_set_host_alloc_limit.argtypes = [C.c_uint64]
    lib.er_debug_set_host_alloc_limit.restype = None
    sc = scenes.soup(2000, 32, 24, seed=2, hdri_size=(64, 32))
    h = C.c_void_p()
    lib.er_debug_set_host_alloc_limit(1024)
    try:
        rc = lib.er_scene_create(C.byref(sc.desc()), C.byref(h))
        assert rc == abi.ER_ERR_OOM, rc
        assert b"out of host memory" in lib.er_last_error()
        assert not h.value
    finally:
        lib.er_debug_set_host_alloc_limit(0)
    assert lib.er_scene_create(C.byref(sc.desc()), C.byref(h)) == abi.ER_OK      # and the library is still usable
    lib.er_scene_destroy(h)


def test_shared_library_carries_gfx950_code_objects_only():
    """Round 1 recorded a host crash inside the HIP runtime from a build of the .so that held no code object the
    device could run ("No compatible code objects found for gfx950:sramecc+:xnack-").  er_render_begin now probes the
    kernels (hipFuncGetAttributes) and returns ER_ERR_HIP; this test catches such a build before it travels: every
    clang offload bundle in the library must hold exactly one device entry, for plain gfx950 (no xnack+/sramecc-
    feature suffix, which the pool's xnack- devices refuse)."""
    blob = open(abi.LIB_PATH, "rb").read()
    bundles = blob.count(b"__CLANG_OFFLOAD_BUNDLE__")
    assert bundles >= 4, bundles            # er_kernels, er_wavefront, er_stream, er_gpu_build, er_debug
    entries = re.findall(rb"hip[v0-9]*-amdgcn-amd-amdhsa-[-A-Za-z0-9_:+]*", blob)
    assert len(entries) >= bundles
    for e in entries:
        assert e.endswith(b"--gfx950"), e


def test_collective_entry_points_validate_without_a_gpu():
    lib = abi.load()
    assert lib.er_gather_pass(None, 0, None, 0) == abi.ER_ERR_INVALID_ARG
    comms = (C.c_void_p * 2)()
    assert lib.er_debug_comm_create_local(2, comms) == abi.ER_OK      # host objects only
    sc = scenes.cornell(16, 16)
    h = C.c_void_p()
    assert lib.er_scene_create(C.byref(sc.desc()), C.byref(h)) == abi.ER_OK
    assert lib.er_gather_pass(h, 0, comms[0], 0) == abi.ER_ERR_STATE   # not begun
    assert lib.er_gather_pass(h, 9, comms[0], 0) == abi.ER_ERR_INVALID_ARG
    lib.er_scene_destroy(h)
    for c in comms:
        lib.er_comm_destroy(c)
    lib.er_comm_destroy(None)
    ident = (C.c_uint8 * 128)()
    comm = C.c_void_p()
    if lib.er_device_count() == 0:       # a communicator needs a device; the failure is a status code with a text
        rc = lib.er_comm_create(ident, 0, 1, 0, C.byref(comm))
        assert rc in (abi.ER_ERR_NO_DEVICE, abi.ER_ERR_HIP) and lib.er_last_error()


def test_no_instruction_touches_a_load_destination_between_its_issue_and_its_wait():
    """The streaming kernel's traversal step issues its eleven loads in one asm statement and waits for them in two later ones
    (csrc/er_trav.h trav_fetch_issue / trav_wait_tri / trav_wait_node: the triangle block runs while the node pieces arrive).  The
    compiler does not know those loads are in flight, so a copy, spill or reuse of one of their destination registers in between
    would be silent corruption on some waves of some launches.  tools/check_split_wait.py reads the device assembly the build
    leaves beside the object (-save-temps=obj) and must find every site clean."""
    import glob
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    asm = glob.glob(os.path.join(root, "elevenrender_amd", "csrc", "build", "er_stream-hip-amdgcn-*gfx950.s"))
    assert asm, "the build did not leave er_stream's device assembly (elevenrender_amd/csrc/Makefile: -save-temps=obj)"
    so = os.path.join(root, "elevenrender_amd", "libeleven_hip.so")
    assert os.path.getmtime(asm[0]) <= os.path.getmtime(so) + 1, "the assembly is newer than the library: rebuild"
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "check_split_wait.py"), asm[0]], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout[-3000:]
    assert "40 split-wait sites, 0 offending instructions" in p.stdout      # 8 instantiations (counters, extensions, fused textures) x 5 forms (16 waves: plain, keep, keep + speculative samples; 12 waves: plain, speculative)


def stream_deal(owned, tiles_x, blocks=256, xcd_aware=1, edge=0):
    lib = abi.load()
    U = C.POINTER(C.c_uint32)
    lib.er_debug_stream_deal.argtypes = [U, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, U, C.c_uint32, U]
    lib.er_debug_stream_deal.restype = C.c_int
    owned = np.ascontiguousarray(owned, np.uint32)
    most = C.c_uint32()
    assert lib.er_debug_stream_deal(owned.ctypes.data_as(U), len(owned), tiles_x, blocks, xcd_aware, edge, None, 0, C.byref(most)) == (abi.ER_ERR_INVALID_ARG if len(owned) else abi.ER_OK)
    out = np.zeros(blocks * most.value, np.uint32)
    assert lib.er_debug_stream_deal(owned.ctypes.data_as(U), len(owned), tiles_x, blocks, xcd_aware, edge, out.ctypes.data_as(U), len(out), C.byref(most)) == abi.ER_OK
    return out.reshape(most.value, blocks)          # [k, b] = the k-th tile of workgroup b


@pytest.mark.parametrize("edge", [0, 1, 8, 16, 32, 100])
@pytest.mark.parametrize("frame", [(240, 135, 1, 0), (480, 270, 1, 0), (240, 135, 8, 3), (37, 5, 1, 0), (240, 135, 128, 37)])
def test_stream_deal_is_a_levelled_partition_of_the_owned_tiles(frame, edge):
    """The streaming schedule's deal of tiles to workgroups (er_stream.hip er_stream_deal_tiles; host code): every owned tile goes to
    exactly one workgroup, the eight XCDs (workgroups b, b + 8, ...) get tile counts that differ by at most one whatever the
    super-tile edge (round 4: whole super-tiles had left them up to one super-tile apart, and a launch lasts as long as its fullest
    XCD), and the workgroups of an XCD differ by at most one tile."""
    tiles_x, tiles_y, world, rank = frame
    t = np.arange(tiles_x * tiles_y, dtype=np.uint32)
    owned = t[((t % tiles_x) + (t // tiles_x)) % world == rank]          # the library's sharding: (tx + ty) % world
    deal = stream_deal(owned, tiles_x, edge=edge)
    got = deal[deal != 0xFFFFFFFF]
    assert len(got) == len(owned) and (np.sort(got) == owned).all()
    per_wg = (deal != 0xFFFFFFFF).sum(0)
    per_xcd = per_wg.reshape(-1, 8).sum(0)
    assert per_xcd.max() - per_xcd.min() <= 1, per_xcd
    for x in range(8):
        w = per_wg[x::8]
        assert w.max() - w.min() <= 1, (x, w)
    assert per_wg.max() == deal.shape[0]
    # super-tile locality: with the default edge the tiles of an XCD come from far fewer super-tiles than a round-robin deal would touch
    if edge == 0 and world == 1 and tiles_x >= 240:
        e = 8
        sup = lambda tiles: len(np.unique((tiles // tiles_x // e) * ((tiles_x + e - 1) // e) + (tiles % tiles_x) // e))
        x0 = deal[:, 0::8]
        n_super_total = sup(owned)
        assert sup(x0[x0 != 0xFFFFFFFF]) <= n_super_total // 8 + 8
