"""The C++ host mirror (elevenrender_amd/host/eleven_host.hpp) and the raw C ABI driven from C++ on the GPU:
same Cornell scene through the C++ Scene/RenderingManager classes and through the Python mirror must give
bit-identical images."""
import os
import subprocess

import numpy as np
import pytest

from elevenrender_amd import render, scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NATIVE = os.path.join(ROOT, "tests", "native")


def build(name):
    exe = os.path.join(NATIVE, name)
    src = exe + ".cpp"
    hdrs = [os.path.join(ROOT, "elevenrender_amd", "host", h) for h in ("eleven_host.hpp", "eleven_obj.hpp")]
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(f) for f in [src] + hdrs):
        subprocess.check_call(["g++", "-O1", "-std=c++17", src, "-o", exe, "-L", os.path.join(ROOT, "elevenrender_amd"),
                               "-leleven_hip", "-Wl,-rpath,$ORIGIN/../../elevenrender_amd"])
    return exe


def fnv1a(img):
    h = 1469598103934665603
    for u in img.reshape(-1).view(np.uint32).tolist():
        h = ((h ^ u) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


@pytest.mark.gpu
def test_cpp_host_mirror_matches_python_mirror():
    exe = build("host_cornell")
    out = subprocess.check_output([exe, "48", "5"], text=True)
    assert "samples 6" in out
    rm = render.RenderingManager()
    rm.start_rendering(scenes.cornell(48, 48))
    rm.render(5)
    img = rm.get_pass("beauty")
    rm.close()
    assert f"fnv1a {fnv1a(img):016x}" in out, out


@pytest.mark.gpu
def test_obj_file_through_cpp_host_against_the_oracle(tmp_path, oracle_mod):
    """BASELINE config 1 names a Cornell-box .obj: the box written as OBJ text, read by eleven::load_obj
    (elevenrender_amd/host/eleven_obj.hpp) and rendered through the C++ host mirror, against the ORACLE fed the arrays the
    loader produced (dumped by tests/native/obj_dump: positions, normals, uvs and the MikkTSpace-style tangents) -- not
    against another run of the HIP path."""
    from test_obj_cpu import build as build_dump, parse, write_obj
    from elevenrender_amd import abi
    sc = scenes.cornell(48, 48)
    path = str(tmp_path / "cornell.obj")
    names = ["default", "red", "green", "light"]
    write_obj(path, sc, names=names)
    exe = build("host_cornell")
    out = subprocess.check_output([exe, "48", "5", path], text=True)
    c, signs, mats = parse(subprocess.check_output([build_dump(), path], text=True))
    loaded = abi.SceneData(c[:, :, 0:3], c[:, :, 3:6], c[:, :, 8:11], c[:, :, 6:8], signs, np.array([names.index(m) for m in mats], np.int32),
                           sc.materials, camera=sc.camera, x_res=48, y_res=48)
    # (mikktspace.c's grouping needs edge connectivity: the two triangles of a wall are not neighbours across their diagonal --
    # its uvs differ on the two sides -- so every corner keeps its own triangle's dP/du, which is the generator's tangent)
    assert np.allclose(loaded.tangents, sc.tangents, atol=1e-5)
    o = oracle_mod.Oracle(loaded, math_mode=oracle_mod.MATH_ER, max_bounces=5, threads=4)
    o.render(5)
    ref = o.read_pass(0)
    o.close()
    assert f"fnv1a {fnv1a(ref):016x}" in out, out


@pytest.mark.gpu
def test_c_abi_smoke_binary():
    exe = build("abi_smoke")
    out = subprocess.check_output([exe, "4"], text=True)
    assert "samples_done 5" in out and "paths 3072" in out


def test_native_drivers_compile_and_fail_loudly_without_gpu():
    from elevenrender_amd import abi
    for name in ("host_cornell", "abi_smoke"):
        exe = build(name)
        if abi.load().er_device_count() == 0:
            p = subprocess.run([exe], capture_output=True, text=True)
            assert p.returncode != 0 and "no HIP device" in (p.stderr + p.stdout)
