"""the dependency-free host re-host (elevenrender_amd/host/eleven_server: own JSON + POSIX sockets,
the reference's wire format and command grammar on top of the C ABI), driven over localhost by the Python client
(elevenrender_amd/client.py) the way the Blender plug-in drives the reference (src/main.cpp:190-238).

Without a GPU the whole load sequence must be answered with "ok" and `--start` with a clean error reply that names the
missing device (the reference would reply "ok" and die later); on the GPU box the image fetched over the wire must equal
a direct C-ABI render of the same arrays, bit for bit, and an oracle render of them."""
import os
import subprocess

import numpy as np
import pytest

from elevenrender_amd import abi, client, render, scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "elevenrender_amd", "host")


def build_server():
    exe, src = os.path.join(HOST, "eleven_server"), os.path.join(HOST, "eleven_server.cpp")
    deps = [src] + [os.path.join(HOST, h) for h in os.listdir(HOST) if h.endswith(".hpp")]
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(f) for f in deps):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-pthread", src, "-o", exe, "-L", os.path.join(ROOT, "elevenrender_amd"),
                               "-leleven_hip", "-Wl,-rpath,$ORIGIN/.."])
    return exe


class Server:
    def __init__(self):
        self.p = subprocess.Popen([build_server(), "--port", "0", "--loopback", "--once"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        line = self.p.stdout.readline()
        assert line.startswith("listening on "), line
        self.port = int(line.split()[-1])

    def finish(self):
        try:
            self.p.wait(timeout=30)
        except subprocess.TimeoutExpired:
            self.p.kill()
            raise
        return self.p.returncode


def test_json_and_framing_edge_cases():
    """Malformed input gets an error reply or a closed session, never a crash: unknown options, a wrong payload size,
    bad JSON, a non-command message, a garbage header."""
    s = Server()
    c = client.Client(port=s.port)
    t, f, d = c.command("--no_such_option")
    assert t == "status" and d.startswith(b"error:") and b"unrecognised option" in d
    t, f, d = c.command("--load_camera", ("json", b'{"position": {"x": 1}}'))
    assert d.startswith(b"error:") and b"missing" in d
    t, f, d = c.command("--load_camera", ("json", b"{not json"))
    assert d.startswith(b"error:") and b"json" in d
    meta = dict(name="t", width=4, height=4, channels=3, color_space="LINEAR")
    t, f, d = c.command("--load_texture", client.Client._json(meta), ("float3", np.zeros(5, np.float32).tobytes()))
    assert d.startswith(b"error:") and b"payload" in d
    t, f, d = c.command('--get_pass "beauty"')                       # quoted tokens, as std::quoted reads them
    assert d.startswith(b"error:") and b"no render" in d
    c.write("data", "string", b"stray")
    assert c.read()[2].startswith(b"error:")
    c.expect_ok("--load_camera", client.Client._json(dict(position=dict(x=0, y=0, z=-1.5), rotation=dict(x=0, y=0, z=0), aperture=2.8, bokeh=False,
                                                         focus_distance=1e6, focal_length=0.035, sensor_width=0.036, sensor_height=0.024)))
    c.sock.sendall(b"garbage".ljust(1024, b"\0"))                     # not JSON: error reply, then the server drops the session
    assert c.read()[2].startswith(b"error: bad message")
    c.sock.close()
    assert s.finish() == 0


def test_recorded_session_without_a_gpu_fails_cleanly():
    if abi.load().er_device_count() > 0:
        pytest.skip("a HIP device is present: the GPU variant of this test runs instead")
    s = Server()
    c = client.Client(port=s.port)
    assert c.get_sycl_info() == {"devices": []}
    with pytest.raises(client.ProtocolError) as e:
        client.play_cornell_session(c, client.cornell_session_assets(32, 24), sample_target=2)
    assert "--start" in str(e.value) and "no HIP device" in str(e.value)      # every load before it was answered "ok"
    t, f, d = c.command("--get_info")
    assert d.startswith(b"error:")
    c.close()
    assert s.finish() == 0


def session_scene(assets, tmp_path):
    """The scene the server must have built, made directly: the same OBJ through eleven::load_obj (obj_dump), the same
    materials / texture / HDRI processing, as abi.SceneData."""
    from test_obj_cpu import build as build_dump, parse
    path = str(tmp_path / "session.obj")
    open(path, "w").write(assets["obj"])
    c, signs, mats = parse(subprocess.check_output([build_dump(), path], text=True))
    names = ["default"] + [m["name"] for m in assets["materials"]]
    mat_id = np.array([names.index(m) if m in names else 0 for m in mats], np.int32)
    materials = [abi.default_material()]
    for m in assets["materials"]:
        kw = {}
        if "albedo" in m:
            kw["albedo"] = (m["albedo"]["r"], m["albedo"]["g"], m["albedo"]["b"])
        if "emission" in m:
            kw["emission"] = (m["emission"]["r"], m["emission"]["g"], m["emission"]["b"])
        if "roughness" in m:
            kw["roughness"] = m["roughness"]
        if "metalness" in m:
            kw["metallic"] = m["metalness"]
        if m.get("albedo_map") == "checker":
            kw["albedo_tex"] = 0
        mm = abi.default_material(**kw)
        aspect = np.float32(np.sqrt(1.0 - mm.anisotropic * 0.9))
        mm.ax, mm.ay = max(np.float32(0.001), np.float32(mm.roughness) / aspect), max(np.float32(0.001), np.float32(mm.roughness) * aspect)
        materials.append(mm)
    hd = assets["hdri"]
    hd = np.roll(hd, hd.shape[1] // 2, axis=1)                       # Texture::pixel_shift(0.5, 0), src/CommandManager.cpp:188
    cam = abi.default_camera()
    cam.position = abi.ErVec3(*assets["camera"]["position"])
    return abi.SceneData(c[:, :, 0:3], c[:, :, 3:6], c[:, :, 8:11], c[:, :, 6:8], signs, mat_id, materials,
                         textures=[(assets["checker"], 8, 8, 3, 0)], hdri=(np.ascontiguousarray(hd), hd.shape[1], hd.shape[0], 3, 0),
                         camera=cam, x_res=assets["x_res"], y_res=assets["y_res"])


def test_session_scene_equals_the_generated_cornell_geometry(tmp_path):
    """CPU check of the ingest half: OBJ text -> eleven::load_obj -> arrays == the generator's arrays (positions, normals,
    uvs, material ids); only the tangents differ (MikkTSpace-style grouping at the quads' shared corners)."""
    a = client.cornell_session_assets(32, 24)
    sc, ref = session_scene(a, tmp_path), scenes.cornell(32, 24)
    assert (sc.vertices == ref.vertices).all() and np.allclose(sc.normals, ref.normals, atol=1e-6) and (sc.uvs == ref.uvs).all()
    assert (sc.material_id == ref.material_id + 1).all()             # Scene() holds the default material at index 0 (src/Scene.h:45)


@pytest.mark.gpu
def test_recorded_session_on_the_gpu_equals_direct_and_oracle_renders(tmp_path, oracle_mod):
    a = client.cornell_session_assets(48, 48)
    s = Server()
    c = client.Client(port=s.port)
    devs = c.get_sycl_info()["devices"]
    assert devs and devs[0]["is_compatible"] and devs[0]["type"] == "gpu"
    img = client.play_cornell_session(c, a, sample_target=6, device=f'{devs[0]["name"]}|{devs[0]["platform"]}')
    info = c.get_info()
    assert info["samples"] == 7 and info["gpus"] == 1 and info["samples_per_call"] >= 1
    normal = c.get_pass("normal", 48, 48)
    unknown = c.get_pass("no_such_pass", 48, 48)                     # parsePass: unknown names -> BEAUTY (src/kernel.cpp:50-73)
    assert (unknown.view(np.uint32) == img.view(np.uint32)).all()
    den = c.get_pass("denoise", 48, 48)                              # the library's filter fills the plane the reference never writes
    assert np.isfinite(den).all() and (den[..., 3] == 1).all() and den[..., :3].std() < img[..., :3].std()
    c.close()
    assert s.finish() == 0
    sc = session_scene(a, tmp_path)
    rm = render.RenderingManager(render.RenderParameters())
    rm.start_rendering(sc)
    rm.render(6)
    direct, direct_n = rm.get_pass("beauty"), rm.get_pass("normal")
    rm.close()
    assert (img.view(np.uint32) == direct.view(np.uint32)).all()
    assert (normal.view(np.uint32) == direct_n.view(np.uint32)).all()
    o = oracle_mod.Oracle(sc, math_mode=oracle_mod.MATH_ER, max_bounces=5, threads=4)
    o.render(6)
    ref = o.read_pass(0)
    o.close()
    assert (img.view(np.uint32) == ref.view(np.uint32)).all(-1).mean() >= 0.999


def test_multi_gpu_config_is_validated_and_fails_cleanly_without_a_gpu():
    """`gpus` / `devices` / `transport` of load_config (the C++ host drives several GPUs itself): bad values are refused with an
    error reply; without a device `--start` says so instead of crashing N threads."""
    s = Server()
    c = client.Client(port=s.port)
    base = dict(x_res=32, y_res=24, sample_target=2, denoise=False, device="", block_size=8)
    for bad, text in ((dict(gpus=0), b"gpus out of range"), (dict(gpus=2, devices=[0]), b"one ordinal per gpu"), (dict(gpus=65), b"gpus out of range")):
        t, f, d = c.command("--load_config", client.Client._json(dict(base, **bad)))
        assert d.startswith(b"error:") and text in d, d
    if abi.load().er_device_count() == 0:
        with pytest.raises(client.ProtocolError) as e:
            client.play_cornell_session(c, client.cornell_session_assets(32, 24), sample_target=2, gpus=3)
        assert "--start" in str(e.value) and ("no HIP device" in str(e.value) or "no gfx950 device" in str(e.value))
    c.close()
    assert s.finish() == 0


@pytest.mark.gpu
def test_three_ranks_behind_one_session_equal_the_single_gpu_frame(tmp_path):
    """north_star: the host API stays intact AND the tiles shard over the GPUs of the node.  Three ranks on this box's one GPU
    (devices [0, 0, 0], in-process transport -- the code path of `gpus: 8` on a node, only the wire differs): every plane fetched
    over the wire, the sample count and the denoised plane equal the one-GPU session's bit for bit."""
    a = client.cornell_session_assets(100, 76)                       # partial tiles on both edges
    frames = {}
    for gpus in (1, 3):
        s = Server()
        c = client.Client(port=s.port)
        extra = dict(gpus=3, devices=[0, 0, 0], transport="local") if gpus == 3 else {}
        img = client.play_cornell_session(c, a, sample_target=5, **extra)
        info = c.get_info()
        assert info["samples"] == 6 and info["gpus"] == gpus
        if gpus == 3:
            assert info["transport"] == "in-process"
        frames[gpus] = dict(beauty=img, normal=c.get_pass("normal", 100, 76), tangent=c.get_pass("tangent", 100, 76), denoise=c.get_pass("denoise", 100, 76))
        c.close()
        assert s.finish() == 0
    for k in frames[1]:
        assert (frames[1][k].view(np.uint32) == frames[3][k].view(np.uint32)).all(), k
    # a transport the config names but the box cannot give: refused at --start with a reason
    s = Server()
    c = client.Client(port=s.port)
    with pytest.raises(client.ProtocolError) as e:
        client.play_cornell_session(c, a, sample_target=2, gpus=2, devices=[0, 0], transport="rccl")
    assert "one device per rank" in str(e.value)
    c.close()
    assert s.finish() == 0
