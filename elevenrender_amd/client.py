"""A client of the host's wire protocol -- what the Blender plug-in does, as a script.

Protocol (reference src/TCPInterface.cpp:3-58, src/Managers.cpp:6-177, src/main.cpp:36-238): every message is a
1024-byte NUL-padded JSON header {"type", "data_format", "data_size"} followed by data_size payload bytes.  A session:
the server greets with STATUS "ok"; the client sends COMMAND messages (a command line such as `--load_camera`), each
load command followed by its DATA messages; the server answers every command with one message; the client ends with
STATUS "close_session".  Works against elevenrender_amd/host/eleven_server (this build's host) and, by construction of
the format, against the reference's own server.

    python -m elevenrender_amd.client --port 5557 --demo out.npy      # plays the recorded Cornell session
"""
import json
import socket
import struct

import numpy as np

HEADER = 1024


class ProtocolError(RuntimeError):
    pass


class Client:
    def __init__(self, host="127.0.0.1", port=5557, timeout=120.0):
        self.sock = socket.create_connection((host, port), timeout=timeout)
        self.sock.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
        greeting = self.read()
        if greeting != ("status", "string", b"ok"):
            raise ProtocolError(f"unexpected greeting {greeting[:2]}")

    # ---- framing ----
    def write(self, mtype, fmt, payload=b""):
        header = json.dumps({"type": mtype, "data_format": fmt, "data_size": len(payload)}).encode()
        self.sock.sendall(header.ljust(HEADER, b"\0"))
        if payload:
            self.sock.sendall(payload)

    def _recv(self, n):
        buf = bytearray()
        while len(buf) < n:
            chunk = self.sock.recv(min(1 << 20, n - len(buf)))
            if not chunk:
                raise ProtocolError("connection closed by the server")
            buf += chunk
        return bytes(buf)

    def read(self):
        h = json.loads(self._recv(HEADER).split(b"\0", 1)[0].decode())
        data = self._recv(h["data_size"]) if h["data_size"] else b""
        return h["type"], h["data_format"], data

    # ---- commands ----
    def command(self, line, *follow):
        """Send a command line and its follow-up DATA messages ((format, payload) pairs); return the one reply."""
        self.write("command", "string", line.encode())
        for fmt, payload in follow:
            self.write("data", fmt, payload)
        return self.read()

    def expect_ok(self, line, *follow):
        t, f, d = self.command(line, *follow)
        if (t, d) != ("status", b"ok"):
            raise ProtocolError(f"{line}: {d[:300].decode('utf-8', 'replace')}")

    @staticmethod
    def _json(obj):
        return "json", json.dumps(obj).encode()

    def load_config(self, x_res, y_res, sample_target, device="", denoise=False, block_size=8, **extra):
        self.expect_ok("--load_config", self._json(dict(x_res=x_res, y_res=y_res, sample_target=sample_target, denoise=denoise,
                                                        device=device, block_size=block_size, **extra)))

    def load_camera(self, position, rotation=(0, 0, 0), aperture=2.8, bokeh=False, focus_distance=1e6, focal_length=0.035,
                    sensor_width=0.036, sensor_height=0.024):
        xyz = lambda v: dict(x=float(v[0]), y=float(v[1]), z=float(v[2]))
        self.expect_ok("--load_camera", self._json(dict(position=xyz(position), rotation=xyz(rotation), aperture=aperture, bokeh=bokeh,
                                                        focus_distance=focus_distance, focal_length=focal_length,
                                                        sensor_width=sensor_width, sensor_height=sensor_height)))

    def _texture(self, cmd, name, pixels, color_space):
        px = np.ascontiguousarray(pixels, np.float32)
        h, w, ch = px.shape
        meta = dict(name=name, width=w, height=h, channels=ch, color_space=color_space)
        self.expect_ok(cmd, self._json(meta), ("float4" if ch == 4 else "float3", px.tobytes()))

    def load_texture(self, name, pixels, color_space="LINEAR"):
        self._texture("--load_texture", name, pixels, color_space)

    def load_hdri(self, pixels, mirror_x=False, mirror_y=False):
        self._texture("--load_hdri" + (" --mirror_x" if mirror_x else "") + (" --mirror_y" if mirror_y else ""), "hdri", pixels, "LINEAR")

    def load_brdf_material(self, **mat):
        self.expect_ok("--load_brdf_material", self._json(mat))

    def load_object(self, obj_text, mtl_text="", recompute_normals=False):
        self.expect_ok("--load_object" + (" --recompute_normals" if recompute_normals else ""),
                       ("string", obj_text.encode()), ("string", mtl_text.encode() or b"\n"))

    def start(self):
        self.expect_ok("--start")

    def get_info(self):
        t, f, d = self.command("--get_info")
        if (t, f) != ("data", "json"):
            raise ProtocolError(d[:300].decode("utf-8", "replace"))
        return json.loads(d.decode())

    def get_sycl_info(self):
        t, f, d = self.command("--get_sycl_info")
        if (t, f) != ("data", "json"):
            raise ProtocolError(d[:300].decode("utf-8", "replace"))
        return json.loads(d.decode())

    def get_pass(self, name, x_res, y_res):
        t, f, d = self.command(f"--get_pass {name}")
        if (t, f) != ("data", "float4"):
            raise ProtocolError(d[:300].decode("utf-8", "replace"))
        return np.frombuffer(d, np.float32).reshape(y_res, x_res, 4).copy()

    def close(self):
        try:
            self.write("status", "string", b"close_session")
        finally:
            self.sock.close()


# ---- the recorded demo session: Cornell box as OBJ + MTL, one textured floor, a small HDRI ----
def cornell_session_assets(x_res=48, y_res=48):
    """Everything the demo session sends, as plain data (also used by the tests to build the same scene directly)."""
    from . import scenes
    sc = scenes.cornell(x_res, y_res)
    names = ["floor", "red", "green", "light"]
    v, n, uv = sc.vertices.reshape(-1, 3, 3), sc.normals.reshape(-1, 3, 3), sc.uvs.reshape(-1, 3, 2)
    lines = ["# cornell box", "mtllib cornell.mtl", "o box"]
    for t in range(len(v)):
        for j in range(3):
            lines.append("v %.9g %.9g %.9g" % (v[t, j, 0], v[t, j, 1], -v[t, j, 2]))       # the loader flips z (src/ObjLoader.cpp:116)
            lines.append("vn %.9g %.9g %.9g" % (n[t, j, 0], n[t, j, 1], -n[t, j, 2]))
            lines.append("vt %.9g %.9g" % (uv[t, j, 0], uv[t, j, 1]))
    for t in range(len(v)):
        lines.append("usemtl %s" % names[int(sc.material_id[t])])
        a = 3 * t + 1
        lines.append("f %d/%d/%d %d/%d/%d %d/%d/%d" % (a, a, a, a + 1, a + 1, a + 1, a + 2, a + 2, a + 2))
    obj = "\n".join(lines) + "\n"
    mtl = "".join(f"newmtl {nm}\nKd 0.5 0.5 0.5\n" for nm in names)
    yy, xx = np.mgrid[0:8, 0:8]
    checker = np.where(((xx + yy) % 2)[..., None] == 0, np.float32(0.9), np.float32(0.2)) * np.array([1.0, 0.9, 0.7], np.float32)
    r = np.random.default_rng(3)
    hdri = (0.3 + r.random((8, 16, 3))).astype(np.float32)
    materials = [dict(name="floor", albedo=dict(r=1.0, g=1.0, b=1.0), roughness=0.8, albedo_map="checker"),
                 dict(name="red", albedo=dict(r=0.8, g=0.1, b=0.1)),
                 dict(name="green", albedo=dict(r=0.1, g=0.8, b=0.1), metalness=0.3, roughness=0.4),
                 dict(name="light", emission=dict(r=5.0, g=5.0, b=5.0))]
    camera = dict(position=(0.0, 0.0, -1.5))
    return dict(obj=obj, mtl=mtl, checker=checker.astype(np.float32), hdri=hdri, materials=materials, camera=camera, x_res=x_res, y_res=y_res)


def play_cornell_session(client, assets, sample_target=6, wait=True, **config_extra):
    import time
    a = assets
    client.load_config(a["x_res"], a["y_res"], sample_target, **config_extra)
    client.load_camera(**a["camera"])
    client.load_hdri(a["hdri"])
    client.load_texture("checker", a["checker"])
    for m in a["materials"]:
        client.load_brdf_material(**m)
    client.load_object(a["obj"], a["mtl"])
    client.start()
    if wait:
        t0 = time.time()
        while client.get_info()["samples"] < sample_target + 1:      # reference semantic: dev_samples[0] = samples done + 1
            if time.time() - t0 > 120:
                raise ProtocolError("render did not reach the sample target")
            time.sleep(0.01)
    return client.get_pass("beauty", a["x_res"], a["y_res"])


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--host", default="127.0.0.1")
    ap.add_argument("--port", type=int, default=5557)
    ap.add_argument("--demo", metavar="OUT.npy", help="play the recorded Cornell session and save the beauty pass")
    args = ap.parse_args()
    c = Client(args.host, args.port)
    print(json.dumps(c.get_sycl_info()))
    if args.demo:
        img = play_cornell_session(c, cornell_session_assets())
        np.save(args.demo, img)
        print("beauty mean", float(img[..., :3].mean()))
    c.close()
