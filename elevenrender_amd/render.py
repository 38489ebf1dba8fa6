"""Python host mirror of the reference's device-boundary objects, on top of the C ABI.

Mirrors (names and call semantics) the part of the reference host API that drives the
hot path -- RenderParameters (reference src/kernel.h:51-69) and RenderingManager
(src/Managers.h:41-66: start_rendering / get_pass / get_render_info) -- so that a driver
written against the reference reads the same.  The production host is C++
(elevenrender_amd/host/); this mirror exists for tests, bench.py and scripting.  It never
computes anything itself: every call goes through libeleven_hip.so and fails loudly if
that library or a HIP device is missing.
"""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import abi


@dataclass
class RenderParameters:
    """reference RenderParameters (width/height/sampleTarget/denoise/device/block_size) plus
    the two knobs the MI355X build adds with reference defaults (max_bounces 5, one GPU)."""
    width: int = 1280
    height: int = 720
    sampleTarget: int = 100
    denoise: bool = False
    device: str = ""          # "name|platform" (src/Managers.cpp:201) or "" / "hip:N" for ordinal N
    block_size: int = 8
    max_bounces: int = 5
    rank: int = 0
    world: int = 1
    flags: int = 0


@dataclass
class RenderInfo:
    samples: int = 0


def list_devices():
    """get_sycl_info equivalent (src/CommandManager.cpp:303-362): one dict per HIP device."""
    lib = abi.load()
    out = []
    for i in range(lib.er_device_count()):
        info = abi.ErDeviceInfo()
        abi.check(lib.er_device_info(i, C.byref(info)))
        out.append({"name": info.name.decode(), "platform": info.platform.decode(),
                    "memory": int(info.memory_bytes), "max_compute_units": int(info.compute_units),
                    "is_compatible": bool(info.compatible), "online_compiler": False, "type": "gpu",
                    "arch": info.arch.decode()})
    return out


def measure_hbm_peak(device=0, nbytes=0, iters=0):
    """er_measure_hbm_peak: (copy GB/s counting read + write, read-only GB/s) of a streaming kernel over `nbytes`."""
    lib = abi.load()
    a, b = C.c_float(), C.c_float()
    abi.check(lib.er_measure_hbm_peak(device, nbytes, iters, C.byref(a), C.byref(b)))
    return a.value, b.value


class RenderingManager:
    """start_rendering(scene) / render(n) / get_pass(name) / get_render_info() over the C ABI."""

    def __init__(self, pars: RenderParameters = None):
        self.pars = pars or RenderParameters()
        self.lib = abi.load()
        self.handle = C.c_void_p()
        self.scene = None

    # -- reference: RenderingManager::start_rendering(Scene*) (src/Managers.cpp:234-275).  The
    #    reference also spawns the render thread here; callers of this mirror call render().
    def start_rendering(self, scene: abi.SceneData):
        self.close()
        self.scene = scene
        self.pars.width, self.pars.height = scene.x_res, scene.y_res
        abi.check(self.lib.er_scene_create(C.byref(scene.desc()), C.byref(self.handle)))
        dev = 0
        sel = self.pars.device
        if sel.startswith("hip:"):
            dev = int(sel[4:])
        elif sel:
            dev = self.lib.er_device_find(sel.encode())
            if dev < 0:
                abi.check(dev)
        p = abi.ErRenderParams(self.pars.sampleTarget, self.pars.block_size, self.pars.max_bounces, dev,
                               self.pars.rank, self.pars.world, self.pars.flags)
        abi.check(self.lib.er_render_begin(self.handle, C.byref(p)))

    # -- reference: kernel_render_enqueue's sample loop (src/kernel.cpp:689-700)
    def render(self, n_samples, blocking=True):
        if blocking:
            abi.check(self.lib.er_render_samples(self.handle, n_samples))
        else:
            abi.check(self.lib.er_render_samples_async(self.handle, n_samples))

    def wait(self):
        ms = C.c_float()
        abi.check(self.lib.er_wait(self.handle, C.byref(ms)))
        return ms.value

    # -- reference: RenderingManager::get_pass(std::string) (src/Managers.cpp:287-302, parsePass kernel.cpp:50-73)
    def get_pass(self, name="beauty"):
        p = abi.PASS_NAMES.get(str(name).lower(), abi.PASS_BEAUTY)   # unknown names -> BEAUTY, as parsePass
        out = np.empty((self.scene.y_res, self.scene.x_res, 4), np.float32)
        abi.check(self.lib.er_read_pass(self.handle, p, out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    # -- reference: RenderingManager::get_render_info (src/Managers.cpp:211-232)
    def get_render_info(self):
        v = C.c_uint32()
        abi.check(self.lib.er_samples_done(self.handle, C.byref(v)))
        return RenderInfo(samples=v.value)

    def denoise(self, levels=0, colour_sigma=0.0):
        """Fill the DENOISE plane from BEAUTY + NORMAL (er_denoise); get_pass("denoise") then returns it."""
        abi.check(self.lib.er_denoise(self.handle, levels, colour_sigma))

    def read_samples(self):
        out = np.empty(self.scene.x_res * self.scene.y_res, np.uint32)
        abi.check(self.lib.er_read_samples(self.handle, out.ctypes.data_as(C.POINTER(C.c_uint32))))
        return out

    def read_rng(self):
        out = np.empty(self.scene.x_res * self.scene.y_res, np.uint32)
        abi.check(self.lib.er_read_rng(self.handle, out.ctypes.data_as(C.POINTER(C.c_uint32))))
        return out

    def counters(self):
        c = abi.ErCounters()
        abi.check(self.lib.er_get_counters(self.handle, C.byref(c)))
        return {n: int(getattr(c, n)) for n, _ in abi.ErCounters._fields_}

    def profile(self):
        pr = abi.ErProfile()
        abi.check(self.lib.er_get_profile(self.handle, C.byref(pr)))
        return {n: getattr(pr, n) for n, _ in abi.ErProfile._fields_}

    def stream_info(self):
        """include/eleven_hip_debug.h: the streaming schedule's configuration and readings of the last completed call."""
        si = abi.ErStreamInfo()
        abi.check(self.lib.er_debug_stream_info(self.handle, C.byref(si)))
        return {n: getattr(si, n) for n, _ in abi.ErStreamInfo._fields_}

    def accel_info(self):
        a = abi.ErAccelInfo()
        abi.check(self.lib.er_accel_info(self.handle, C.byref(a)))
        return {n: getattr(a, n) for n, _ in abi.ErAccelInfo._fields_}

    def debug_closest_hit(self, origins, dirs):
        """include/eleven_hip_debug.h: (triangle id, Hit.position, distance) of arbitrary rays through the exact routine."""
        o = np.ascontiguousarray(origins, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
        n = len(o)
        tri = np.empty(n, np.int32)
        pos = np.empty((n, 3), np.float32)
        dist = np.empty(n, np.float32)
        fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
        abi.check(self.lib.er_debug_closest_hit(self.handle, fp(o), fp(d), n, tri.ctypes.data_as(C.POINTER(C.c_int32)), fp(pos), fp(dist)))
        return tri, pos, dist

    def debug_trace_rays(self, origins, dirs, self_slots=None, limits=None):
        """include/eleven_hip_debug.h: rays through the PRODUCTION traversal (er_trav.h + resolve_closest/resolve_shadow).
        Closest queries return (tri, slot, pos, dist, info); shadow queries (self_slots given) return (occluded, info)."""
        o = np.ascontiguousarray(origins, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
        n = len(o)
        tri, slot, info = np.empty(n, np.int32), np.empty(n, np.int32), np.empty(n, np.int32)
        pos, dist = np.empty((n, 3), np.float32), np.empty(n, np.float32)
        fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
        ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
        if self_slots is None:
            abi.check(self.lib.er_debug_trace_rays(self.handle, fp(o), fp(d), n, None, None, ip(tri), ip(slot), fp(pos), fp(dist), ip(info)))
            return tri, slot, pos, dist, info
        ss = np.ascontiguousarray(self_slots, np.int32)
        lim = np.ascontiguousarray(limits, np.float32)
        abi.check(self.lib.er_debug_trace_rays(self.handle, fp(o), fp(d), n, ip(ss), fp(lim), ip(tri), ip(slot), fp(pos), fp(dist), ip(info)))
        return tri.astype(bool), info

    def debug_eval(self, kind, items, out_width):
        """include/eleven_hip_debug.h er_debug_eval: one DEVICE function of the path per row of `items` ([n, in_width] float32;
        integer inputs as float bits).  Returns [n, out_width] float32."""
        a = np.ascontiguousarray(items, np.float32)
        out = np.zeros((a.shape[0], out_width), np.float32)
        fp = lambda x: x.ctypes.data_as(C.POINTER(C.c_float))
        abi.check(self.lib.er_debug_eval(self.handle, kind, fp(a), a.shape[0], a.shape[1], fp(out), out_width))
        return out

    def debug_trace_pixel(self, idx, max_recs=64):
        """include/eleven_hip_debug.h: one more sample of pixel idx, one ErTraceRec per bounce-loop iteration."""
        recs = (abi.ErTraceRec * max_recs)()
        n = C.c_int()
        abi.check(self.lib.er_debug_trace_pixel(self.handle, idx, recs, max_recs, C.byref(n)))
        return [recs[i] for i in range(n.value)]

    def state_export(self):
        """er_state_export: the whole progressive state (planes + sample counts + RNG) as bytes."""
        n = C.c_uint64()
        abi.check(self.lib.er_state_size(self.handle, C.byref(n)))
        buf = np.empty(n.value, np.uint8)
        abi.check(self.lib.er_state_export(self.handle, buf.ctypes.data_as(C.c_void_p), n.value))
        return buf

    def state_import(self, buf):
        b = np.ascontiguousarray(buf, np.uint8)
        abi.check(self.lib.er_state_import(self.handle, b.ctypes.data_as(C.c_void_p), b.size))

    def owned_count(self, rank):
        v = C.c_uint64()
        abi.check(self.lib.er_owned_count(self.handle, rank, C.byref(v)))
        return int(v.value)

    def pack_owned(self, pass_id, dev_ptr):
        abi.check(self.lib.er_pack_owned(self.handle, pass_id, C.c_void_p(dev_ptr)))

    def unpack_owned(self, pass_id, src_rank, dev_ptr):
        abi.check(self.lib.er_unpack_owned(self.handle, pass_id, src_rank, C.c_void_p(dev_ptr)))

    def close(self):
        if self.handle:
            self.lib.er_scene_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
