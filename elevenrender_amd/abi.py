"""ctypes mirror of include/eleven_hip.h and loader of libeleven_hip.so.

The library is the product; this module only declares its C ABI for Python callers
(tests, bench.py, the Python host mirror in render.py).  There is no fallback: if the
shared library is missing, `load()` raises -- build it with `python -c "import
__graft_entry__ as g; g.build()"` or `make -C elevenrender_amd/csrc`.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# ELEVEN_HIP_LIB: load another build of the same library (tools/sanitize_cpu.sh points it at the ASan + UBSan build)
LIB_PATH = os.environ.get("ELEVEN_HIP_LIB") or os.path.join(_HERE, "libeleven_hip.so")

ER_OK = 0
ER_ERR_INVALID_ARG, ER_ERR_NO_DEVICE, ER_ERR_HIP, ER_ERR_STATE, ER_ERR_OOM = -1, -2, -3, -4, -5
PASS_BEAUTY, PASS_DENOISE, PASS_NORMAL, PASS_TANGENT, PASS_BITANGENT, PASS_COUNT = 0, 1, 2, 3, 4, 5
PASS_NAMES = {"beauty": 0, "denoise": 1, "normal": 2, "tangent": 3, "bitangent": 4}
FLAG_POINT_LIGHTS, FLAG_COUNTERS, FLAG_MEGAKERNEL, FLAG_PROFILE, FLAG_FUSED, FLAG_WAVEFRONT, FLAG_GPU_BUILD, FLAG_MIS, FLAG_STREAM = 1, 2, 4, 8, 16, 32, 64, 128, 256
FLAG_HOST_BUILD = 512


class ErVec3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class ErCamera(C.Structure):
    _fields_ = [("focal_length", C.c_float), ("sensor_width", C.c_float), ("sensor_height", C.c_float),
                ("aperture", C.c_float), ("focus_distance", C.c_float), ("rotation", ErVec3),
                ("bokeh", C.c_int32), ("position", ErVec3)]


class ErMaterial(C.Structure):
    _fields_ = [("albedo_tex", C.c_int32), ("emission_tex", C.c_int32), ("roughness_tex", C.c_int32),
                ("metallic_tex", C.c_int32), ("normal_tex", C.c_int32), ("opacity_tex", C.c_int32),
                ("transmission_tex", C.c_int32), ("albedo_shader_id", C.c_int32),
                ("albedo", ErVec3), ("emission", ErVec3),
                ("opacity", C.c_float), ("roughness", C.c_float), ("metallic", C.c_float),
                ("clearcoat_gloss", C.c_float), ("clearcoat", C.c_float), ("anisotropic", C.c_float),
                ("eta", C.c_float), ("transmission", C.c_float), ("specular", C.c_float),
                ("specular_tint", C.c_float), ("sheen_tint", C.c_float), ("subsurface", C.c_float),
                ("sheen", C.c_float), ("ax", C.c_float), ("ay", C.c_float)]


class ErTexture(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("channels", C.c_int32), ("filter", C.c_int32),
                ("data", C.POINTER(C.c_float))]


class ErHdri(C.Structure):
    _fields_ = [("texture", ErTexture), ("cdf", C.POINTER(C.c_float)), ("radiance_sum", C.c_float)]


class ErPointLight(C.Structure):
    _fields_ = [("position", ErVec3), ("radiance", ErVec3)]


class ErSceneDesc(C.Structure):
    _fields_ = [("tri_count", C.c_uint32),
                ("vertices", C.POINTER(C.c_float)), ("normals", C.POINTER(C.c_float)),
                ("tangents", C.POINTER(C.c_float)), ("uvs", C.POINTER(C.c_float)),
                ("tangent_sign", C.POINTER(C.c_float)), ("material_id", C.POINTER(C.c_int32)),
                ("material_count", C.c_uint32), ("materials", C.POINTER(ErMaterial)),
                ("texture_count", C.c_uint32), ("textures", C.POINTER(ErTexture)),
                ("hdri", ErHdri), ("camera", ErCamera),
                ("point_light_count", C.c_uint32), ("point_lights", C.POINTER(ErPointLight)),
                ("x_res", C.c_uint32), ("y_res", C.c_uint32)]


class ErRenderParams(C.Structure):
    _fields_ = [("sample_target", C.c_uint32), ("block_size", C.c_uint32), ("max_bounces", C.c_uint32),
                ("device", C.c_int32), ("rank", C.c_uint32), ("world", C.c_uint32), ("flags", C.c_uint32)]


class ErDeviceInfo(C.Structure):
    _fields_ = [("name", C.c_char * 256), ("platform", C.c_char * 64), ("memory_bytes", C.c_uint64),
                ("compute_units", C.c_uint32), ("compatible", C.c_int32), ("arch", C.c_char * 64)]


class ErCounters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("paths", "bounce_samples", "rays", "node_visits", "tri_tests",
                                          "shaded_hits", "texel_fetches", "hdri_samples", "trace_wave_steps", "trace_busy_lanes",
                                          "trace_node_lanes", "trace_tri_lanes")]


class ErProfile(C.Structure):
    _fields_ = [("trace_ms", C.c_float), ("shade_ms", C.c_float), ("trace_launches", C.c_uint32), ("shade_launches", C.c_uint32), ("schedule", C.c_uint32),
                ("concurrency", C.c_uint32), ("empty_launches", C.c_uint32), ("empty_ms", C.c_float), ("rays_logged", C.c_uint64)]


class ErAccelInfo(C.Structure):
    _fields_ = [("node_count", C.c_uint32), ("node_bytes", C.c_uint32), ("leaf_count", C.c_uint32),
                ("max_depth", C.c_uint32), ("tri_record_bytes", C.c_uint32), ("build_ms", C.c_float),
                ("upload_ms", C.c_float), ("lift_bound", C.c_float), ("builder", C.c_uint32)]


class ErStreamInfo(C.Structure):   # include/eleven_hip_debug.h
    _fields_ = [("waves", C.c_uint32), ("tracers", C.c_uint32), ("large_regions", C.c_uint32), ("deal_pending", C.c_uint32), ("launches", C.c_uint32),
                ("pixels_per_cu", C.c_uint32), ("lanes_busy", C.c_double), ("launch_ms", C.c_double), ("cost_spread", C.c_double),
                ("spec_started", C.c_uint64), ("spec_right", C.c_uint64), ("spec_wrong", C.c_uint64),
                ("form", C.c_uint32), ("reserved", C.c_uint32)]


class ErTraceRec(C.Structure):   # include/eleven_hip_debug.h; same layout as the oracle's OracleTraceRec
    _fields_ = [("bounce", C.c_int32), ("tri", C.c_int32), ("shadow_tri", C.c_int32), ("opaque", C.c_int32),
                ("position", C.c_float * 3), ("wi", C.c_float * 3), ("light", C.c_float * 3), ("reduction", C.c_float * 3),
                ("shadow_occ", C.c_int32), ("light_occ", C.c_int32)]


# every symbol include/eleven_hip.h declares: name -> (restype, argtypes)
_P = C.c_void_p
_FP, _IP = C.POINTER(C.c_float), C.POINTER(C.c_int32)
SYMBOLS = {
    "er_abi_version": (C.c_int, []),
    "er_last_error": (C.c_char_p, []),
    "er_device_count": (C.c_int, []),
    "er_device_info": (C.c_int, [C.c_int, C.POINTER(ErDeviceInfo)]),
    "er_device_find": (C.c_int, [C.c_char_p]),
    "er_scene_create": (C.c_int, [C.POINTER(ErSceneDesc), C.POINTER(_P)]),
    "er_scene_destroy": (None, [_P]),
    "er_render_begin": (C.c_int, [_P, C.POINTER(ErRenderParams)]),
    "er_render_samples": (C.c_int, [_P, C.c_uint32]),
    "er_render_samples_async": (C.c_int, [_P, C.c_uint32]),
    "er_wait": (C.c_int, [_P, C.POINTER(C.c_float)]),
    "er_samples_done": (C.c_int, [_P, C.POINTER(C.c_uint32)]),
    "er_read_pass": (C.c_int, [_P, C.c_int, C.POINTER(C.c_float)]),
    "er_read_samples": (C.c_int, [_P, C.POINTER(C.c_uint32)]),
    "er_read_rng": (C.c_int, [_P, C.POINTER(C.c_uint32)]),
    "er_owned_count": (C.c_int, [_P, C.c_uint32, C.POINTER(C.c_uint64)]),
    "er_pack_owned": (C.c_int, [_P, C.c_int, _P]),
    "er_unpack_owned": (C.c_int, [_P, C.c_int, C.c_uint32, _P]),
    "er_get_counters": (C.c_int, [_P, C.POINTER(ErCounters)]),
    "er_accel_info": (C.c_int, [_P, C.POINTER(ErAccelInfo)]),
    "er_get_profile": (C.c_int, [_P, C.POINTER(ErProfile)]),
    "er_denoise": (C.c_int, [_P, C.c_uint32, C.c_float]),
    "er_state_size": (C.c_int, [_P, C.POINTER(C.c_uint64)]),
    "er_state_export": (C.c_int, [_P, _P, C.c_uint64]),
    "er_state_import": (C.c_int, [_P, _P, C.c_uint64]),
    "er_comm_unique_id": (C.c_int, [C.POINTER(C.c_uint8)]),
    "er_comm_create": (C.c_int, [C.POINTER(C.c_uint8), C.c_uint32, C.c_uint32, C.c_int, C.POINTER(_P)]),
    "er_comm_destroy": (None, [_P]),
    "er_gather_pass": (C.c_int, [_P, C.c_int, _P, C.c_uint32]),
    "er_debug_comm_create_local": (C.c_int, [C.c_uint32, C.POINTER(_P)]),
    "er_comm_create_local": (C.c_int, [C.c_uint32, C.POINTER(_P)]),
    "er_debug_stream_info": (C.c_int, [_P, C.POINTER(ErStreamInfo)]),
    "er_debug_gather_buffers": (C.c_int, [_P, C.c_uint32, C.POINTER(_P), C.POINTER(C.c_uint64), C.POINTER(_P), C.POINTER(C.c_uint64)]),
    "er_debug_comm_loopback": (C.c_int, [_P, C.c_uint64, C.POINTER(C.c_double)]),
    "er_measure_hbm_peak": (C.c_int, [C.c_int, C.c_uint64, C.c_uint32, _FP, _FP]),
    "er_debug_eval": (C.c_int, [_P, C.c_int, _FP, C.c_uint32, C.c_uint32, _FP, C.c_uint32]),
    "er_debug_trace_rays": (C.c_int, [_P, _FP, _FP, C.c_uint32, _IP, _FP, _IP, _IP, _FP, _FP, _IP]),
    "er_debug_trace_pixel": (C.c_int, [_P, C.c_uint32, C.POINTER(ErTraceRec), C.c_int, C.POINTER(C.c_int)]),
    "er_debug_set_host_alloc_limit": (None, [C.c_uint64]),
    "er_debug_set_gpu_build_failure": (None, [C.c_int]),
    "er_debug_closest_hit": (C.c_int, [_P, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_uint32, C.POINTER(C.c_int32), C.POINTER(C.c_float),
                                       C.POINTER(C.c_float)]),
}

_lib = None


class ErError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"eleven_hip error {code}: {msg}")
        self.code = code


ABI_VERSION = 2     # include/eleven_hip.h ER_ABI_VERSION


def load():
    """Load libeleven_hip.so and declare its prototypes.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ErError(ER_ERR_NO_DEVICE, f"{LIB_PATH} is missing: the HIP extension is not built "
                      "(run __graft_entry__.build()); there is no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.er_abi_version() != ABI_VERSION:      # the structs below are filled completely by the library: a layout mismatch overwrites memory
        raise ErError(ER_ERR_STATE, f"{LIB_PATH} has ABI version {lib.er_abi_version()}, this binding was written for {ABI_VERSION}: rebuild")
    _lib = lib
    return lib


def check(rc):
    if rc != ER_OK:
        raise ErError(rc, load().er_last_error().decode("utf-8", "replace"))


def _fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _f32(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if shape is not None:
        a = a.reshape(shape)
    return a


class SceneData:
    """A scene as flat numpy arrays (the layout of ErSceneDesc) + the ctypes descriptor.

    Keeps every array alive for as long as the descriptor is in use.  Used for BOTH the HIP
    library and (by tests/bench) the CPU oracle, so the two see bit-identical inputs.
    """

    def __init__(self, vertices, normals, tangents, uvs, tangent_sign, material_id, materials,
                 textures=(), hdri=None, hdri_cdf=None, hdri_radiance_sum=0.0, camera=None,
                 x_res=64, y_res=64, point_lights=()):
        n = 0 if vertices is None else int(np.asarray(vertices).size // 9)
        self.tri_count = n
        self.vertices = _f32(vertices if n else np.zeros((0, 3, 3)), (n, 3, 3))
        self.normals = _f32(normals if n else np.zeros((0, 3, 3)), (n, 3, 3))
        self.tangents = _f32(tangents if n else np.zeros((0, 3, 3)), (n, 3, 3))
        self.uvs = _f32(uvs if n else np.zeros((0, 3, 2)), (n, 3, 2))
        self.tangent_sign = _f32(tangent_sign if n else np.zeros((0,)), (n,))
        self.material_id = np.ascontiguousarray(material_id if n else np.zeros((0,)), dtype=np.int32).reshape(n)
        self.materials = list(materials)
        self.textures = [(_f32(d), int(w), int(h), int(ch), int(flt)) for (d, w, h, ch, flt) in textures]
        if hdri is None:   # HDRI() default: 1x1 texel (0.5,0.5,0.5), reference src/HDRI.cpp:18
            hdri = (np.full((1, 1, 3), 0.5, np.float32), 1, 1, 3, 0)
        d, w, h, ch, flt = hdri
        self.hdri = (_f32(d), int(w), int(h), int(ch), int(flt))
        self.hdri_cdf = None if hdri_cdf is None else _f32(hdri_cdf)
        self.hdri_radiance_sum = float(hdri_radiance_sum)
        self.camera = camera if camera is not None else default_camera()
        self.x_res, self.y_res = int(x_res), int(y_res)
        self.point_lights = list(point_lights)
        self._desc = None

    def desc(self):
        if self._desc is not None:
            return self._desc
        d = ErSceneDesc()
        d.tri_count = self.tri_count
        d.vertices, d.normals, d.tangents = _fptr(self.vertices), _fptr(self.normals), _fptr(self.tangents)
        d.uvs, d.tangent_sign = _fptr(self.uvs), _fptr(self.tangent_sign)
        d.material_id = self.material_id.ctypes.data_as(C.POINTER(C.c_int32))
        self._mats = (ErMaterial * len(self.materials))(*self.materials)
        d.material_count, d.materials = len(self.materials), self._mats
        self._texs = (ErTexture * max(1, len(self.textures)))()
        for i, (data, w, h, ch, flt) in enumerate(self.textures):
            self._texs[i] = ErTexture(w, h, ch, flt, _fptr(data))
        d.texture_count, d.textures = len(self.textures), self._texs
        data, w, h, ch, flt = self.hdri
        d.hdri.texture = ErTexture(w, h, ch, flt, _fptr(data))
        if self.hdri_cdf is not None:
            d.hdri.cdf = _fptr(self.hdri_cdf)
            d.hdri.radiance_sum = self.hdri_radiance_sum
        d.camera = self.camera
        self._pls = (ErPointLight * max(1, len(self.point_lights)))(*self.point_lights)
        d.point_light_count, d.point_lights = len(self.point_lights), self._pls
        d.x_res, d.y_res = self.x_res, self.y_res
        self._desc = d
        return d


def default_camera():
    """reference Camera defaults, src/Camera.h:9-19."""
    c = ErCamera()
    c.focal_length = 35 * 0.001     # evaluated in double, then narrowed, as the C++ initialisers are
    c.sensor_width = 36 * 0.001
    c.sensor_height = 24 * 0.001
    c.aperture = 2.8
    c.focus_distance = 1000000.0
    c.rotation = ErVec3(0, 0, 0)
    c.bokeh = 0
    c.position = ErVec3(0, 0, 0)
    return c


def default_material(**kw):
    """reference Material defaults, src/Material.h:20-47."""
    m = ErMaterial()
    for f in ("albedo_tex", "emission_tex", "roughness_tex", "metallic_tex", "normal_tex", "opacity_tex",
              "transmission_tex", "albedo_shader_id"):
        setattr(m, f, -1)
    m.albedo = ErVec3(0.5, 0.5, 0.5)
    m.emission = ErVec3(0, 0, 0)
    m.opacity, m.roughness, m.metallic = 1, 1, 0
    m.clearcoat_gloss = m.clearcoat = m.anisotropic = m.eta = m.transmission = 0
    m.specular, m.specular_tint, m.sheen_tint = 0.5, 0, 0.5
    m.subsurface = m.sheen = m.ax = m.ay = 0
    for k, v in kw.items():
        if isinstance(v, (tuple, list)):
            v = ErVec3(*v)
        setattr(m, k, v)
    return m
