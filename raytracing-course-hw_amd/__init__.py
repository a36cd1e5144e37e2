"""ctypes binding of librtamd.so — the C-ABI in include/rtamd.h.

This Python layer is plumbing for tests and bench.py only: it marshals numpy arrays into
``rt_scene_desc`` and calls the library.  There is no Python or CPU implementation of the render
path here; if the HIP extension has not been built the import fails loudly.

The package directory name contains a hyphen, so import it with
``importlib.import_module("raytracing-course-hw_amd")``.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RTAMD_LIB") or os.path.join(_HERE, "librtamd.so")  # RTAMD_LIB: an experimental build of the same library (tools/tuning)

RT_INTEGRATOR_HW1, RT_INTEGRATOR_HW2, RT_INTEGRATOR_HW3, RT_INTEGRATOR_HW4, RT_INTEGRATOR_HW5 = 1, 2, 3, 4, 5
RT_INTEGRATOR_HW6, RT_INTEGRATOR_HW7, RT_INTEGRATOR_HW8 = 6, 7, 8
RT_FLAG_OUT_DEVICE, RT_FLAG_COUNTERS, RT_FLAG_SAMPLE_SEEDS, RT_FLAG_RUSSIAN_ROULETTE = 1, 2, 4, 8
RT_BUILD_DEVICE_BVH = 1
RT_PIPELINE_SINGLE, RT_PIPELINE_ROUNDS, RT_PIPELINE_PERSISTENT = 0, 1, 2
RT_OK = 0
RT_ERR_NO_DEVICE = -2


class RtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"rtamd error {code}: {msg}")
        self.code = code


class rt_material(C.Structure):
    _fields_ = [("base_color", C.c_float * 3), ("emission", C.c_float * 3), ("metallic_factor", C.c_float),
                ("roughness_factor", C.c_float), ("base_color_texture", C.c_int32), ("emissive_texture", C.c_int32),
                ("metallic_roughness_texture", C.c_int32), ("normal_texture", C.c_int32), ("kind", C.c_int32),
                ("ior", C.c_float)]


class rt_image(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("rgb", C.POINTER(C.c_uint8))]


class rt_primitive(C.Structure):
    _fields_ = [("type", C.c_int32), ("data", C.c_float * 3), ("position", C.c_float * 3), ("rotation", C.c_float * 4),
                ("color", C.c_float * 3), ("emission", C.c_float * 3), ("kind", C.c_int32), ("ior", C.c_float),
                ("data2", C.c_float * 3), ("data3", C.c_float * 3)]


class rt_light(C.Structure):
    _fields_ = [("type", C.c_int32), ("intensity", C.c_float * 3), ("position", C.c_float * 3),
                ("attenuation", C.c_float * 3), ("direction", C.c_float * 3)]


class rt_camera(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("right", C.c_float * 3), ("up", C.c_float * 3), ("forward", C.c_float * 3),
                ("fov_y", C.c_float), ("fov_x", C.c_float)]


class rt_scene_desc(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("n_triangles", C.c_uint32),
                ("positions", C.POINTER(C.c_float)), ("texcoords", C.POINTER(C.c_float)),
                ("normals", C.POINTER(C.c_float)), ("tangents", C.POINTER(C.c_float)),
                ("material_index", C.POINTER(C.c_uint32)),
                ("n_materials", C.c_uint32), ("materials", C.POINTER(rt_material)),
                ("n_textures", C.c_uint32), ("texture_source", C.POINTER(C.c_uint32)),
                ("n_images", C.c_uint32), ("images", C.POINTER(rt_image)),
                ("environment_map", C.POINTER(rt_image)),
                ("n_primitives", C.c_uint32), ("primitives", C.POINTER(rt_primitive)),
                ("camera", rt_camera), ("bg_color", C.c_float * 3),
                ("n_lights", C.c_uint32), ("lights", C.POINTER(rt_light)), ("ambient_light", C.c_float * 3),
                ("build_flags", C.c_uint32), ("reserved", C.c_uint32)]


class rt_render_params(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("width", C.c_int32), ("height", C.c_int32), ("samples", C.c_int32),
                ("ray_depth", C.c_int32), ("integrator", C.c_int32), ("tile_w", C.c_int32), ("tile_h", C.c_int32),
                ("shard_index", C.c_int32), ("shard_count", C.c_int32), ("flags", C.c_uint32), ("stream", C.c_void_p),
                ("sample_streams", C.c_int32), ("reserved", C.c_int32)]


class rt_stats(C.Structure):
    _fields_ = [("kernel_ms", C.c_double), ("total_ms", C.c_double), ("samples", C.c_uint64),
                ("closest_hit_queries", C.c_uint64), ("light_pdf_queries", C.c_uint64), ("node_visits", C.c_uint64),
                ("triangle_tests", C.c_uint64), ("launches", C.c_uint32), ("dominant_kernel_launches", C.c_uint32),
                ("dominant_kernel_ms", C.c_double), ("pipeline", C.c_uint32), ("reference_exact", C.c_uint32),
                ("exact_closest_hits", C.c_uint64), ("exact_light_sums", C.c_uint64)]


class rt_scene_info(C.Structure):
    _fields_ = [("n_triangles", C.c_uint32), ("n_lights", C.c_uint32), ("n_bvh_nodes", C.c_uint32),
                ("n_light_bvh_nodes", C.c_uint32), ("bvh_depth", C.c_uint32), ("light_bvh_depth", C.c_uint32),
                ("device_bytes", C.c_uint64), ("prep_ms", C.c_double), ("upload_ms", C.c_double),
                ("bvh_build_ms", C.c_double), ("bvh_on_device", C.c_uint32), ("reserved", C.c_uint32)]


# Every symbol include/rtamd.h declares; tests/test_abi.py checks the library exports them all.
ABI_SYMBOLS = ["rt_abi_version", "rt_last_error", "rt_scene_create", "rt_scene_destroy", "rt_output_elems", "rt_render",
               "rt_unshard", "rt_scene_get_info", "rt_scene_get_light_order", "rt_load_gltf", "rt_load_txt",
               "rt_host_scene_set_environment", "rt_host_scene_desc", "rt_host_scene_free", "rt_write_ppm",
               "rt_decode_png", "rt_free", "rt_host_prepare_orders", "rt_device_count", "rt_multi_create", "rt_multi_render",
               "rt_multi_destroy"]

if not os.path.exists(LIB_PATH):
    raise ImportError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                      "(hipcc --offload-arch=gfx950); there is no fallback implementation")
lib = C.CDLL(LIB_PATH)
lib.rt_last_error.restype = C.c_char_p
lib.rt_scene_create.argtypes = [C.POINTER(rt_scene_desc), C.POINTER(C.c_void_p)]
lib.rt_scene_destroy.argtypes = [C.c_void_p]
lib.rt_scene_destroy.restype = None
lib.rt_output_elems.argtypes = [C.POINTER(rt_render_params)]
lib.rt_output_elems.restype = C.c_size_t
lib.rt_render.argtypes = [C.c_void_p, C.POINTER(rt_render_params), C.c_void_p, C.c_void_p, C.POINTER(rt_stats)]
lib.rt_unshard.argtypes = [C.POINTER(rt_render_params), C.c_void_p, C.c_size_t, C.c_void_p]
lib.rt_scene_get_info.argtypes = [C.c_void_p, C.POINTER(rt_scene_info)]
lib.rt_scene_get_light_order.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_uint32]
lib.rt_host_prepare_orders.argtypes = [C.POINTER(rt_scene_desc), C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
lib.rt_load_gltf.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]
lib.rt_load_txt.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_void_p)] + [C.POINTER(C.c_int32)] * 4
lib.rt_host_scene_set_environment.argtypes = [C.c_void_p, C.c_char_p]
lib.rt_host_scene_desc.argtypes = [C.c_void_p]
lib.rt_host_scene_desc.restype = C.POINTER(rt_scene_desc)
lib.rt_host_scene_free.argtypes = [C.c_void_p]
lib.rt_host_scene_free.restype = None
lib.rt_write_ppm.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.c_void_p]
lib.rt_decode_png.argtypes = [C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.POINTER(C.c_uint8))]
lib.rt_free.argtypes = [C.c_void_p]
lib.rt_free.restype = None
lib.rt_multi_create.argtypes = [C.POINTER(rt_scene_desc), C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]
lib.rt_multi_render.argtypes = [C.c_void_p, C.POINTER(rt_render_params), C.c_void_p, C.c_void_p, C.POINTER(rt_stats)]
lib.rt_multi_destroy.argtypes = [C.c_void_p]
lib.rt_multi_destroy.restype = None


def _check(code):
    if code < 0:
        raise RtError(code, lib.rt_last_error().decode("utf-8", "replace"))
    return code


def _fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None and a.size else None


class SceneData:
    """Scene arrays in LOAD order as numpy (owns its memory) + an rt_scene_desc pointing at them."""

    def __init__(self, positions, texcoords, normals, tangents, material_index, materials, texture_source=(), images=(),
                 camera=None, bg=(0, 0, 0), environment=None, primitives=None, lights=None, ambient=(0, 0, 0)):
        f32 = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float32)
        self.positions = f32(positions if positions is not None else np.zeros(0, np.float32)).reshape(-1, 9)
        n = self.positions.shape[0]
        self.texcoords = None if texcoords is None else f32(texcoords).reshape(n, 6)
        self.normals = None if normals is None else f32(normals).reshape(n, 9)
        self.tangents = None if tangents is None else f32(tangents).reshape(n, 12)
        self.material_index = np.ascontiguousarray(material_index, dtype=np.uint32).reshape(n)
        self.materials = (rt_material * max(1, len(materials)))(*materials)
        self.n_materials = len(materials)
        self.texture_source = np.ascontiguousarray(texture_source, dtype=np.uint32)
        self.images = [np.ascontiguousarray(im, dtype=np.uint8) for im in images]  # each (h, w, 3)
        self.camera = camera if camera is not None else rt_camera()
        self.bg = tuple(float(x) for x in bg)
        self.environment = None if environment is None else np.ascontiguousarray(environment, dtype=np.uint8)
        self.primitives = primitives
        self.lights = lights
        self.ambient = tuple(float(x) for x in ambient)
        self._build_desc()

    def _build_desc(self):
        d = rt_scene_desc()
        d.struct_size = C.sizeof(rt_scene_desc)
        d.n_triangles = self.positions.shape[0]
        d.positions = _fptr(self.positions)
        d.texcoords = _fptr(self.texcoords)
        d.normals = _fptr(self.normals)
        d.tangents = _fptr(self.tangents)
        d.material_index = self.material_index.ctypes.data_as(C.POINTER(C.c_uint32))
        d.n_materials = self.n_materials
        d.materials = self.materials
        d.n_textures = self.texture_source.size
        d.texture_source = self.texture_source.ctypes.data_as(C.POINTER(C.c_uint32))
        self._images = (rt_image * max(1, len(self.images)))()
        for i, im in enumerate(self.images):
            self._images[i] = rt_image(im.shape[1], im.shape[0], im.ctypes.data_as(C.POINTER(C.c_uint8)))
        d.n_images = len(self.images)
        d.images = self._images
        if self.environment is not None:
            self._env = rt_image(self.environment.shape[1], self.environment.shape[0],
                                 self.environment.ctypes.data_as(C.POINTER(C.c_uint8)))
            d.environment_map = C.pointer(self._env)
        if self.primitives is not None and len(self.primitives):
            self._prims = (rt_primitive * len(self.primitives))(*self.primitives)
            d.n_primitives = len(self.primitives)
            d.primitives = self._prims
        d.camera = self.camera
        d.bg_color = (C.c_float * 3)(*self.bg)
        if self.lights is not None and len(self.lights):
            self._lights = (rt_light * len(self.lights))(*self.lights)
            d.n_lights = len(self.lights)
            d.lights = self._lights
        d.ambient_light = (C.c_float * 3)(*self.ambient)
        self.desc = d

    @staticmethod
    def from_desc(desc):
        """Deep-copy an rt_scene_desc (e.g. the loader's) into numpy-owned memory."""
        n = desc.n_triangles
        arr = lambda p, k: None if not p else np.ctypeslib.as_array(p, shape=(n * k,)).copy()
        mats = [rt_material.from_buffer_copy(desc.materials[i]) for i in range(desc.n_materials)]
        images = []
        for i in range(desc.n_images):
            im = desc.images[i]
            images.append(np.ctypeslib.as_array(im.rgb, shape=(im.height, im.width, 3)).copy())
        env = None
        if desc.environment_map:
            im = desc.environment_map.contents
            env = np.ctypeslib.as_array(im.rgb, shape=(im.height, im.width, 3)).copy()
        tsrc = np.ctypeslib.as_array(desc.texture_source, shape=(desc.n_textures,)).copy() if desc.n_textures else ()
        prims = [rt_primitive.from_buffer_copy(desc.primitives[i]) for i in range(desc.n_primitives)]
        lights = [rt_light.from_buffer_copy(desc.lights[i]) for i in range(desc.n_lights)]
        return SceneData(arr(desc.positions, 9), arr(desc.texcoords, 6), arr(desc.normals, 9), arr(desc.tangents, 12),
                         np.ctypeslib.as_array(desc.material_index, shape=(n,)).copy() if n else np.zeros(0, np.uint32),
                         mats, tsrc, images, rt_camera.from_buffer_copy(desc.camera), tuple(desc.bg_color), env, prims, lights,
                         tuple(desc.ambient_light))


def load_gltf(path, flavor=RT_INTEGRATOR_HW8, environment=None):
    """Load a glTF scene with the product's C++ loader and return it as SceneData."""
    hs = C.c_void_p()
    _check(lib.rt_load_gltf(os.fsencode(path), flavor, C.byref(hs)))
    try:
        if environment is not None:
            _check(lib.rt_host_scene_set_environment(hs, os.fsencode(environment)))
        return SceneData.from_desc(lib.rt_host_scene_desc(hs).contents)
    finally:
        lib.rt_host_scene_free(hs)


def load_txt(path, flavor=RT_INTEGRATOR_HW3):
    """Load a .txt scene (grammar of snapshot hw1..hw5 by flavor). Returns (SceneData, width, height, samples, ray_depth)."""
    hs = C.c_void_p()
    w, h, s, d = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
    _check(lib.rt_load_txt(os.fsencode(path), flavor, C.byref(hs), C.byref(w), C.byref(h), C.byref(s), C.byref(d)))
    try:
        return SceneData.from_desc(lib.rt_host_scene_desc(hs).contents), w.value, h.value, s.value, d.value
    finally:
        lib.rt_host_scene_free(hs)


def host_prepare_orders(data, integrator):
    """(figure order, light order) of the host-side scene preparation, without a GPU."""
    n = max(1, data.positions.shape[0], len(data.primitives or ()))
    fo, lo = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
    k = _check(lib.rt_host_prepare_orders(C.byref(data.desc), integrator, fo.ctypes.data, n, lo.ctypes.data, n))
    nf = data.positions.shape[0] if data.positions.shape[0] else len(data.primitives or ())
    return fo[:nf], lo[:k]


def decode_png(path):
    w, h, p = C.c_int32(), C.c_int32(), C.POINTER(C.c_uint8)()
    _check(lib.rt_decode_png(os.fsencode(path), C.byref(w), C.byref(h), C.byref(p)))
    try:
        return np.ctypeslib.as_array(p, shape=(h.value, w.value, 3)).copy()
    finally:
        lib.rt_free(p)


def write_ppm(path, rgb8):
    rgb8 = np.ascontiguousarray(rgb8, dtype=np.uint8)
    _check(lib.rt_write_ppm(os.fsencode(path), rgb8.shape[1], rgb8.shape[0], rgb8.ctypes.data))


def dominant_kernel_name(st):
    """Name of the kernel rt_stats.dominant_kernel_ms / _launches refer to (for the hw8 / hw7 integrators)."""
    return {RT_PIPELINE_ROUNDS: "wf_traverse_kernel", RT_PIPELINE_PERSISTENT: "pt_persistent_kernel"}.get(st.pipeline, "render_hw8_kernel")


def make_params(width, height, samples, integrator=RT_INTEGRATOR_HW8, ray_depth=0, shard_index=0, shard_count=1,
                tile=32, flags=0, stream=None, sample_streams=0):
    p = rt_render_params()
    p.struct_size = C.sizeof(rt_render_params)
    p.width, p.height, p.samples, p.ray_depth, p.integrator = width, height, samples, ray_depth, integrator
    p.tile_w = p.tile_h = tile
    p.shard_index, p.shard_count, p.flags = shard_index, shard_count, flags
    p.stream = stream
    p.sample_streams = sample_streams  # 0/1 = replay mode (reference pixels); K > 1 = throughput mode, statistical parity only
    return p


def unshard(params, buf):
    buf = np.ascontiguousarray(buf)
    full = np.zeros((params.height, params.width, 3), dtype=buf.dtype)
    _check(lib.rt_unshard(C.byref(params), buf.ctypes.data, buf.dtype.itemsize, full.ctypes.data))
    return full


class MultiScene:
    """One scene replica per listed HIP device (rt_multi); render() shards the frame over them and returns the whole frame."""

    def __init__(self, data, devices):
        self.data = data
        self._h = C.c_void_p()
        arr = (C.c_int * len(devices))(*devices)
        _check(lib.rt_multi_create(C.byref(data.desc), arr, len(devices), C.byref(self._h)))

    def close(self):
        if self._h:
            lib.rt_multi_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def render(self, width, height, samples, want_float=True, want_rgb8=True, **kw):
        p = make_params(width, height, samples, **kw)
        p.shard_count, p.shard_index = 0, 0
        rgb = np.zeros((height, width, 3), dtype=np.float32) if want_float else None
        rgb8 = np.zeros((height, width, 3), dtype=np.uint8) if want_rgb8 else None
        st = rt_stats()
        _check(lib.rt_multi_render(self._h, C.byref(p), rgb.ctypes.data if want_float else None,
                                   rgb8.ctypes.data if want_rgb8 else None, C.byref(st)))
        return rgb, rgb8, st


class Scene:
    """A prepared scene resident in HBM on the current HIP device (rt_scene)."""

    def __init__(self, data, build_flags=0):
        self.data = data
        self._h = C.c_void_p()
        desc = rt_scene_desc.from_buffer_copy(data.desc)   # the arrays stay owned by `data`
        desc.build_flags = build_flags
        _check(lib.rt_scene_create(C.byref(desc), C.byref(self._h)))

    def close(self):
        if self._h:
            lib.rt_scene_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self):
        i = rt_scene_info()
        _check(lib.rt_scene_get_info(self._h, C.byref(i)))
        return i

    def light_order(self):
        n = self.info().n_lights
        out = np.zeros(max(n, 1), dtype=np.uint32)
        _check(lib.rt_scene_get_light_order(self._h, out.ctypes.data_as(C.POINTER(C.c_uint32)), out.size))
        return out[:n]

    def render(self, width, height, samples, want_float=True, want_rgb8=True, counters=False, flags=0, **kw):
        """Render to host arrays. Returns (rgb float32 (H,W,3) or shard buffer, rgb8, rt_stats)."""
        p = make_params(width, height, samples, flags=(RT_FLAG_COUNTERS if counters else 0) | flags, **kw)
        n = lib.rt_output_elems(C.byref(p))
        if n == 0:
            raise RtError(-1, "bad render parameters")
        rgb = np.zeros(n, dtype=np.float32) if want_float else None
        rgb8 = np.zeros(n, dtype=np.uint8) if want_rgb8 else None
        st = rt_stats()
        _check(lib.rt_render(self._h, C.byref(p), rgb.ctypes.data if want_float else None,
                             rgb8.ctypes.data if want_rgb8 else None, C.byref(st)))
        if p.shard_count <= 1:
            rgb = None if rgb is None else rgb.reshape(height, width, 3)
            rgb8 = None if rgb8 is None else rgb8.reshape(height, width, 3)
        return rgb, rgb8, st

    def render_device(self, params, out_rgb_ptr, out_rgb8_ptr):
        """Render into device buffers (raw pointers, e.g. torch tensor data_ptr())."""
        st = rt_stats()
        params.flags |= RT_FLAG_OUT_DEVICE
        _check(lib.rt_render(self._h, C.byref(params), out_rgb_ptr, out_rgb8_ptr, C.byref(st)))
        return st
