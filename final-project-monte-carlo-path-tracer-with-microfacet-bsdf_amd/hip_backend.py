"""ctypes binding to libmcpt_hip.so (the C ABI of include/mcpt.h) and a Python mirror of the reference's
`Renderer` seam: `HipScene` plays Scene (after Add/buildBVH), `HipScene.render` plays Renderer::Render
(reference src/Renderer.hpp:16-22, src/Scene.hpp:104-131).

There is no CPU fallback here: if the HIP library is missing or no GPU is usable, every entry point raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MCPT_LIB") or os.path.join(HERE, "libmcpt_hip.so")  # MCPT_LIB: diagnostic builds only
CHECK_LIB_PATH = os.path.join(HERE, "libmcpt_hip_check.so")  # the checking build (build.build_check); tests only

EXPORTS = ["mcpt_scene_create", "mcpt_scene_destroy", "mcpt_render", "mcpt_render_device", "mcpt_intersect",
           "mcpt_cast_rays", "mcpt_camera_rays", "mcpt_scene_get_info", "mcpt_bvh_dump", "mcpt_scene_create_ex", "mcpt_scene_dump_bvh", "mcpt_tonemap", "mcpt_tonemap_device", "mcpt_debug_fmath", "mcpt_debug_material", "mcpt_debug_scene", "mcpt_debug_counters",
           "mcpt_group_create", "mcpt_group_render", "mcpt_group_size", "mcpt_group_get_info", "mcpt_group_scene", "mcpt_group_destroy", "mcpt_group_last_error",
           "mcpt_last_error", "mcpt_version"]


class McptError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("mcpt error %d: %s" % (code, msg))
        self.code = code


class SceneDesc(C.Structure):
    _fields_ = [("n_objects", C.c_int32), ("n_triangles", C.c_int32), ("n_materials", C.c_int32),
                ("env_w", C.c_int32), ("env_h", C.c_int32), ("background", C.c_float * 3),
                ("objects", C.c_void_p), ("triangles", C.c_void_p), ("materials", C.c_void_p), ("env_pixels", C.c_void_p)]


class Params(C.Structure):
    _fields_ = [("spp", C.c_int32), ("spp_total", C.c_int32), ("sample_offset", C.c_int32), ("rr_rate", C.c_float),
                ("n_dir_sample", C.c_int32), ("enable_shadow", C.c_int32), ("seed", C.c_uint32), ("accumulate", C.c_int32),
                ("tile_size", C.c_int32), ("rank", C.c_int32), ("nranks", C.c_int32), ("spp_per_pass", C.c_int32),
                ("pool_paths", C.c_int32), ("max_depth", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("paths", C.c_uint64), ("vertices", C.c_uint64), ("shaded", C.c_uint64),
                ("closest_rays", C.c_uint64), ("shadow_rays", C.c_uint64), ("ref_scene_rays", C.c_uint64),
                ("iterations", C.c_uint64), ("overflow_paths", C.c_uint64), ("ms_total", C.c_double),
                ("ms_trace_closest", C.c_double), ("ms_trace_shadow", C.c_double), ("ms_shade", C.c_double),
                ("ms_generate", C.c_double), ("ms_resolve", C.c_double),
                ("n_trace_closest", C.c_uint64), ("n_trace_shadow", C.c_uint64), ("n_shade", C.c_uint64),
                ("n_generate", C.c_uint64), ("n_resolve", C.c_uint64), ("ms_direct", C.c_double), ("n_direct", C.c_uint64), ("direct_vertices", C.c_uint64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class BuildOptions(C.Structure):
    _fields_ = [("builder", C.c_int32), ("quantise", C.c_int32), ("instancing", C.c_int32), ("reserved", C.c_int32 * 5)]


BUILDERS = {"default": 0, "sah": 1, "reference": 2, "lbvh": 3, "ploc": 4}  # MCPT_BUILD_*


class SceneInfo(C.Structure):
    _fields_ = [("n_nodes", C.c_int32), ("bvh_height", C.c_int32), ("n_lights", C.c_int32), ("n_prims", C.c_int32),
                ("scene_bytes", C.c_uint64), ("build_ms", C.c_double), ("upload_ms", C.c_double), ("builder", C.c_int32),
                ("quantised", C.c_int32), ("n_instances", C.c_int32), ("lds_resident", C.c_int32), ("init_ms", C.c_double)]


class GroupInfo(C.Structure):
    _fields_ = [("n_devices", C.c_int32), ("uses_rccl", C.c_int32), ("build_ms", C.c_double), ("upload_ms_max", C.c_double),
                ("init_ms_max", C.c_double), ("setup_ms", C.c_double)]


_libs = {}


def lib(path=None):
    """Loads libmcpt_hip.so (or the library at `path`); raises if it has not been built (run __graft_entry__.build())."""
    path = path or LIB_PATH
    if path not in _libs:
        if not os.path.exists(path):
            raise FileNotFoundError("%s is missing: build it with `python __graft_entry__.py` (hipcc, gfx950)" % path)
        L = C.CDLL(path)
        L.mcpt_last_error.restype = C.c_char_p
        L.mcpt_version.restype = C.c_char_p
        L.mcpt_scene_create.restype = C.c_int
        L.mcpt_scene_create.argtypes = [C.POINTER(SceneDesc), C.c_int, C.POINTER(C.c_void_p)]
        L.mcpt_scene_create_ex.restype = C.c_int
        L.mcpt_scene_create_ex.argtypes = [C.POINTER(SceneDesc), C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
        L.mcpt_scene_dump_bvh.restype = C.c_int
        L.mcpt_scene_dump_bvh.argtypes = [C.c_void_p, C.POINTER(BvhInfo), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.mcpt_tonemap.restype = C.c_int
        L.mcpt_tonemap.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.mcpt_tonemap_device.restype = C.c_int
        L.mcpt_tonemap_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        L.mcpt_scene_destroy.restype = None
        L.mcpt_scene_destroy.argtypes = [C.c_void_p]
        L.mcpt_scene_get_info.restype = C.c_int
        L.mcpt_scene_get_info.argtypes = [C.c_void_p, C.POINTER(SceneInfo)]
        L.mcpt_bvh_dump.restype = C.c_int
        L.mcpt_bvh_dump.argtypes = [C.POINTER(SceneDesc), C.POINTER(BvhInfo), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.mcpt_render.restype = C.c_int
        L.mcpt_render.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Params), C.c_void_p, C.POINTER(Stats)]
        L.mcpt_render_device.restype = C.c_int
        L.mcpt_render_device.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_void_p, C.POINTER(Stats)]
        L.mcpt_intersect.restype = C.c_int
        L.mcpt_intersect.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.mcpt_cast_rays.restype = C.c_int
        L.mcpt_cast_rays.argtypes = [C.c_void_p, C.POINTER(Params), C.c_int64] + [C.c_void_p] * 6
        L.mcpt_camera_rays.restype = C.c_int
        L.mcpt_camera_rays.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int64] + [C.c_void_p] * 4
        L.mcpt_debug_fmath.restype = C.c_int
        L.mcpt_debug_fmath.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.mcpt_debug_scene.restype = C.c_int
        L.mcpt_debug_scene.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p]
        L.mcpt_debug_material.restype = C.c_int
        L.mcpt_debug_material.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.mcpt_group_create.restype = C.c_int
        L.mcpt_group_create.argtypes = [C.POINTER(SceneDesc), C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
        L.mcpt_group_render.restype = C.c_int
        L.mcpt_group_render.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Params), C.c_void_p, C.POINTER(Stats)]
        L.mcpt_group_get_info.restype = C.c_int
        L.mcpt_group_get_info.argtypes = [C.c_void_p, C.POINTER(GroupInfo)]
        L.mcpt_group_scene.restype = C.c_void_p
        L.mcpt_group_scene.argtypes = [C.c_void_p, C.c_int]
        L.mcpt_group_size.restype = C.c_int
        L.mcpt_group_size.argtypes = [C.c_void_p]
        L.mcpt_group_destroy.restype = None
        L.mcpt_group_destroy.argtypes = [C.c_void_p]
        L.mcpt_group_last_error.restype = C.c_char_p
        L.mcpt_debug_counters.restype = C.c_int
        L.mcpt_debug_counters.argtypes = [C.c_void_p, C.c_void_p]
        _libs[path] = L
    return _libs[path]


def _check(rc, allow=(), L=None):
    if rc != 0 and rc not in allow:
        raise McptError(rc, (L or lib()).mcpt_last_error().decode("utf-8", "replace"))
    return rc


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class BvhInfo(C.Structure):
    _fields_ = [("n_nodes", C.c_int32), ("root", C.c_int32), ("stack_entries", C.c_int32), ("quantised", C.c_int32),
                ("root_min", C.c_float * 3), ("root_max", C.c_float * 3), ("q_origin", C.c_float * 3), ("q_cell", C.c_float * 3),
                ("n_instances", C.c_int32), ("n_leaf_prims", C.c_int32)]


def _make_desc(sd, keep):
    """SceneDesc over the arrays of a scenes.SceneData; `keep` receives the arrays that must outlive the call."""
    tri, mat, obj = np.ascontiguousarray(sd.triangles), np.ascontiguousarray(sd.materials), np.ascontiguousarray(sd.objects)
    keep.extend([tri, mat, obj])
    d = SceneDesc()
    d.n_objects, d.n_triangles, d.n_materials = len(obj), len(tri), len(mat)
    d.background = (C.c_float * 3)(*[float(x) for x in sd.background])
    d.objects, d.triangles, d.materials = _ptr(obj), _ptr(tri), _ptr(mat)
    if sd.env_pixels is not None:
        env = np.ascontiguousarray(sd.env_pixels, dtype=np.float32)
        keep.append(env)
        d.env_h, d.env_w = env.shape[:2]
        d.env_pixels = _ptr(env)
    return d


def bvh_dump(sd):
    """Host-only (no GPU): the traversal tree mcpt_scene_create would build. Returns (info dict, boxes[n,12], children[n,2], qboxes[n,12] or None)."""
    keep = []
    d = _make_desc(sd, keep)
    info = BvhInfo()
    _check(lib().mcpt_bvh_dump(C.byref(d), C.byref(info), None, None, None, None, None))
    n = info.n_nodes
    boxes = np.zeros((n, 12), np.float32)
    children = np.zeros((n, 2), np.int32)
    qboxes = np.zeros((n, 12), np.uint16)
    shift, root_first = np.zeros((info.n_instances, 3), np.float32), np.zeros((info.n_instances, 2), np.int32)
    _check(lib().mcpt_bvh_dump(C.byref(d), C.byref(info), _ptr(boxes), _ptr(children), _ptr(qboxes), _ptr(shift), _ptr(root_first)))
    return _bvh_info_dict(info, shift, root_first), boxes, children, (qboxes if info.quantised else None)


def _bvh_info_dict(info, shift, root_first):
    out = {k: (list(getattr(info, k)) if k in ("root_min", "root_max", "q_origin", "q_cell") else getattr(info, k)) for k, _ in info._fields_}
    out["inst_shift"], out["inst_root_first"] = shift, root_first
    return out


class HipScene:
    """A scene resident in the HBM of one GPU (mcpt_scene_create)."""

    def __init__(self, sd, device=-1, library=None, builder=None, quantise=-1, instancing=None):
        """library: path of an alternative build of the same ABI (the checking build); None = the product library.
        builder: None (mcpt_scene_create: environment / default) or "sah" | "reference" | "lbvh" | "ploc" (mcpt_scene_create_ex)."""
        self.sd = sd
        self._keep = []
        self.L = lib(library)
        d = _make_desc(sd, self._keep)
        h = C.c_void_p()
        self.h = None
        if builder is None and quantise == -1 and instancing is None:
            _check(self.L.mcpt_scene_create(C.byref(d), int(device), C.byref(h)), L=self.L)
        else:
            # instancing: None automatic, False never, True whenever a mesh repeats (MCPT_INSTANCING_*)
            opt = BuildOptions(builder=BUILDERS[builder or "default"], quantise=int(quantise),
                               instancing=0 if instancing is None else (2 if instancing else 1))
            _check(self.L.mcpt_scene_create_ex(C.byref(d), int(device), C.byref(opt), C.byref(h)), L=self.L)
        self.h = h

    def dump_bvh(self):
        """The traversal tree as it sits in HBM: (info dict, boxes[n,12], children[n,2], qboxes[n,12] or None)."""
        info = BvhInfo()
        _check(self.L.mcpt_scene_dump_bvh(self.h, C.byref(info), None, None, None, None, None), L=self.L)
        n = info.n_nodes
        boxes, children, qboxes = np.zeros((n, 12), np.float32), np.zeros((n, 2), np.int32), np.zeros((n, 12), np.uint16)
        shift, root_first = np.zeros((info.n_instances, 3), np.float32), np.zeros((info.n_instances, 2), np.int32)
        _check(self.L.mcpt_scene_dump_bvh(self.h, C.byref(info), _ptr(boxes), _ptr(children), _ptr(qboxes), _ptr(shift), _ptr(root_first)), L=self.L)
        return _bvh_info_dict(info, shift, root_first), boxes, children, (qboxes if info.quantised else None)

    def close(self):
        if getattr(self, "h", None):
            self.L.mcpt_scene_destroy(self.h)
            self.h = None

    def tonemap(self, fb):
        """Renderer.cpp:95-103 on the GPU: (H, W, 3) float32 -> (H, W, 4) uint8."""
        fb = np.ascontiguousarray(fb, dtype=np.float32)
        out = np.zeros(fb.shape[:-1] + (4,), dtype=np.uint8)
        _check(self.L.mcpt_tonemap(self.h, _ptr(fb), fb.size // 3, _ptr(out)), L=self.L)
        return out

    def debug_counters(self):
        out = np.zeros(16, dtype=np.uint64)
        _check(self.L.mcpt_debug_counters(self.h, _ptr(out)), L=self.L)
        return out

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self):
        i = SceneInfo()
        _check(self.L.mcpt_scene_get_info(self.h, C.byref(i)), L=self.L)
        return {k: getattr(i, k) for k, _ in i._fields_}

    def params(self, spp=None, seed=1, spp_total=0, sample_offset=0, accumulate=0, tile_size=32, rank=0, nranks=1,
               n_dir_sample=None, spp_per_pass=0, pool_paths=0, max_depth=0):
        sd = self.sd
        return Params(spp=int(spp if spp is not None else sd.spp), spp_total=int(spp_total), sample_offset=int(sample_offset),
                      rr_rate=float(sd.rr_rate), n_dir_sample=int(n_dir_sample if n_dir_sample is not None else sd.n_dir_sample),
                      enable_shadow=int(sd.enable_shadow), seed=int(seed), accumulate=int(accumulate), tile_size=int(tile_size),
                      rank=int(rank), nranks=int(nranks), spp_per_pass=int(spp_per_pass), pool_paths=int(pool_paths),
                      max_depth=int(max_depth))

    def render(self, camera=None, fb=None, **kw):
        """Renderer::Render up to the float framebuffer: returns (fb[H,W,3] float32, Stats)."""
        cam = np.ascontiguousarray(camera if camera is not None else self.sd.camera)
        W, H = int(cam["width"].reshape(-1)[0]), int(cam["height"].reshape(-1)[0])
        if fb is None:
            fb = np.zeros((H, W, 3), dtype=np.float32)
        p = self.params(**kw)
        st = Stats()
        _check(self.L.mcpt_render(self.h, _ptr(cam), C.byref(p), _ptr(fb), C.byref(st)), L=self.L)
        return fb, st

    def render_device(self, fb_ptr, stream_ptr=0, camera=None, **kw):
        """Same, into a device framebuffer (W*H*3 floats at fb_ptr) on the given hipStream_t handle."""
        cam = np.ascontiguousarray(camera if camera is not None else self.sd.camera)
        p = self.params(**kw)
        st = Stats()
        _check(self.L.mcpt_render_device(self.h, _ptr(cam), C.byref(p), C.c_void_p(int(fb_ptr)), C.c_void_p(int(stream_ptr)),
                                         C.byref(st)), L=self.L)
        return st

    MATERIAL_KINDS = {"eval": 0, "pdf": 1, "fresnel": 2, "sample": 3, "refract": 4, "eval_pdf": 5, "reflect": 6}

    def debug_material(self, kind, rows, sel):
        """The device's Material functions on arrays (mcpt_debug_material): rows [n, 13] = {a, b, c, uv, u1, u2}, sel [n, 3] =
        {material index, channel, is_reflect}; returns [n, 4]."""
        rows = np.ascontiguousarray(rows, dtype=np.float32).reshape(-1, 13)
        sel = np.ascontiguousarray(sel, dtype=np.int32).reshape(-1, 3)
        out = np.zeros((len(rows), 4), np.float32)
        _check(self.L.mcpt_debug_material(self.h, self.MATERIAL_KINDS[kind], len(rows), _ptr(rows), _ptr(sel), _ptr(out)), L=self.L)
        return out

    def sample_light(self, u):
        """The device's Scene::sampleLight for rows of four uniforms (mcpt_debug_scene) -> [n, 10] = {point, normal, emission, pdf}."""
        u = np.ascontiguousarray(u, dtype=np.float32).reshape(-1, 4)
        out = np.zeros((len(u), 10), np.float32)
        _check(self.L.mcpt_debug_scene(self.h, 0, len(u), _ptr(u), _ptr(out)), L=self.L)
        return out

    def sample_env(self, dirs):
        """The device's Scene::sampleEnv for rows of directions -> [n, 3]."""
        d = np.ascontiguousarray(dirs, dtype=np.float32).reshape(-1, 3)
        out = np.zeros((len(d), 3), np.float32)
        _check(self.L.mcpt_debug_scene(self.h, 1, len(d), _ptr(d), _ptr(out)), L=self.L)
        return out

    def intersect(self, origins, dirs):
        o = np.ascontiguousarray(origins, dtype=np.float32)
        d = np.ascontiguousarray(dirs, dtype=np.float32)
        n = len(o)
        t = np.zeros(n, dtype=np.float64)
        prim = np.zeros(n, dtype=np.int32)
        _check(self.L.mcpt_intersect(self.h, n, _ptr(o), _ptr(d), _ptr(t), _ptr(prim)), L=self.L)
        return t, prim

    def cast_rays(self, origins, dirs, pixel, sample, channel, **kw):
        o = np.ascontiguousarray(origins, dtype=np.float32)
        d = np.ascontiguousarray(dirs, dtype=np.float32)
        n = len(o)
        px = np.ascontiguousarray(pixel, dtype=np.uint32)
        sm = np.ascontiguousarray(sample, dtype=np.uint32)
        ch = np.ascontiguousarray(channel, dtype=np.int32)
        out = np.zeros(n, dtype=np.float32)
        p = self.params(**kw)
        _check(self.L.mcpt_cast_rays(self.h, C.byref(p), n, _ptr(o), _ptr(d), _ptr(px), _ptr(sm), _ptr(ch), _ptr(out)), L=self.L)
        return out

    def camera_rays(self, pixels, samples, seed=1, camera=None):
        cam = np.ascontiguousarray(camera if camera is not None else self.sd.camera)
        px = np.ascontiguousarray(pixels, dtype=np.uint32)
        sm = np.ascontiguousarray(samples, dtype=np.uint32)
        n = len(px)
        o = np.zeros((n, 3), dtype=np.float32)
        d = np.zeros((n, 3), dtype=np.float32)
        _check(self.L.mcpt_camera_rays(self.h, _ptr(cam), int(seed), n, _ptr(px), _ptr(sm), _ptr(o), _ptr(d)), L=self.L)
        return o, d


class HipGroup:
    """One replica of the scene per listed device; render() = mcpt_group_render (tile partition + RCCL merge inside the library)."""

    def __init__(self, sd, devices, library=None):
        self.sd = sd
        self._keep = []
        self.L = lib(library)
        d = _make_desc(sd, self._keep)
        dev = np.ascontiguousarray(devices, dtype=np.int32)
        h = C.c_void_p()
        self.h = None
        rc = self.L.mcpt_group_create(C.byref(d), len(dev), _ptr(dev), C.byref(h))
        if rc != 0:
            raise McptError(rc, self.L.mcpt_group_last_error().decode("utf-8", "replace"))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.mcpt_group_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self):
        i = GroupInfo()
        rc = self.L.mcpt_group_get_info(self.h, C.byref(i))
        if rc != 0:
            raise McptError(rc, self.L.mcpt_group_last_error().decode("utf-8", "replace"))
        return {k: getattr(i, k) for k, _ in i._fields_}

    def render(self, camera=None, fb=None, **kw):
        cam = np.ascontiguousarray(camera if camera is not None else self.sd.camera)
        W, H = int(cam["width"].reshape(-1)[0]), int(cam["height"].reshape(-1)[0])
        if fb is None:
            fb = np.zeros((H, W, 3), dtype=np.float32)
        p = HipScene.params(self, **kw)
        st = Stats()
        rc = self.L.mcpt_group_render(self.h, _ptr(cam), C.byref(p), _ptr(fb), C.byref(st))
        if rc != 0:
            raise McptError(rc, self.L.mcpt_group_last_error().decode("utf-8", "replace"))
        return fb, st


_FMATH_SCENE = None


def debug_fmath(kind, x, y=None):
    """csrc/mcpt_fmath.h evaluated on the device (mcpt_debug_fmath): kind "sin" | "cos" | "atan2" | "acos"."""
    global _FMATH_SCENE
    if _FMATH_SCENE is None:
        from . import scenes
        _FMATH_SCENE = HipScene(scenes.cornell_rc(8, 8, 1))
    k = {"sin": 0, "cos": 1, "atan2": 2, "acos": 3, "pow": 4, "tonemap": 5}[kind]
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.ascontiguousarray(y if y is not None else np.zeros_like(x), dtype=np.float32)
    out = np.zeros_like(x)
    _check(lib().mcpt_debug_fmath(_FMATH_SCENE.h, k, x.size, _ptr(x), _ptr(y), _ptr(out)))
    return out
