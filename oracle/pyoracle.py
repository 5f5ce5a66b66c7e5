"""ctypes access to oracle/liboracle.so, the CPU restatement of the reference.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Product code never imports this module.
"""
import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_HERE))
from raytracer_amd import abi  # noqa: E402  (struct layouts of include/mi355rt.h)

MODE_REFERENCE = 0       # reference BVH topology, recursive ray_color
MODE_ITERATIVE = 1       # reference BVH topology, iterative estimator
MODE_FLAT = 2            # flattened BVH of the product, iterative estimator


class Hit(C.Structure):
    _fields_ = [("p", C.c_double * 3), ("normal", C.c_double * 3), ("t", C.c_double),
                ("u", C.c_double), ("v", C.c_double), ("object", C.c_uint32),
                ("material", C.c_int32), ("front_face", C.c_uint32), ("hit", C.c_uint32)]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    path = os.path.join(_HERE, "liboracle.so")
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: run `make oracle`")
    L = C.CDLL(path)
    vp, d3 = C.c_void_p, C.POINTER(C.c_double)
    L.oracle_create.argtypes = [C.POINTER(abi.Scene), C.POINTER(vp)]
    L.oracle_destroy.argtypes = [vp]
    L.oracle_destroy.restype = None
    L.oracle_world_hit.argtypes = [vp, d3, d3, C.c_double, C.c_double, C.c_double, C.c_uint64,
                                   C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(Hit)]
    L.oracle_scatter.argtypes = [vp, C.c_int32, d3, d3, C.c_double, C.POINTER(Hit), C.c_uint64,
                                 C.c_uint32, C.c_uint32, C.c_uint32, d3, d3, d3, d3,
                                 C.POINTER(C.c_int)]
    L.oracle_texture_value.argtypes = [vp, C.c_uint32, C.c_double, C.c_double, d3, d3]
    L.oracle_camera_ray.argtypes = [vp, C.POINTER(abi.RenderParams), C.c_uint32, C.c_uint32,
                                    C.c_uint32, d3, d3, d3]
    L.oracle_first_hit.argtypes = [vp, C.c_uint32, C.c_uint32, C.POINTER(abi.Window), vp, vp, vp]
    L.oracle_first_hit_mt.argtypes = [vp, C.c_uint32, C.c_uint32, C.POINTER(abi.Window), C.c_int, vp, vp, vp]
    L.oracle_render.argtypes = [vp, C.POINTER(abi.RenderParams), C.c_int, vp, C.c_uint32, vp,
                                C.c_uint32, C.c_int, vp, vp]
    L.oracle_ranks.argtypes = [vp, vp, C.c_uint32]
    L.oracle_bvh_shape.argtypes = [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                   C.POINTER(C.c_uint32)]
    L.oracle_philox.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                C.POINTER(C.c_uint32)]
    L.oracle_philox.restype = None
    L.oracle_vec3_project.argtypes = [d3, d3, d3]
    L.oracle_vec3_project.restype = None
    L.oracle_aabb_surrounding.argtypes = [d3, C.c_uint32, d3]
    L.oracle_aabb_intersect.argtypes = [d3, d3, d3, d3, C.c_double, C.c_double]
    L.oracle_reflect.argtypes = [d3, d3, d3]
    L.oracle_reflect.restype = None
    L.oracle_refract.argtypes = [d3, d3, C.c_double, d3]
    L.oracle_refract.restype = None
    L.oracle_reflectance.argtypes = [C.c_double, C.c_double]
    L.oracle_reflectance.restype = C.c_double
    L.oracle_moving_center.argtypes = [d3, d3, C.c_double, C.c_double, C.c_double, d3]
    L.oracle_moving_center.restype = None
    L.oracle_into_srgb8.argtypes = [d3, C.c_uint32, C.POINTER(C.c_uint8)]
    L.oracle_into_srgb8.restype = None
    L.oracle_camera_build.argtypes = [C.POINTER(abi.CameraDesc), C.c_uint32, C.c_uint32,
                                      C.POINTER(abi.Camera)]
    L.oracle_hardware_threads.restype = C.c_int
    _lib = L
    return L


def _d3(v):
    return (C.c_double * 3)(*[float(x) for x in v])


def _check(code, what):
    if code != 0:
        raise RuntimeError(f"oracle: {what} failed with {code}")


def make_params(width, height, sample_begin, sample_end, max_depth, seed, window=None, flags=0):
    p = abi.RenderParams()
    p.width, p.height = width, height
    p.sample_begin, p.sample_end = sample_begin, sample_end
    p.max_depth, p.flags, p.seed = max_depth, flags, seed
    if window is not None:
        p.window = abi.Window(*window)
    return p


def window_shape(params):
    if params.window.w == 0 or params.window.h == 0:
        return params.height, params.width
    return params.window.h, params.window.w


class Oracle:
    """The reference's world + camera + materials rebuilt on the CPU from an rt_scene."""

    def __init__(self, scene_view_ptr):
        self._h = C.c_void_p()
        _check(lib().oracle_create(scene_view_ptr, C.byref(self._h)), "create")

    def close(self):
        if self._h:
            lib().oracle_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def world_hit(self, origin, direction, time=0.0, t_start=0.001, t_end=float("inf"),
                  seed=0, pixel=0, sample=0, bounce=0, with_media=False):
        h = Hit()
        _check(lib().oracle_world_hit(self._h, _d3(origin), _d3(direction), time, t_start, t_end,
                                      seed, pixel, sample, bounce, int(with_media), C.byref(h)),
               "world_hit")
        return h

    def scatter(self, material, origin, direction, time, rec, seed=0, pixel=0, sample=0, bounce=0):
        att, emi, oo, od = _d3([0] * 3), _d3([0] * 3), _d3([0] * 3), _d3([0] * 3)
        sc = C.c_int()
        _check(lib().oracle_scatter(self._h, material, _d3(origin), _d3(direction), time,
                                    C.byref(rec), seed, pixel, sample, bounce, att, emi, oo, od,
                                    C.byref(sc)), "scatter")
        return dict(attenuation=list(att), emitted=list(emi), origin=list(oo), direction=list(od),
                    scattered=bool(sc.value))

    def texture_value(self, texture, u, v, p):
        out = _d3([0] * 3)
        _check(lib().oracle_texture_value(self._h, texture, u, v, _d3(p), out), "texture_value")
        return list(out)

    def camera_ray(self, params, x, y, s):
        o, d, t = _d3([0] * 3), _d3([0] * 3), C.c_double()
        _check(lib().oracle_camera_ray(self._h, C.byref(params), x, y, s, o, d,
                                       C.cast(C.byref(t), C.POINTER(C.c_double))), "camera_ray")
        return list(o), list(d), t.value

    def first_hit(self, width, height, window=None, threads=1):
        w = abi.Window(*window) if window is not None else abi.Window(0, 0, 0, 0)
        hh, ww = (height, width) if window is None else (window[3], window[2])
        ids = np.empty((hh, ww), dtype=np.uint32)
        t = np.empty((hh, ww), dtype=np.float64)
        n = np.empty((hh, ww, 3), dtype=np.float64)
        _check(lib().oracle_first_hit_mt(self._h, width, height, C.byref(w), threads, ids.ctypes.data,
                                         t.ctypes.data, n.ctypes.data), "first_hit")
        return ids, t, n

    def render(self, params, mode=MODE_REFERENCE, bvh=None, threads=1, stats=False):
        """Sum over the sample range of ray_color per pixel: (h, w, 3) float64."""
        hh, ww = window_shape(params)
        out = np.empty((hh, ww, 3), dtype=np.float64)
        st = np.zeros(abi.RT_STAT_COUNT, dtype=np.uint64) if stats else None
        nodes_p = items_p = None
        n_nodes = n_items = 0
        if mode == MODE_FLAT:
            nodes, items = bvh
            nodes_p, items_p = nodes.ctypes.data, items.ctypes.data
            n_nodes, n_items = len(nodes), len(items)
        _check(lib().oracle_render(self._h, C.byref(params), mode, nodes_p, n_nodes, items_p,
                                   n_items, threads, out.ctypes.data,
                                   st.ctypes.data if stats else None), "render")
        return (out, st) if stats else out

    def ranks(self, n_objects):
        r = np.empty(n_objects, dtype=np.uint32)
        _check(lib().oracle_ranks(self._h, r.ctypes.data, n_objects), "ranks")
        return r

    def bvh_shape(self):
        a, b, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
        _check(lib().oracle_bvh_shape(self._h, C.byref(a), C.byref(b), C.byref(c)), "bvh_shape")
        return a.value, b.value, c.value


def philox(ctr, key):
    out = (C.c_uint32 * 4)()
    lib().oracle_philox((C.c_uint32 * 4)(*ctr), (C.c_uint32 * 2)(*key), out)
    return list(out)


def vec3_project(a, b):
    out = _d3([0] * 3)
    lib().oracle_vec3_project(_d3(a), _d3(b), out)
    return list(out)


def aabb_surrounding(boxes):
    flat = np.ascontiguousarray(boxes, dtype=np.float64).reshape(-1)
    out = (C.c_double * 6)()
    some = lib().oracle_aabb_surrounding(flat.ctypes.data_as(C.POINTER(C.c_double)),
                                         len(flat) // 6, C.cast(out, C.POINTER(C.c_double)))
    return list(out) if some else None


def aabb_intersect(bmin, bmax, origin, direction, t_start, t_end):
    return bool(lib().oracle_aabb_intersect(_d3(bmin), _d3(bmax), _d3(origin), _d3(direction),
                                            t_start, t_end))


def reflect(v, n):
    out = _d3([0] * 3)
    lib().oracle_reflect(_d3(v), _d3(n), out)
    return list(out)


def refract(v, n, eta):
    out = _d3([0] * 3)
    lib().oracle_refract(_d3(v), _d3(n), eta, out)
    return list(out)


def reflectance(cosine, ref_idx):
    return lib().oracle_reflectance(cosine, ref_idx)


def moving_center(c0, c1, t0, t1, time):
    out = _d3([0] * 3)
    lib().oracle_moving_center(_d3(c0), _d3(c1), t0, t1, time, out)
    return list(out)


def into_srgb8(rgb_sum, spp):
    out = (C.c_uint8 * 3)()
    lib().oracle_into_srgb8(_d3(rgb_sum), spp, out)
    return list(out)


def camera_build(desc, width, height):
    cam = abi.Camera()
    _check(lib().oracle_camera_build(C.byref(desc), width, height, C.byref(cam)), "camera_build")
    return cam


def hardware_threads():
    return lib().oracle_hardware_threads()
