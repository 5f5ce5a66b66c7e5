"""Render context: a thin owner of an `rt_ctx*` (the HIP render path).

Plays the part of the reference's render loop (src/main.rs:57-101): give it a
realised scene, get back per-pixel sums of ray_color over a sample range.  There
is no CPU path here: without libmi355rt.so or without a HIP device every call
fails.
"""
import ctypes as C

import numpy as np

from . import abi


class Context:
    def __init__(self, device=0, max_paths_in_flight=0, devices=None):
        """devices: a list of HIP ordinals makes a device group (devices[0] is the root)."""
        self._h = C.c_void_p()
        L = abi.device_lib()
        if devices:
            self._devices = (C.c_int32 * len(devices))(*devices)
            device = devices[0]
            cfg = abi.Config(device, len(devices), max_paths_in_flight, self._devices)
        else:
            cfg = abi.Config(device, 0, max_paths_in_flight, None)
        code = L.rt_create(C.byref(cfg), C.byref(self._h))
        if code != abi.RT_OK:
            self._h = None
            raise abi.RtError(code, f"rt_create failed on device {device} (is a HIP device present?)")
        self.device = device

    def _check(self, code):
        if code != abi.RT_OK:
            msg = abi.device_lib().rt_last_error(self._h)
            raise abi.RtError(code, msg.decode("utf-8", "replace") if msg else "")

    def upload(self, scene):
        """scene: a HostScene (or anything with .view_ptr -> POINTER(rt_scene))."""
        self._check(abi.device_lib().rt_upload_scene(self._h, scene.view_ptr))

    @staticmethod
    def params(width, height, sample_begin, sample_end, max_depth, seed=1, window=None, flags=0):
        p = abi.RenderParams()
        p.width, p.height = width, height
        p.sample_begin, p.sample_end = sample_begin, sample_end
        p.max_depth, p.flags, p.seed = max_depth, flags, seed
        if window is not None:
            p.window = abi.Window(*window)
        return p

    @staticmethod
    def _shape(p):
        if p.window.w == 0 or p.window.h == 0:
            return p.height, p.width
        return p.window.h, p.window.w

    def render(self, p, dtype=np.float32, stats=False, timing=False):
        """Sums of ray_color over [sample_begin, sample_end) per window pixel, (h, w, 3).

        stats / timing: also return the RT_STAT_* array (operation counts from the
        instrumented kernels / device time of the extend and shade launches).
        """
        L = abi.device_lib()
        h, w = self._shape(p)
        out = np.empty((h, w, 3), dtype=dtype)
        st = None
        if stats:
            p.flags |= abi.RT_FLAG_STATS
        if timing:
            p.flags |= abi.RT_FLAG_TIMING
        stats = stats or timing
        if stats:
            st = np.zeros(abi.RT_STAT_COUNT, dtype=np.uint64)
        fn = L.rt_render if dtype == np.float32 else L.rt_render_f64
        self._check(fn(self._h, C.byref(p), out.ctypes.data, st.ctypes.data if stats else None))
        return (out, st) if stats else out

    def render_into(self, p, device_ptr, timing=False):
        """Same sums as float32 into device memory at `device_ptr` (e.g. tensor.data_ptr())."""
        st = None
        if timing:
            p.flags |= abi.RT_FLAG_TIMING
            st = np.zeros(abi.RT_STAT_COUNT, dtype=np.uint64)
        self._check(abi.device_lib().rt_render_device(self._h, C.byref(p), C.c_void_p(device_ptr),
                                                      st.ctypes.data if timing else None))
        return st

    def render_into_noalloc(self, p, device_ptr):
        """render_into through the entry point that allocates nothing (fixed scratch array)."""
        self._check(abi.device_lib().rt_render_device_noalloc(self._h, C.byref(p), C.c_void_p(device_ptr)))

    def set_progress(self, fn):
        """fn(done, total), in pixel-samples of the running call, after every finished pass; None removes it."""
        # (the ctypes thunk has to outlive the registration)
        self._progress = abi.ProgressFn((lambda user, done, total: fn(done, total))) if fn else abi.ProgressFn()
        self._check(abi.device_lib().rt_set_progress(self._h, self._progress, None))

    def group_info(self):
        """(members, "none" | "rccl" | "host-staged", {render_ms_min, render_ms_max, reduce_ms} of the last render)."""
        r = abi.GroupReport()
        self._check(abi.device_lib().rt_group_info(self._h, C.byref(r)))
        return r.n_members, abi.REDUCE_PATHS[r.reduce_path], {
            "render_ms_min": r.render_ms_min, "render_ms_max": r.render_ms_max, "reduce_ms": r.reduce_ms}

    def first_hit(self, width, height, window=None):
        L = abi.device_lib()
        w = abi.Window(*window) if window is not None else abi.Window(0, 0, 0, 0)
        hh, ww = (height, width) if window is None else (window[3], window[2])
        ids = np.empty((hh, ww), dtype=np.uint32)
        t = np.empty((hh, ww), dtype=np.float64)
        n = np.empty((hh, ww, 3), dtype=np.float64)
        self._check(L.rt_first_hit(self._h, width, height, C.byref(w), ids.ctypes.data,
                                   t.ctypes.data, n.ctypes.data))
        return ids, t, n

    def close(self):
        if self._h:
            abi.device_lib().rt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def build_bvh(scene):
    """(nodes, items, depth) of the flattened acceleration structure rt_upload_scene builds.

    Host only: needs no device.  nodes/items are structured arrays laid out as
    rt_bvh_node / rt_bvh_item.
    """
    L = abi.device_lib()
    h = C.c_void_p()
    code = L.rt_bvh_build(scene.view_ptr, C.byref(h))
    if code != abi.RT_OK:
        raise abi.RtError(code, "rt_bvh_build failed")
    try:
        nn, ni, depth = C.c_uint32(), C.c_uint32(), C.c_uint32()
        L.rt_bvh_info(h, C.byref(nn), C.byref(ni), C.byref(depth))
        node_t = np.dtype([("lo", np.float32, 3), ("hi", np.float32, 3), ("a", np.uint32),
                           ("b", np.uint32)])
        item_t = np.dtype([("object", np.uint32), ("prim", np.uint32), ("rank", np.uint32),
                           ("reserved", np.uint32)])
        nodes = np.zeros(max(nn.value, 1), dtype=node_t)
        items = np.zeros(max(ni.value, 1), dtype=item_t)
        code = L.rt_bvh_copy(h, nodes.ctypes.data, items.ctypes.data)
        if code != abi.RT_OK:
            raise abi.RtError(code, "rt_bvh_copy failed")
        return nodes[:nn.value], items[:ni.value], depth.value
    finally:
        L.rt_bvh_free(h)
