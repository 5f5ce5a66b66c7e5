"""Host-side scene handle: a thin owner of an `rt_host_scene*`.

Mirrors the reference's `SceneLoader::new(path).load()` (src/scene/loader.rs:39-93):
same scene files in, a realised scene out, errors carrying the loader's messages.
"""
import ctypes as C

import numpy as np

from . import abi


def _check(code):
    if code != abi.RT_OK:
        raise abi.RtError(code, abi.host_lib().rt_host_last_error().decode("utf-8", "replace"))


class HostScene:
    def __init__(self, handle):
        self._h = handle

    # -- constructors -------------------------------------------------------
    @staticmethod
    def _options(scene_seed, width, height, spp, max_depth):
        return abi.HostOptions(scene_seed, width or 0, height or 0, spp or 0, max_depth or 0)

    @classmethod
    def load_ron(cls, path, scene_seed=1, width=None, height=None, spp=None, max_depth=None):
        L = abi.host_lib()
        h = C.c_void_p()
        opt = cls._options(scene_seed, width, height, spp, max_depth)
        _check(L.rt_host_load_ron(str(path).encode(), C.byref(opt), C.byref(h)))
        return cls(h)

    @classmethod
    def load_ron_text(cls, text, base_dir="", scene_seed=1, width=None, height=None, spp=None,
                      max_depth=None):
        L = abi.host_lib()
        h = C.c_void_p()
        opt = cls._options(scene_seed, width, height, spp, max_depth)
        _check(L.rt_host_load_ron_text(text.encode(), str(base_dir).encode(), C.byref(opt),
                                       C.byref(h)))
        return cls(h)

    @classmethod
    def load_frozen(cls, path):
        L = abi.host_lib()
        h = C.c_void_p()
        _check(L.rt_host_load_frozen(str(path).encode(), C.byref(h)))
        return cls(h)

    @classmethod
    def generate_cover(cls, half_extent, scene_seed=1, width=1920, height=1080, spp=1024,
                       max_depth=50):
        L = abi.host_lib()
        h = C.c_void_p()
        _check(L.rt_host_generate_cover(half_extent, scene_seed, width, height, spp, max_depth,
                                        C.byref(h)))
        return cls(h)

    # -- accessors ----------------------------------------------------------
    @property
    def view(self):
        """The `rt_scene` (ctypes structure, borrowed from the handle)."""
        return abi.host_lib().rt_host_scene_view(self._h).contents

    @property
    def view_ptr(self):
        return abi.host_lib().rt_host_scene_view(self._h)

    @property
    def camera_desc(self):
        return abi.host_lib().rt_host_camera_desc(self._h).contents

    @property
    def width(self):
        return self.view.width

    @property
    def height(self):
        return self.view.height

    @property
    def samples_per_pixel(self):
        return self.view.samples_per_pixel

    @property
    def max_depth(self):
        return self.view.max_depth

    def set_image(self, width, height):
        _check(abi.host_lib().rt_host_set_image(self._h, width, height))

    def save_frozen(self, path):
        _check(abi.host_lib().rt_host_save_frozen(self._h, str(path).encode()))

    def close(self):
        if self._h:
            abi.host_lib().rt_host_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def add_asset_dir(path):
    _check(abi.host_lib().rt_host_add_asset_dir(str(path).encode()))


def resolve_srgb8(rgb_sum, samples_per_pixel):
    """Color::into_srgb8 (src/geom/vec3.rs:123-133) over an (h, w, 3) array of sums."""
    a = np.ascontiguousarray(rgb_sum)
    out = np.empty(a.shape, dtype=np.uint8)
    n = a.size // 3
    L = abi.host_lib()
    if a.dtype == np.float64:
        L.rt_host_resolve_srgb8(a.ctypes.data, n, samples_per_pixel, out.ctypes.data)
    elif a.dtype == np.float32:
        L.rt_host_resolve_srgb8_f32(a.ctypes.data, n, samples_per_pixel, out.ctypes.data)
    else:
        raise TypeError("rgb_sum must be float32 or float64")
    return out


def write_png(path, rgb8):
    a = np.ascontiguousarray(rgb8, dtype=np.uint8)
    h, w, _ = a.shape
    _check(abi.host_lib().rt_host_write_png(str(path).encode(), w, h, a.ctypes.data))


def decode_jpeg(data):
    """Baseline JPEG bytes -> (h, w, 3) uint8, with the decoder Image("x.jpg") textures go through."""
    L = abi.host_lib()
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
    w, h = C.c_uint32(), C.c_uint32()
    _check(L.rt_host_decode_jpeg(buf, len(data), C.byref(w), C.byref(h), None, 0))
    out = np.empty((h.value, w.value, 3), dtype=np.uint8)
    _check(L.rt_host_decode_jpeg(buf, len(data), C.byref(w), C.byref(h), out.ctypes.data, out.nbytes))
    return out


def human_duration(millis):
    buf = C.create_string_buffer(64)
    abi.host_lib().rt_host_human_duration(int(millis), buf, 64)
    return buf.value.decode()
