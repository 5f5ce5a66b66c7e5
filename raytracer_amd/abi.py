"""ctypes mirror of include/mi355rt.h and the loaders of the two native libraries.

Nothing here computes: a missing library is an error, never a fallback.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "lib")

RT_OK = 0
RT_STAT_COUNT = 24
STAT_NAMES = [
    "rays", "box_tests", "sphere_tests", "msphere_tests", "rect_tests", "medium_tests",
    "op_translate", "op_rotate", "hits", "shade_miss", "shade_lambertian", "shade_metal",
    "shade_dielectric", "shade_light", "shade_isotropic", "tex_checker", "tex_perlin", "tex_image",
    "sphere_roots", "node_visits", "extend_ns", "extend_launches", "shade_ns", "shade_launches",
]
RT_FLAG_STATS = 1
RT_FLAG_TIMING = 2
RT_FLAG_BINNED = 4
MISS_ID = 0xFFFFFFFF


class RtError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"[{code}] {message}")
        self.code = code
        self.message = message


class Camera(C.Structure):
    _fields_ = [
        ("origin", C.c_double * 3), ("lower_left", C.c_double * 3),
        ("horizontal", C.c_double * 3), ("vertical", C.c_double * 3),
        ("u", C.c_double * 3), ("v", C.c_double * 3),
        ("lens_radius", C.c_double), ("time0", C.c_double), ("time1", C.c_double),
    ]


class CameraDesc(C.Structure):
    _fields_ = [
        ("look_from", C.c_double * 3), ("look_at", C.c_double * 3), ("v_up", C.c_double * 3),
        ("vertical_fov", C.c_double), ("aperture", C.c_double), ("focus_distance", C.c_double),
        ("shutter0", C.c_double), ("shutter1", C.c_double),
        ("has_look_at", C.c_uint32), ("has_v_up", C.c_uint32),
        ("has_focus_distance", C.c_uint32), ("has_shutter", C.c_uint32),
    ]


class Prim(C.Structure):
    _fields_ = [
        ("kind", C.c_uint32), ("axis", C.c_uint32), ("material", C.c_uint32), ("reserved", C.c_uint32),
        ("center", C.c_double * 3), ("center1", C.c_double * 3), ("radius", C.c_double),
        ("time0", C.c_double), ("time1", C.c_double), ("width", C.c_double), ("height", C.c_double),
    ]


class Op(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("reserved", C.c_uint32), ("a", C.c_double * 3)]


class Object(C.Structure):
    _fields_ = [
        ("first_prim", C.c_uint32), ("prim_count", C.c_uint32),
        ("first_op", C.c_uint32), ("op_count", C.c_uint32),
        ("inner_first_op", C.c_uint32), ("inner_op_count", C.c_uint32),
        ("is_medium", C.c_uint32), ("medium_material", C.c_uint32),
        ("is_cuboid", C.c_uint32), ("reserved", C.c_uint32),
        ("neg_inv_density", C.c_double),
        ("cuboid_center", C.c_double * 3), ("cuboid_half", C.c_double * 3),
    ]


class Material(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("texture", C.c_uint32), ("albedo", C.c_double * 3),
                ("param", C.c_double)]


class Texture(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("a", C.c_uint32), ("b", C.c_uint32), ("reserved", C.c_uint32),
                ("color", C.c_double * 3), ("scale", C.c_double)]


class Perlin(C.Structure):
    _fields_ = [("rand_vec", (C.c_double * 3) * 256), ("perm_x", C.c_uint32 * 256),
                ("perm_y", C.c_uint32 * 256), ("perm_z", C.c_uint32 * 256)]


class Image(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("offset", C.c_uint64)]


class Scene(C.Structure):
    _fields_ = [
        ("camera", Camera), ("background", C.c_double * 3),
        ("width", C.c_uint32), ("height", C.c_uint32),
        ("samples_per_pixel", C.c_uint32), ("max_depth", C.c_uint32),
        ("n_objects", C.c_uint32), ("n_prims", C.c_uint32), ("n_ops", C.c_uint32),
        ("n_materials", C.c_uint32), ("n_textures", C.c_uint32), ("n_perlin", C.c_uint32),
        ("n_images", C.c_uint32), ("reserved", C.c_uint32), ("n_texel_bytes", C.c_uint64),
        ("objects", C.POINTER(Object)), ("prims", C.POINTER(Prim)), ("ops", C.POINTER(Op)),
        ("materials", C.POINTER(Material)), ("textures", C.POINTER(Texture)),
        ("perlin", C.POINTER(Perlin)), ("images", C.POINTER(Image)),
        ("texels", C.POINTER(C.c_uint8)),
    ]


class HostOptions(C.Structure):
    _fields_ = [("scene_seed", C.c_uint64), ("width", C.c_uint32), ("height", C.c_uint32),
                ("samples_per_pixel", C.c_uint32), ("max_depth", C.c_uint32)]


class Config(C.Structure):
    _fields_ = [("device", C.c_int32), ("n_devices", C.c_uint32),
                ("max_paths_in_flight", C.c_uint64), ("devices", C.POINTER(C.c_int32))]


class Window(C.Structure):
    _fields_ = [("x0", C.c_uint32), ("y0", C.c_uint32), ("w", C.c_uint32), ("h", C.c_uint32)]


class RenderParams(C.Structure):
    _fields_ = [
        ("width", C.c_uint32), ("height", C.c_uint32),
        ("sample_begin", C.c_uint32), ("sample_end", C.c_uint32),
        ("max_depth", C.c_uint32), ("flags", C.c_uint32), ("seed", C.c_uint64),
        ("window", Window),
    ]


class GroupReport(C.Structure):
    _fields_ = [("n_members", C.c_uint32), ("reduce_path", C.c_uint32), ("render_ms_min", C.c_double),
                ("render_ms_max", C.c_double), ("reduce_ms", C.c_double)]


REDUCE_PATHS = ["none", "rccl", "host-staged"]  # rt_reduce_path


class BvhNode(C.Structure):
    _fields_ = [("lo", C.c_float * 3), ("hi", C.c_float * 3), ("a", C.c_uint32), ("b", C.c_uint32)]


class BvhItem(C.Structure):
    _fields_ = [("object", C.c_uint32), ("prim", C.c_uint32), ("rank", C.c_uint32),
                ("reserved", C.c_uint32)]


# void fn(void* user, uint64_t done, uint64_t total)
ProgressFn = C.CFUNCTYPE(None, C.c_void_p, C.c_uint64, C.c_uint64)

HOST_SYMBOLS = [
    "rt_host_add_asset_dir", "rt_host_load_ron", "rt_host_load_ron_text", "rt_host_save_frozen",
    "rt_host_load_frozen", "rt_host_generate_cover", "rt_host_set_image", "rt_host_scene_view",
    "rt_host_camera_desc", "rt_host_free", "rt_host_last_error", "rt_host_resolve_srgb8",
    "rt_host_resolve_srgb8_f32", "rt_host_write_png", "rt_host_human_duration", "rt_host_decode_jpeg",
]
DEVICE_SYMBOLS = [
    "rt_create", "rt_upload_scene", "rt_render", "rt_render_f64", "rt_render_device",
    "rt_render_device_noalloc", "rt_set_progress", "rt_group_info", "rt_first_hit", "rt_bvh_build", "rt_bvh_info", "rt_bvh_copy", "rt_bvh_free", "rt_last_error",
    "rt_destroy",
]

_host = None
_device = None


def _open(name):
    # MI355RT_LIB_DIR points at an alternative build of the same libraries (A/B timing)
    path = os.path.join(os.environ.get("MI355RT_LIB_DIR", LIB_DIR), name)
    if not os.path.exists(path):
        raise RtError(-4, f"{path} is missing: run `make` (or __graft_entry__.build()) first")
    return C.CDLL(path, mode=C.RTLD_GLOBAL)


def host_lib():
    """libmi355rt_host.so with argument types declared."""
    global _host
    if _host is not None:
        return _host
    L = _open("libmi355rt_host.so")
    vp, cp = C.c_void_p, C.c_char_p
    L.rt_host_add_asset_dir.argtypes = [cp]
    L.rt_host_load_ron.argtypes = [cp, C.POINTER(HostOptions), C.POINTER(vp)]
    L.rt_host_load_ron_text.argtypes = [cp, cp, C.POINTER(HostOptions), C.POINTER(vp)]
    L.rt_host_save_frozen.argtypes = [vp, cp]
    L.rt_host_load_frozen.argtypes = [cp, C.POINTER(vp)]
    L.rt_host_generate_cover.argtypes = [C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32,
                                         C.c_uint32, C.c_uint32, C.POINTER(vp)]
    L.rt_host_set_image.argtypes = [vp, C.c_uint32, C.c_uint32]
    L.rt_host_scene_view.argtypes = [vp]
    L.rt_host_scene_view.restype = C.POINTER(Scene)
    L.rt_host_camera_desc.argtypes = [vp]
    L.rt_host_camera_desc.restype = C.POINTER(CameraDesc)
    L.rt_host_free.argtypes = [vp]
    L.rt_host_free.restype = None
    L.rt_host_last_error.restype = cp
    L.rt_host_resolve_srgb8.argtypes = [vp, C.c_uint64, C.c_uint32, vp]
    L.rt_host_resolve_srgb8.restype = None
    L.rt_host_resolve_srgb8_f32.argtypes = [vp, C.c_uint64, C.c_uint32, vp]
    L.rt_host_resolve_srgb8_f32.restype = None
    L.rt_host_write_png.argtypes = [cp, C.c_uint32, C.c_uint32, vp]
    L.rt_host_human_duration.argtypes = [C.c_uint64, cp, C.c_size_t]
    L.rt_host_decode_jpeg.argtypes = [vp, C.c_uint64, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), vp,
                                      C.c_uint64]
    _host = L
    return L


def device_lib():
    """libmi355rt.so (the HIP render path) with argument types declared."""
    global _device
    if _device is not None:
        return _device
    L = _open("libmi355rt.so")
    vp = C.c_void_p
    L.rt_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    L.rt_upload_scene.argtypes = [vp, C.POINTER(Scene)]
    L.rt_render.argtypes = [vp, C.POINTER(RenderParams), vp, vp]
    L.rt_render_f64.argtypes = [vp, C.POINTER(RenderParams), vp, vp]
    L.rt_render_device.argtypes = [vp, C.POINTER(RenderParams), vp, vp]
    L.rt_render_device_noalloc.argtypes = [vp, C.POINTER(RenderParams), vp]
    L.rt_set_progress.argtypes = [vp, ProgressFn, vp]
    L.rt_group_info.argtypes = [vp, C.POINTER(GroupReport)]
    L.rt_first_hit.argtypes = [vp, C.c_uint32, C.c_uint32, C.POINTER(Window), vp, vp, vp]
    L.rt_bvh_build.argtypes = [C.POINTER(Scene), C.POINTER(vp)]
    L.rt_bvh_info.argtypes = [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                              C.POINTER(C.c_uint32)]
    L.rt_bvh_copy.argtypes = [vp, vp, vp]
    L.rt_bvh_free.argtypes = [vp]
    L.rt_bvh_free.restype = None
    L.rt_last_error.argtypes = [vp]
    L.rt_last_error.restype = C.c_char_p
    L.rt_destroy.argtypes = [vp]
    L.rt_destroy.restype = None
    _device = L
    return L
