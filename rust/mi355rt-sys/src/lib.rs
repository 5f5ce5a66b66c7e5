//! Raw bindings to `include/mi355rt.h`, field for field.  Never compiled in the image this was
//! written in (no Rust toolchain there); the size checks below are the ones the C++ and Python
//! sides are held to (`tests/test_abi.py`).
#![allow(non_camel_case_types)]
use core::mem::size_of;
use std::os::raw::{c_char, c_int, c_void};

pub const RT_OK: c_int = 0;
pub const RT_ERR_INVALID: c_int = -1;
pub const RT_ERR_DEVICE: c_int = -2;
pub const RT_ERR_NOMEM: c_int = -3;
pub const RT_ERR_IO: c_int = -4;
pub const RT_ERR_PARSE: c_int = -5;
pub const RT_ERR_UNSUPPORTED: c_int = -6;
pub const RT_ERR_LIMIT: c_int = -7;

pub const RT_PRIM_SPHERE: u32 = 0;
pub const RT_PRIM_MOVING_SPHERE: u32 = 1;
pub const RT_PRIM_RECT: u32 = 2;
pub const RT_OP_TRANSLATE: u32 = 0;
pub const RT_OP_ROTATE_Y: u32 = 1;
pub const RT_MAT_LAMBERTIAN: u32 = 0;
pub const RT_MAT_METAL: u32 = 1;
pub const RT_MAT_DIELECTRIC: u32 = 2;
pub const RT_MAT_DIFFUSE_LIGHT: u32 = 3;
pub const RT_MAT_ISOTROPIC: u32 = 4;
pub const RT_TEX_SOLID: u32 = 0;
pub const RT_TEX_CHECKER: u32 = 1;
pub const RT_TEX_PERLIN: u32 = 2;
pub const RT_TEX_IMAGE: u32 = 3;
pub const RT_FLAG_STATS: u32 = 1;
pub const RT_FLAG_TIMING: u32 = 2;
pub const RT_FLAG_BINNED: u32 = 4;
pub const RT_STAT_COUNT: usize = 24;
pub const RT_NOALLOC_SCRATCH_BYTES: u32 = 64 << 20;

/// `Camera` after `Builder::done` (src/camera.rs:6-15, :52-86).
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rt_camera {
    pub origin: [f64; 3],
    pub lower_left: [f64; 3],
    pub horizontal: [f64; 3],
    pub vertical: [f64; 3],
    pub u: [f64; 3],
    pub v: [f64; 3],
    pub lens_radius: f64,
    pub time0: f64,
    pub time1: f64,
}

/// What the camera builder was given (src/scene/desc.rs:149-159).
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rt_camera_desc {
    pub look_from: [f64; 3],
    pub look_at: [f64; 3],
    pub v_up: [f64; 3],
    pub vertical_fov: f64,
    pub aperture: f64,
    pub focus_distance: f64,
    pub shutter0: f64,
    pub shutter1: f64,
    pub has_look_at: u32,
    pub has_v_up: u32,
    pub has_focus_distance: u32,
    pub has_shutter: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rt_prim {
    pub kind: u32,
    pub axis: u32,
    pub material: u32,
    pub reserved: u32,
    pub center: [f64; 3],
    pub center1: [f64; 3],
    pub radius: f64,
    pub time0: f64,
    pub time1: f64,
    pub width: f64,
    pub height: f64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rt_op {
    pub kind: u32,
    pub reserved: u32,
    pub a: [f64; 3],
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rt_object {
    pub first_prim: u32,
    pub prim_count: u32,
    pub first_op: u32,
    pub op_count: u32,
    pub inner_first_op: u32,
    pub inner_op_count: u32,
    pub is_medium: u32,
    pub medium_material: u32,
    pub is_cuboid: u32,
    pub reserved: u32,
    pub neg_inv_density: f64,
    pub cuboid_center: [f64; 3],
    pub cuboid_half: [f64; 3],
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rt_material {
    pub kind: u32,
    pub texture: u32,
    pub albedo: [f64; 3],
    pub param: f64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rt_texture {
    pub kind: u32,
    pub a: u32,
    pub b: u32,
    pub reserved: u32,
    pub color: [f64; 3],
    pub scale: f64,
}

/// Tables of one Perlin instance (src/texture/perlin.rs:8-14).
#[repr(C)]
#[derive(Clone, Copy)]
pub struct rt_perlin {
    pub rand_vec: [[f64; 3]; 256],
    pub perm_x: [u32; 256],
    pub perm_y: [u32; 256],
    pub perm_z: [u32; 256],
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rt_image {
    pub width: u32,
    pub height: u32,
    pub offset: u64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct rt_scene {
    pub camera: rt_camera,
    pub background: [f64; 3],
    pub width: u32,
    pub height: u32,
    pub samples_per_pixel: u32,
    pub max_depth: u32,
    pub n_objects: u32,
    pub n_prims: u32,
    pub n_ops: u32,
    pub n_materials: u32,
    pub n_textures: u32,
    pub n_perlin: u32,
    pub n_images: u32,
    pub reserved: u32,
    pub n_texel_bytes: u64,
    pub objects: *const rt_object,
    pub prims: *const rt_prim,
    pub ops: *const rt_op,
    pub materials: *const rt_material,
    pub textures: *const rt_texture,
    pub perlin: *const rt_perlin,
    pub images: *const rt_image,
    pub texels: *const u8,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rt_host_options {
    pub scene_seed: u64,
    pub width: u32,
    pub height: u32,
    pub samples_per_pixel: u32,
    pub max_depth: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct rt_config {
    pub device: i32,
    pub n_devices: u32,
    pub max_paths_in_flight: u64,
    pub devices: *const i32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rt_window {
    pub x0: u32,
    pub y0: u32,
    pub w: u32,
    pub h: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rt_render_params {
    pub width: u32,
    pub height: u32,
    pub sample_begin: u32,
    pub sample_end: u32,
    pub max_depth: u32,
    pub flags: u32,
    pub seed: u64,
    pub window: rt_window,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rt_bvh_node {
    pub lo: [f32; 3],
    pub hi: [f32; 3],
    pub a: u32,
    pub b: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rt_bvh_item {
    pub object: u32,
    pub prim: u32,
    pub rank: u32,
    pub reserved: u32,
}

// The layouts the frozen-scene format and the device upload rely on (x86-64 / aarch64 LP64).
const _: () = assert!(size_of::<rt_camera>() == 168);
const _: () = assert!(size_of::<rt_camera_desc>() == 128);
const _: () = assert!(size_of::<rt_prim>() == 104);
const _: () = assert!(size_of::<rt_op>() == 32);
const _: () = assert!(size_of::<rt_object>() == 96);
const _: () = assert!(size_of::<rt_material>() == 40);
const _: () = assert!(size_of::<rt_texture>() == 48);
const _: () = assert!(size_of::<rt_perlin>() == 256 * 24 + 3 * 1024);
const _: () = assert!(size_of::<rt_image>() == 16);
const _: () = assert!(size_of::<rt_scene>() == 312);
const _: () = assert!(size_of::<rt_host_options>() == 24);
const _: () = assert!(size_of::<rt_config>() == 24);
const _: () = assert!(size_of::<rt_window>() == 16);
const _: () = assert!(size_of::<rt_render_params>() == 48);
const _: () = assert!(size_of::<rt_bvh_node>() == 32);
const _: () = assert!(size_of::<rt_bvh_item>() == 16);

/// `rt_reduce_path`: how a context adds its members' shares.
pub const RT_REDUCE_NONE: u32 = 0;
pub const RT_REDUCE_RCCL: u32 = 1;
pub const RT_REDUCE_HOST_STAGED: u32 = 2;

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rt_group_report {
    pub n_members: u32,
    pub reduce_path: u32,
    pub render_ms_min: f64,
    pub render_ms_max: f64,
    pub reduce_ms: f64,
}
const _: () = assert!(size_of::<rt_group_report>() == 32);

pub enum rt_ctx {}
pub enum rt_bvh {}
pub enum rt_host_scene {}

/// `void fn(void* user, uint64_t done, uint64_t total)`; `None` removes the callback.
pub type rt_progress_fn = Option<unsafe extern "C" fn(user: *mut c_void, done: u64, total: u64)>;

extern "C" {
    // ---- host front-end (libmi355rt_host.so) ----
    pub fn rt_host_add_asset_dir(dir: *const c_char) -> c_int;
    pub fn rt_host_load_ron(path: *const c_char, opt: *const rt_host_options, out: *mut *mut rt_host_scene) -> c_int;
    pub fn rt_host_load_ron_text(text: *const c_char, base_dir: *const c_char, opt: *const rt_host_options,
                                 out: *mut *mut rt_host_scene) -> c_int;
    pub fn rt_host_save_frozen(s: *const rt_host_scene, path: *const c_char) -> c_int;
    pub fn rt_host_load_frozen(path: *const c_char, out: *mut *mut rt_host_scene) -> c_int;
    pub fn rt_host_generate_cover(half_extent: u32, scene_seed: u64, width: u32, height: u32,
                                  samples_per_pixel: u32, max_depth: u32, out: *mut *mut rt_host_scene) -> c_int;
    pub fn rt_host_set_image(s: *mut rt_host_scene, width: u32, height: u32) -> c_int;
    pub fn rt_host_scene_view(s: *const rt_host_scene) -> *const rt_scene;
    pub fn rt_host_camera_desc(s: *const rt_host_scene) -> *const rt_camera_desc;
    pub fn rt_host_free(s: *mut rt_host_scene);
    pub fn rt_host_last_error() -> *const c_char;
    pub fn rt_host_resolve_srgb8(rgb_sum: *const f64, n_pixels: u64, samples_per_pixel: u32, rgb8_out: *mut u8);
    pub fn rt_host_resolve_srgb8_f32(rgb_sum: *const f32, n_pixels: u64, samples_per_pixel: u32, rgb8_out: *mut u8);
    pub fn rt_host_decode_jpeg(data: *const u8, size: u64, width: *mut u32, height: *mut u32, rgb8: *mut u8,
                               capacity: u64) -> c_int;
    pub fn rt_host_write_png(path: *const c_char, width: u32, height: u32, rgb8: *const u8) -> c_int;
    pub fn rt_host_human_duration(millis: u64, buf: *mut c_char, buf_len: usize) -> c_int;

    // ---- render path (libmi355rt.so) ----
    pub fn rt_create(cfg: *const rt_config, out: *mut *mut rt_ctx) -> c_int;
    pub fn rt_upload_scene(ctx: *mut rt_ctx, scene: *const rt_scene) -> c_int;
    pub fn rt_render(ctx: *mut rt_ctx, p: *const rt_render_params, rgb_sum: *mut f32, stats: *mut u64) -> c_int;
    pub fn rt_render_f64(ctx: *mut rt_ctx, p: *const rt_render_params, rgb_sum: *mut f64, stats: *mut u64) -> c_int;
    pub fn rt_render_device(ctx: *mut rt_ctx, p: *const rt_render_params, d_rgb_sum: *mut f32, stats: *mut u64) -> c_int;
    pub fn rt_render_device_noalloc(ctx: *mut rt_ctx, p: *const rt_render_params, d_rgb_sum: *mut f32) -> c_int;
    pub fn rt_set_progress(ctx: *mut rt_ctx, f: rt_progress_fn, user: *mut c_void) -> c_int;
    pub fn rt_group_info(ctx: *const rt_ctx, out: *mut rt_group_report) -> c_int;
    pub fn rt_first_hit(ctx: *mut rt_ctx, width: u32, height: u32, window: *const rt_window, id: *mut u32,
                        t: *mut f64, normal: *mut f64) -> c_int;
    pub fn rt_bvh_build(scene: *const rt_scene, out: *mut *mut rt_bvh) -> c_int;
    pub fn rt_bvh_info(bvh: *const rt_bvh, n_nodes: *mut u32, n_items: *mut u32, max_depth: *mut u32) -> c_int;
    pub fn rt_bvh_copy(bvh: *const rt_bvh, nodes: *mut rt_bvh_node, items: *mut rt_bvh_item) -> c_int;
    pub fn rt_bvh_free(bvh: *mut rt_bvh);
    pub fn rt_last_error(ctx: *const rt_ctx) -> *const c_char;
    pub fn rt_destroy(ctx: *mut rt_ctx);
}
