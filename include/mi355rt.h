/* mi355rt.h — C ABI of the MI355X path-tracing backend.
 *
 * This header is the drop-in boundary for the render loop of mythmon/raytracer.
 * The reference has no FFI of its own; the seam is structural: src/main.rs:49-55
 * produces `Scene{world, camera, image, background}` (src/config.rs:4-9) and
 * src/main.rs:65-101 consumes it to fill a w*h raster.  Everything the reference
 * expresses as trait objects (Hittable src/hittable/mod.rs:26-30, Material
 * src/material/mod.rs:16-27, Texture src/texture/mod.rs:16-19) crosses this
 * boundary as plain data: flat arrays of PODs, no callbacks, no C++ types.
 *
 * Two libraries implement it:
 *   libmi355rt_host.so  (g++)    rt_host_*  : scene front-end (RON files, pattern
 *                                 expansion, camera builder, Perlin tables, frozen
 *                                 scene files, resolve to 8 bit, PNG).
 *   libmi355rt.so       (hipcc)  rt_*       : BVH build + the HIP render path.
 *
 * All functions return 0 on success and a negative rt_status otherwise; the
 * message is available from rt_last_error()/rt_host_last_error().  No C++
 * exception crosses the boundary.  A context is used by one host thread at a
 * time; calls are synchronous.  The caller owns every pointer it passes in; the
 * library copies during rt_upload_scene and keeps no host pointer afterwards.
 */
#ifndef MI355RT_H
#define MI355RT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum rt_status {
  RT_OK = 0,
  RT_ERR_INVALID = -1,     /* bad argument / malformed scene */
  RT_ERR_DEVICE = -2,      /* HIP runtime error */
  RT_ERR_NOMEM = -3,
  RT_ERR_IO = -4,
  RT_ERR_PARSE = -5,       /* scene file syntax / semantics */
  RT_ERR_UNSUPPORTED = -6, /* valid in the reference, not representable here */
  RT_ERR_LIMIT = -7        /* exceeds a stated maximum */
} rt_status;

/* ------------------------------------------------------------------ scene */

/* Camera constants as computed once by Builder::done (src/camera.rs:52-86). */
typedef struct rt_camera {
  double origin[3];
  double lower_left[3];
  double horizontal[3];
  double vertical[3];
  double u[3];
  double v[3];
  double lens_radius;
  double time0, time1; /* shutter_time range (src/camera.rs:14) */
} rt_camera;

/* What the camera builder was given (src/scene/desc.rs:149-159); kept so the
 * camera can be rebuilt when the image size (hence aspect ratio) is overridden. */
typedef struct rt_camera_desc {
  double look_from[3];
  double look_at[3];
  double v_up[3];
  double vertical_fov;
  double aperture;
  double focus_distance; /* valid iff has_focus_distance */
  double shutter0, shutter1;
  uint32_t has_look_at, has_v_up, has_focus_distance, has_shutter;
} rt_camera_desc;

typedef enum rt_prim_kind {
  RT_PRIM_SPHERE = 0,        /* src/hittable/sphere.rs */
  RT_PRIM_MOVING_SPHERE = 1, /* src/hittable/moving_sphere.rs */
  RT_PRIM_RECT = 2           /* src/hittable/rect.rs */
} rt_prim_kind;

/* One base primitive.  A Cuboid (src/hittable/cuboid.rs:18-37) is six
 * consecutive RT_PRIM_RECT entries in the reference's side order. */
typedef struct rt_prim {
  uint32_t kind;     /* rt_prim_kind */
  uint32_t axis;     /* rect: 0=X 1=Y 2=Z */
  uint32_t material; /* index into materials[] */
  uint32_t reserved;
  double center[3];  /* sphere centre / moving-sphere centre.start / rect centre */
  double center1[3]; /* moving sphere: centre.end */
  double radius;
  double time0, time1; /* moving sphere: time range */
  double width, height; /* rect */
} rt_prim;

typedef enum rt_op_kind {
  RT_OP_TRANSLATE = 0, /* src/hittable/instance.rs:5-26  a = offset */
  RT_OP_ROTATE_Y = 1   /* src/hittable/instance.rs:28-103 a[0]=sin, a[1]=cos */
} rt_op_kind;

typedef struct rt_op {
  uint32_t kind;
  uint32_t reserved;
  double a[3];
} rt_op;

/* One top-level hittable, in the order the reference's loader appends them
 * (src/scene/loader.rs:95-189).  Its index in objects[] is the object id
 * reported by rt_first_hit.  Structure, outermost first:
 *   ops[first_op .. +op_count)              Translate / RotateY wrappers
 *   [ConstantMedium]                        iff is_medium
 *   ops[inner_first_op .. +inner_op_count)  wrappers between medium and base
 *   prims[first_prim .. +prim_count)        1 primitive, or the 6 cuboid sides
 */
typedef struct rt_object {
  uint32_t first_prim, prim_count;
  uint32_t first_op, op_count;
  uint32_t inner_first_op, inner_op_count;
  uint32_t is_medium;
  uint32_t medium_material; /* isotropic material (src/hittable/constant_medium.rs:21) */
  uint32_t is_cuboid;
  uint32_t reserved;
  double neg_inv_density;   /* src/hittable/constant_medium.rs:20 */
  double cuboid_center[3];  /* Cuboid::bounding_box uses center +- half (cuboid.rs:45-47) */
  double cuboid_half[3];
} rt_object;

typedef enum rt_material_kind {
  RT_MAT_LAMBERTIAN = 0,    /* src/material/lambertian.rs  texture */
  RT_MAT_METAL = 1,         /* src/material/metal.rs       albedo, param=fuzziness */
  RT_MAT_DIELECTRIC = 2,    /* src/material/dielectric.rs  param=index_of_refraction */
  RT_MAT_DIFFUSE_LIGHT = 3, /* src/material/emissive.rs    texture */
  RT_MAT_ISOTROPIC = 4      /* src/texture/isotropic.rs    texture */
} rt_material_kind;
#define RT_MAT_KINDS 5

typedef struct rt_material {
  uint32_t kind;
  uint32_t texture;
  double albedo[3];
  double param;
} rt_material;

typedef enum rt_texture_kind {
  RT_TEX_SOLID = 0,   /* src/texture/solid_color.rs   color */
  RT_TEX_CHECKER = 1, /* src/texture/checkerboard.rs  a=even b=odd (texture indices) */
  RT_TEX_PERLIN = 2,  /* src/texture/perlin.rs        a=index into perlin[], scale */
  RT_TEX_IMAGE = 3    /* src/texture/image.rs         a=index into images[] */
} rt_texture_kind;

typedef struct rt_texture {
  uint32_t kind;
  uint32_t a, b;
  uint32_t reserved;
  double color[3];
  double scale;
} rt_texture;

/* Tables of one Perlin instance (src/texture/perlin.rs:8-14). */
typedef struct rt_perlin {
  double rand_vec[256][3];
  uint32_t perm_x[256];
  uint32_t perm_y[256];
  uint32_t perm_z[256];
} rt_perlin;

/* One decoded RGB8 image, row-major, row 0 on top, at texels[offset]. */
typedef struct rt_image {
  uint32_t width, height;
  uint64_t offset;
} rt_image;

typedef struct rt_scene {
  rt_camera camera;
  double background[3];
  /* image block of the scene file (src/config.rs:11-17) */
  uint32_t width, height, samples_per_pixel, max_depth;
  uint32_t n_objects, n_prims, n_ops, n_materials, n_textures, n_perlin, n_images;
  uint32_t reserved;
  uint64_t n_texel_bytes;
  const rt_object* objects;
  const rt_prim* prims;
  const rt_op* ops;
  const rt_material* materials;
  const rt_texture* textures;
  const rt_perlin* perlin;
  const rt_image* images;
  const uint8_t* texels;
} rt_scene;

/* ------------------------------------------------------- host front-end */

typedef struct rt_host_scene rt_host_scene;

typedef struct rt_host_options {
  uint64_t scene_seed;   /* seeds every load-time random draw (desc.rs:137-140,
                            loader.rs:341-352, perlin.rs:17-36) */
  uint32_t width, height;          /* 0 = keep the file's value */
  uint32_t samples_per_pixel;      /* 0 = keep */
  uint32_t max_depth;              /* 0 = keep */
} rt_host_options;

/* Image("x.jpg") textures (src/texture/image.rs:12-20): the file itself when it is a
 * baseline JPEG (next to the scene file, then in every directory registered here),
 * else a binary PPM (P6) of the same stem in the same places. */
int rt_host_add_asset_dir(const char* dir);
/* Replaces SceneLoader::new(path).load() (src/scene/loader.rs:39-93). */
int rt_host_load_ron(const char* path, const rt_host_options* opt, rt_host_scene** out);
/* Same, from text already in memory; `base_dir` resolves Image("...") paths. */
int rt_host_load_ron_text(const char* text, const char* base_dir, const rt_host_options* opt,
                          rt_host_scene** out);
/* Frozen (fully expanded) scene files: this build's own binary format. */
int rt_host_save_frozen(const rt_host_scene* s, const char* path);
int rt_host_load_frozen(const char* path, rt_host_scene** out);
/* book_cover.ron:26-82 scaled up: ground + (2*half_extent)^2 small spheres + 3
 * big ones, camera/background as book_cover.ron:9-17. */
int rt_host_generate_cover(uint32_t half_extent, uint64_t scene_seed, uint32_t width,
                           uint32_t height, uint32_t samples_per_pixel, uint32_t max_depth,
                           rt_host_scene** out);
/* Change the image size and rebuild the camera (aspect = width/height,
 * src/scene/loader.rs:66). */
int rt_host_set_image(rt_host_scene* s, uint32_t width, uint32_t height);
const rt_scene* rt_host_scene_view(const rt_host_scene* s);
const rt_camera_desc* rt_host_camera_desc(const rt_host_scene* s);
void rt_host_free(rt_host_scene* s);
const char* rt_host_last_error(void);

/* Color::into_srgb8 (src/geom/vec3.rs:123-133) over n pixels of RGB sums. */
void rt_host_resolve_srgb8(const double* rgb_sum, uint64_t n_pixels, uint32_t samples_per_pixel,
                           uint8_t* rgb8_out);
void rt_host_resolve_srgb8_f32(const float* rgb_sum, uint64_t n_pixels,
                               uint32_t samples_per_pixel, uint8_t* rgb8_out);
/* RGB8 PNG, rows top to bottom (src/main.rs:103-112). */
/* Baseline JPEG (8-bit Huffman, grey or YCbCr) to RGB8: the decoder behind Image("x.jpg")
 * textures (src/texture/image.rs:12-20).  Call with rgb8 == NULL to learn the size;
 * `capacity` is the size of rgb8 in bytes (width * height * 3 needed). */
int rt_host_decode_jpeg(const uint8_t* data, uint64_t size, uint32_t* width, uint32_t* height,
                        uint8_t* rgb8, uint64_t capacity);
int rt_host_write_png(const char* path, uint32_t width, uint32_t height, const uint8_t* rgb8);
/* "Done in ..." duration text (src/main.rs:146-165); returns bytes written. */
int rt_host_human_duration(uint64_t millis, char* buf, size_t buf_len);

/* ------------------------------------------------------------ render path */

typedef struct rt_ctx rt_ctx;

typedef struct rt_config {
  int32_t device;            /* HIP device ordinal (n_devices <= 1 and devices == NULL) */
  uint32_t n_devices;        /* > 1: a device group, see below; 0 or 1: one device */
  uint64_t max_paths_in_flight; /* 0 = default; bounds device memory for path state, per device */
  /* Device group: the HIP ordinals of its members; devices[0] is the root.  The context then
   * renders with all of them: rt_upload_scene copies the scene to each, rt_render and
   * rt_render_device give member k of N the samples [begin + n k / N, begin + n (k + 1) / N) of every
   * pixel (the pixel-sample space of src/main.rs:87-97 shards by sample range) and add the
   * float shares onto the root with one reduce(sum) over RCCL communicators the context owns
   * (ncclCommInitAll in this process).  rt_render_device's buffer lives on the root's device.
   * rt_first_hit runs on the root; rt_render_f64 and rt_render_device_noalloc are not for groups.
   * A group of distinct devices needs librccl (looked up at run time): without it rt_create fails,
   * unless MI355RT_STAGED_REDUCE=1 asks for the shares to be added through host memory instead.
   * A group that names one device more than once cannot have communicators and always adds
   * through host memory (a rehearsal of the group logic on a one-GPU machine).  n_devices == 1
   * with MI355RT_FORCE_RCCL=1 makes a group of one member with a one-rank communicator, so that
   * the whole RCCL path (lookup, ncclCommInitAll, ncclGroupStart/End, ncclReduce on the member's
   * stream) can run on a machine with a single GPU.  rt_group_info says which path a context uses. */
  const int32_t* devices;
} rt_config;

typedef struct rt_window { /* sub-rectangle of the image; w==0 means the whole image */
  uint32_t x0, y0, w, h;
} rt_window;

#define RT_FLAG_STATS 1u       /* fill the operation counts of the stats array (instrumented kernels) */
#define RT_FLAG_BINNED 4u      /* always run the extend / bin-by-material / shade kernels; without it,
                                  scenes with only SolidColor textures use one fused kernel per bounce
                                  (RT_STAT_EXTEND_NS then covers it and RT_STAT_SHADE_NS is 0) */
#define RT_FLAG_TIMING 2u      /* fill the RT_STAT_*_NS / *_LAUNCHES slots: device time of the extend
                                  and shade kernels, from events recorded around every launch */

typedef struct rt_render_params {
  uint32_t width, height;      /* full image size: sets u,v (src/main.rs:91-93) and RNG pixel index */
  uint32_t sample_begin, sample_end; /* render samples [begin,end) of every pixel */
  uint32_t max_depth;          /* src/main.rs:95 */
  uint32_t flags;
  uint64_t seed;               /* sampling seed: Philox key */
  rt_window window;            /* output buffers cover only this window */
} rt_render_params;

/* Indices into the stats array filled under RT_FLAG_STATS. */
enum {
  RT_STAT_RAYS = 0,        /* rays traced (extend invocations) */
  RT_STAT_BOX_TESTS,       /* node box tests */
  RT_STAT_SPHERE_TESTS,
  RT_STAT_MSPHERE_TESTS,
  RT_STAT_RECT_TESTS,
  RT_STAT_MEDIUM_TESTS,
  RT_STAT_OP_TRANSLATE,    /* instance ops applied to rays */
  RT_STAT_OP_ROTATE,
  RT_STAT_HITS,
  RT_STAT_SHADE_MISS,
  RT_STAT_SHADE_LAMBERTIAN,
  RT_STAT_SHADE_METAL,
  RT_STAT_SHADE_DIELECTRIC,
  RT_STAT_SHADE_LIGHT,
  RT_STAT_SHADE_ISOTROPIC,
  RT_STAT_TEX_CHECKER,
  RT_STAT_TEX_PERLIN,
  RT_STAT_TEX_IMAGE,
  RT_STAT_SPHERE_ROOTS,    /* sphere tests whose discriminant was >= 0 */
  RT_STAT_NODE_VISITS,     /* interior records visited (each tests its 2-4 children) */
  RT_STAT_EXTEND_NS = 20,  /* RT_FLAG_TIMING: summed device time of the extend launches */
  RT_STAT_EXTEND_LAUNCHES = 21,
  RT_STAT_SHADE_NS = 22,
  RT_STAT_SHADE_LAUNCHES = 23,
  RT_STAT_COUNT = 24
};

/* Replaces nothing in the reference one-to-one; plays the part of
 * `let Scene{..} = loader.load()?` handing the world to the render closure
 * (src/main.rs:50-55,77-80). */
int rt_create(const rt_config* cfg, rt_ctx** out);
/* Copies the scene and builds the acceleration structure: on the host below 2^17 primitives, on
 * the device above (environment MI355RT_DEVICE_BUILD=0 / 1 forces either).  What a render
 * returns does not depend on where or how the hierarchy was built: the reference's own BVH
 * (src/hittable/bvh.rs:33-77) only decides between hits at exactly the same t, and that order
 * is carried by per-slot ranks. */
int rt_upload_scene(rt_ctx* ctx, const rt_scene* scene);

/* Replaces the render closure src/main.rs:82-101 minus the 8-bit resolve:
 * for every pixel of the window, the sum over samples [sample_begin,sample_end)
 * of ray_color (src/main.rs:120-140), in sample order (src/main.rs:97).
 * rgb_sum is a HOST buffer of window.w*window.h*3 values, row 0 on top. */
int rt_render(rt_ctx* ctx, const rt_render_params* p, float* rgb_sum, uint64_t* stats);
int rt_render_f64(rt_ctx* ctx, const rt_render_params* p, double* rgb_sum, uint64_t* stats);
/* Same, but rgb_sum is a DEVICE buffer on the context's device (e.g. a tensor a
 * collective will reduce); the call returns after the device work completed. */
int rt_render_device(rt_ctx* ctx, const rt_render_params* p, float* d_rgb_sum, uint64_t* stats);

/* No counterpart in the reference.  For callers that cannot allocate device memory while
 * they render (tight serving loops, harnesses that forbid hipMalloc, code that sets up a HIP
 * graph capture elsewhere and must not disturb the allocator): the same sums as rt_render_device, with every intermediate buffer (path state,
 * live-path queues, counters, the f64 sums) in a fixed RT_NOALLOC_SCRATCH_BYTES array that is
 * part of the library's code object, one copy per GPU.  The call allocates and frees nothing.
 * It is NOT itself capturable into a HIP graph: like rt_render_device it synchronises its stream
 * and reads back live-path counts between launches to size the next ones.
 * Limits: a window whose single-sample pass does not fit the array is refused with
 * RT_ERR_LIMIT (about 2^18 pixels at depth 50); no statistics or launch timing.  Every call on
 * a GPU shares that GPU's array, whichever context makes it: calls that overlap on one GPU
 * race, and the array keeps the last call's contents between calls (each call clears what it
 * needs at zero). */
#define RT_NOALLOC_SCRATCH_BYTES (64u << 20)
int rt_render_device_noalloc(rt_ctx* ctx, const rt_render_params* p, float* d_rgb_sum);

/* Replaces the progress bar of src/main.rs:58-61,83,99 (one tick per finished pixel there).
 * While a callback is set, every render call on `ctx` calls fn(user, done, total) after each
 * finished pass over the window: done and total count pixel-samples of that call, done grows
 * to total.  Single context: called from the thread that renders.  Device group: from the
 * threads that drive the members, never two at a time.  The render waits for a pass to
 * finish before it queues the next one, which costs a little; fn == NULL removes the
 * callback.  The callback must not call into the library with the same context. */
typedef void (*rt_progress_fn)(void* user, uint64_t done, uint64_t total);
int rt_set_progress(rt_ctx* ctx, rt_progress_fn fn, void* user);

/* Deterministic parity pass: pinhole rays through pixel centres
 * (u=(x+0.5)/(w-1), v=((h-y)+0.5)/(h-1), lens offset 0, time = shutter start),
 * world.hit(ray, 0.001..inf) (src/main.rs:123) with constant media left out.
 * id = object index or 0xFFFFFFFF on a miss; t and normal are HitRecord.t and
 * HitRecord.normal (src/hittable/mod.rs:32-40).  Host buffers, window-sized. */
int rt_first_hit(rt_ctx* ctx, uint32_t width, uint32_t height, const rt_window* window,
                 uint32_t* id, double* t, double* normal);

/* The flattened acceleration structure the device walks, for inspection and
 * for instrumented CPU traversal.  nodes: 32-byte records
 * {float lo[3]; float hi[3]; uint32 a; uint32 b}, record 0 is the root.
 * Interior: a = first child, b = 0x80000000 | (children - 1) with 2..4 children
 * stored consecutively at a..; leaf: a = first item, b = item count (1..8).
 * Records are numbered breadth-first.  items: for each leaf slot the index of
 * the top-level object and the primitive inside it (or 0xFFFFFFFF when the slot
 * is the whole object: a constant medium). */
typedef struct rt_bvh_node {
  float lo[3];
  float hi[3];
  uint32_t a, b;
} rt_bvh_node;
typedef struct rt_bvh_item {
  uint32_t object;
  uint32_t prim;  /* absolute index into rt_scene.prims, or 0xFFFFFFFF */
  uint32_t rank;  /* tie-break rank: position in the reference's resolution order */
  uint32_t reserved;
} rt_bvh_item;
typedef struct rt_bvh rt_bvh;
/* Host-only (needs no device): the same build rt_upload_scene performs. */
int rt_bvh_build(const rt_scene* scene, rt_bvh** out);
int rt_bvh_info(const rt_bvh* bvh, uint32_t* n_nodes, uint32_t* n_items, uint32_t* max_depth);
int rt_bvh_copy(const rt_bvh* bvh, rt_bvh_node* nodes, rt_bvh_item* items);
void rt_bvh_free(rt_bvh* bvh);

/* How a context adds the shares of its members (the cross-device form of the per-pixel
 * `.sum()` over samples, src/main.rs:97), and what the last render call on it spent where. */
typedef enum rt_reduce_path {
  RT_REDUCE_NONE = 0,        /* one device: nothing to add */
  RT_REDUCE_RCCL = 1,        /* one ncclReduce(sum, float32) to the root over the group's communicators */
  RT_REDUCE_HOST_STAGED = 2  /* shares copied to the host, added there, the sums copied back */
} rt_reduce_path;
typedef struct rt_group_report {
  uint32_t n_members;      /* 1 for a plain context */
  uint32_t reduce_path;    /* rt_reduce_path */
  double render_ms_min;    /* last render call: the fastest and the slowest member's share, host clock */
  double render_ms_max;
  double reduce_ms;        /* last render call: from the last member's end to the sums on the root */
} rt_group_report;
int rt_group_info(const rt_ctx* ctx, rt_group_report* out);

const char* rt_last_error(const rt_ctx* ctx);
void rt_destroy(rt_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* MI355RT_H */
