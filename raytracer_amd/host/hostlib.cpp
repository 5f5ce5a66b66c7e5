// C ABI of the host front-end (rt_host_* in include/mi355rt.h).
#include <zlib.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <memory>
#include <string>

#include "../csrc/scene_validate.h"
#include "host_scene.h"
#include "jpeg.h"

namespace {

thread_local std::string g_error;

int set_error(int code, const std::string& msg) {
  g_error = msg;
  return code;
}

int code_for(const std::string& msg) {
  if (msg.rfind("unsupported", 0) == 0) return RT_ERR_UNSUPPORTED;
  return RT_ERR_PARSE;
}

const char kMagic[8] = {'M', 'I', '3', '5', '5', 'R', 'T', 'S'};
const uint32_t kVersion = 1;

struct FrozenHeader {
  char magic[8];
  uint32_t version;
  uint32_t sizeof_object, sizeof_prim, sizeof_op, sizeof_material, sizeof_texture, sizeof_perlin,
      sizeof_image;
  rt_camera_desc camera_desc;
  rt_camera camera;
  double background[3];
  uint32_t width, height, samples_per_pixel, max_depth;
  uint32_t n_objects, n_prims, n_ops, n_materials, n_textures, n_perlin, n_images, reserved;
  uint64_t n_texel_bytes;
};

// Frozen scenes are written through zlib's gz layer (gzread also accepts an
// uncompressed file).
bool gz_write(gzFile f, const void* data, size_t bytes) {
  const char* p = (const char*)data;
  while (bytes) {
    unsigned chunk = bytes > (1u << 30) ? (1u << 30) : (unsigned)bytes;
    int n = gzwrite(f, p, chunk);
    if (n <= 0) return false;
    p += n;
    bytes -= (size_t)n;
  }
  return true;
}
bool gz_read(gzFile f, void* data, size_t bytes) {
  char* p = (char*)data;
  while (bytes) {
    unsigned chunk = bytes > (1u << 30) ? (1u << 30) : (unsigned)bytes;
    int n = gzread(f, p, chunk);
    if (n <= 0) return false;
    p += n;
    bytes -= (size_t)n;
  }
  return true;
}
template <typename T>
bool write_vec(gzFile f, const std::vector<T>& v) {
  return v.empty() || gz_write(f, v.data(), v.size() * sizeof(T));
}
// n records in pieces of at most 16 MiB, the vector growing as they arrive: a count that the
// file does not back (a corrupted or crafted header) ends in "truncated", not in an allocation
// of whatever the header claims.
template <typename T>
bool read_vec(gzFile f, std::vector<T>& v, size_t n) {
  const size_t piece = std::max<size_t>(1, ((size_t)16 << 20) / sizeof(T));
  v.clear();
  while (v.size() < n) {
    const size_t at = v.size(), take = std::min(piece, n - at);
    v.resize(at + take);
    if (!gz_read(f, v.data() + at, take * sizeof(T))) return false;
  }
  return true;
}

}  // namespace

extern "C" {

const char* rt_host_last_error(void) { return g_error.c_str(); }

int rt_host_add_asset_dir(const char* dir) {
  if (!dir) return set_error(RT_ERR_INVALID, "null directory");
  rthost::AssetDirs::list().push_back(dir);
  return RT_OK;
}

int rt_host_load_ron_text(const char* text, const char* base_dir, const rt_host_options* opt,
                          rt_host_scene** out) {
  if (!text || !out) return set_error(RT_ERR_INVALID, "null argument");
  rt_host_options o{};
  if (opt) o = *opt;
  try {
    *out = rthost::load_ron_text(text, base_dir ? base_dir : "", o);
  } catch (const std::exception& e) {
    *out = nullptr;
    return set_error(code_for(e.what()), e.what());
  }
  return RT_OK;
}

int rt_host_load_ron(const char* path, const rt_host_options* opt, rt_host_scene** out) {
  if (!path || !out) return set_error(RT_ERR_INVALID, "null argument");
  std::ifstream f(path, std::ios::binary);
  if (!f) {
    *out = nullptr;
    return set_error(RT_ERR_IO, std::string("opening scene file: ") + path);
  }
  std::stringstream ss;
  ss << f.rdbuf();
  std::string dir(path);
  size_t slash = dir.find_last_of('/');
  dir = slash == std::string::npos ? std::string(".") : dir.substr(0, slash);
  return rt_host_load_ron_text(ss.str().c_str(), dir.c_str(), opt, out);
}

int rt_host_save_frozen(const rt_host_scene* s, const char* path) {
  if (!s || !path) return set_error(RT_ERR_INVALID, "null argument");
  gzFile f = gzopen(path, "wb9");
  if (!f) return set_error(RT_ERR_IO, std::string("cannot create ") + path);
  FrozenHeader h{};
  std::memcpy(h.magic, kMagic, 8);
  h.version = kVersion;
  h.sizeof_object = sizeof(rt_object);
  h.sizeof_prim = sizeof(rt_prim);
  h.sizeof_op = sizeof(rt_op);
  h.sizeof_material = sizeof(rt_material);
  h.sizeof_texture = sizeof(rt_texture);
  h.sizeof_perlin = sizeof(rt_perlin);
  h.sizeof_image = sizeof(rt_image);
  h.camera_desc = s->camera_desc;
  h.camera = s->view.camera;
  std::memcpy(h.background, s->view.background, sizeof h.background);
  h.width = s->view.width;
  h.height = s->view.height;
  h.samples_per_pixel = s->view.samples_per_pixel;
  h.max_depth = s->view.max_depth;
  h.n_objects = (uint32_t)s->objects.size();
  h.n_prims = (uint32_t)s->prims.size();
  h.n_ops = (uint32_t)s->ops.size();
  h.n_materials = (uint32_t)s->materials.size();
  h.n_textures = (uint32_t)s->textures.size();
  h.n_perlin = (uint32_t)s->perlin.size();
  h.n_images = (uint32_t)s->images.size();
  h.n_texel_bytes = s->texels.size();
  bool ok = gz_write(f, &h, sizeof h) && write_vec(f, s->objects) && write_vec(f, s->prims) && write_vec(f, s->ops) &&
            write_vec(f, s->materials) && write_vec(f, s->textures) && write_vec(f, s->perlin) &&
            write_vec(f, s->images) && write_vec(f, s->texels);
  ok = (gzclose(f) == Z_OK) && ok;
  if (!ok) return set_error(RT_ERR_IO, std::string("short write to ") + path);
  return RT_OK;
}

int rt_host_load_frozen(const char* path, rt_host_scene** out) {
  if (!path || !out) return set_error(RT_ERR_INVALID, "null argument");
  *out = nullptr;
  gzFile f = gzopen(path, "rb");
  if (!f) return set_error(RT_ERR_IO, std::string("opening scene file: ") + path);
  struct Closer {
    gzFile f;
    ~Closer() { gzclose(f); }
  } closer{f};
  FrozenHeader h{};
  if (!gz_read(f, &h, sizeof h) || std::memcmp(h.magic, kMagic, 8) != 0 || h.version != kVersion)
    return set_error(RT_ERR_PARSE, std::string(path) + ": not a frozen scene of this version");
  if (h.sizeof_object != sizeof(rt_object) || h.sizeof_prim != sizeof(rt_prim) ||
      h.sizeof_op != sizeof(rt_op) || h.sizeof_material != sizeof(rt_material) ||
      h.sizeof_texture != sizeof(rt_texture) || h.sizeof_perlin != sizeof(rt_perlin) ||
      h.sizeof_image != sizeof(rt_image))
    return set_error(RT_ERR_PARSE, std::string(path) + ": record sizes do not match this build");
  // (nothing thrown here may cross the C boundary)
  try {
    std::unique_ptr<rt_host_scene> s(new rt_host_scene());
    s->camera_desc = h.camera_desc;
    s->view.camera = h.camera;
    std::memcpy(s->view.background, h.background, sizeof h.background);
    s->view.width = h.width;
    s->view.height = h.height;
    s->view.samples_per_pixel = h.samples_per_pixel;
    s->view.max_depth = h.max_depth;
    bool ok = read_vec(f, s->objects, h.n_objects) && read_vec(f, s->prims, h.n_prims) &&
              read_vec(f, s->ops, h.n_ops) && read_vec(f, s->materials, h.n_materials) &&
              read_vec(f, s->textures, h.n_textures) && read_vec(f, s->perlin, h.n_perlin) &&
              read_vec(f, s->images, h.n_images) && read_vec(f, s->texels, h.n_texel_bytes);
    if (!ok) return set_error(RT_ERR_IO, std::string(path) + ": truncated frozen scene");
    s->refresh_view();
    std::string bad = rt::validate_scene(s->view);
    if (!bad.empty()) return set_error(RT_ERR_PARSE, std::string(path) + ": invalid " + bad);
    *out = s.release();
    return RT_OK;
  } catch (const std::bad_alloc&) {
    return set_error(RT_ERR_NOMEM, std::string(path) + ": out of memory");
  } catch (const std::exception& e) {
    return set_error(RT_ERR_PARSE, std::string(path) + ": " + e.what());
  }
}

int rt_host_generate_cover(uint32_t half_extent, uint64_t scene_seed, uint32_t width,
                           uint32_t height, uint32_t samples_per_pixel, uint32_t max_depth,
                           rt_host_scene** out) {
  if (!out) return set_error(RT_ERR_INVALID, "null argument");
  *out = nullptr;
  if (width < 2 || height < 2 || half_extent == 0 || half_extent > 4096)
    return set_error(RT_ERR_INVALID, "generate_cover: bad size");
  auto* s = new rt_host_scene();
  try {
    rthost::LoadRng rng(scene_seed);
    auto solid = [&](double r, double g, double b) {
      rt_texture t{};
      t.kind = RT_TEX_SOLID;
      t.color[0] = r;
      t.color[1] = g;
      t.color[2] = b;
      s->textures.push_back(t);
      return (uint32_t)s->textures.size() - 1;
    };
    auto lambert = [&](double r, double g, double b) {
      rt_material m{};
      m.kind = RT_MAT_LAMBERTIAN;
      m.texture = solid(r, g, b);
      s->materials.push_back(m);
      return (uint32_t)s->materials.size() - 1;
    };
    auto sphere = [&](double x, double y, double z, double r, uint32_t mat) {
      rt_object o{};
      o.first_prim = (uint32_t)s->prims.size();
      o.prim_count = 1;
      rt_prim p{};
      p.kind = RT_PRIM_SPHERE;
      p.center[0] = x;
      p.center[1] = y;
      p.center[2] = z;
      p.radius = r;
      p.material = mat;
      s->prims.push_back(p);
      s->objects.push_back(o);
    };
    // shared materials, as scenes/book_cover.ron:19-24
    uint32_t ground = lambert(0.7, 0.7, 0.7);
    rt_material glass{};
    glass.kind = RT_MAT_DIELECTRIC;
    glass.param = 1.5;
    s->materials.push_back(glass);
    uint32_t glass_i = (uint32_t)s->materials.size() - 1;
    uint32_t big_diffuse = lambert(0.4, 0.2, 0.1);
    rt_material bm{};
    bm.kind = RT_MAT_METAL;
    bm.albedo[0] = 0.7;
    bm.albedo[1] = 0.6;
    bm.albedo[2] = 0.5;
    bm.param = 0.03;
    s->materials.push_back(bm);
    uint32_t big_metal = (uint32_t)s->materials.size() - 1;

    sphere(0.0, -1000.0, 0.0, 1000.0, ground);
    const int n = (int)half_extent;
    for (int a = -n; a < n; ++a) {
      for (int b = -n; b < n; ++b) {
        // evaluation order of book_cover.ron:40-62: centre x, centre z, then the material
        double cx = (double)a + rng.range(-0.4, 0.4);
        double cz = (double)b + rng.range(-0.4, 0.4);
        double chosen = rng.range(0.0, 20.0);  // weights 16 : 3 : 1
        uint32_t mat;
        if (chosen < 16.0) {
          double r = rng.range(0.0, 1.0), g = rng.range(0.0, 1.0), bl = rng.range(0.0, 1.0);
          mat = lambert(r, g, bl);
        } else if (chosen < 19.0) {
          rt_material m{};
          m.kind = RT_MAT_METAL;
          for (int k = 0; k < 3; ++k) m.albedo[k] = rng.range(0.0, 1.0);
          m.param = rng.range(0.0, 0.5);
          s->materials.push_back(m);
          mat = (uint32_t)s->materials.size() - 1;
        } else {
          mat = glass_i;
        }
        sphere(cx, 0.2, cz, 0.2, mat);
      }
    }
    sphere(0.0, 1.0, 0.0, 1.0, glass_i);
    sphere(-4.0, 1.0, 0.0, 1.0, big_diffuse);
    sphere(4.0, 1.0, 0.0, 1.0, big_metal);

    rt_camera_desc& cd = s->camera_desc;
    cd.look_from[0] = 13.0;
    cd.look_from[1] = 2.0;
    cd.look_from[2] = 3.0;
    cd.v_up[1] = 1.0;
    cd.has_v_up = 1;
    cd.vertical_fov = 18.0;
    cd.aperture = 0.02;
    cd.focus_distance = 10.0;
    cd.has_focus_distance = 1;
    s->view.width = width;
    s->view.height = height;
    s->view.samples_per_pixel = samples_per_pixel;
    s->view.max_depth = max_depth;
    s->view.background[0] = 0.7;
    s->view.background[1] = 0.8;
    s->view.background[2] = 1.0;
    s->view.camera = rthost::build_camera(cd, width, height);
    s->refresh_view();
  } catch (const std::exception& e) {
    delete s;
    return set_error(RT_ERR_INVALID, e.what());
  }
  *out = s;
  return RT_OK;
}

int rt_host_set_image(rt_host_scene* s, uint32_t width, uint32_t height) {
  if (!s) return set_error(RT_ERR_INVALID, "null scene");
  if (width < 2 || height < 2) return set_error(RT_ERR_INVALID, "image must be at least 2x2");
  try {
    s->view.camera = rthost::build_camera(s->camera_desc, width, height);
  } catch (const std::exception& e) {
    return set_error(RT_ERR_INVALID, e.what());
  }
  s->view.width = width;
  s->view.height = height;
  return RT_OK;
}

const rt_scene* rt_host_scene_view(const rt_host_scene* s) { return s ? &s->view : nullptr; }
const rt_camera_desc* rt_host_camera_desc(const rt_host_scene* s) {
  return s ? &s->camera_desc : nullptr;
}
void rt_host_free(rt_host_scene* s) { delete s; }

// Color::into_srgb8 (src/geom/vec3.rs:123-133).  Rust's f64::clamp keeps NaN and
// `as u8` saturates with NaN -> 0.
static inline uint8_t to_u8(double c, double scale) {
  const double max = 255.0 / 256.0;
  double v = c * scale;
  if (v < 0.0) v = 0.0;
  if (v > max) v = max;
  double q = v * 256.0;
  if (!(q == q)) return 0;
  if (q <= 0.0) return 0;
  if (q >= 255.0) return 255;
  return (uint8_t)q;
}

void rt_host_resolve_srgb8(const double* rgb_sum, uint64_t n_pixels, uint32_t samples_per_pixel,
                           uint8_t* rgb8_out) {
  double scale = 1.0 / (double)samples_per_pixel;
  for (uint64_t i = 0; i < n_pixels * 3; ++i) rgb8_out[i] = to_u8(rgb_sum[i], scale);
}

void rt_host_resolve_srgb8_f32(const float* rgb_sum, uint64_t n_pixels,
                               uint32_t samples_per_pixel, uint8_t* rgb8_out) {
  double scale = 1.0 / (double)samples_per_pixel;
  for (uint64_t i = 0; i < n_pixels * 3; ++i) rgb8_out[i] = to_u8((double)rgb_sum[i], scale);
}

int rt_host_decode_jpeg(const uint8_t* data, uint64_t size, uint32_t* width, uint32_t* height,
                        uint8_t* rgb8, uint64_t capacity) {
  if (!data || !width || !height) return set_error(RT_ERR_INVALID, "null argument");
  std::vector<uint8_t> rgb;
  std::string err;
  uint32_t w = 0, h = 0;
  if (!rt::decode_jpeg(data, (size_t)size, w, h, rgb, err)) return set_error(RT_ERR_UNSUPPORTED, err);
  *width = w;
  *height = h;
  if (rgb8) {
    if (capacity < rgb.size()) return set_error(RT_ERR_INVALID, "output buffer too small");
    std::memcpy(rgb8, rgb.data(), rgb.size());
  }
  return RT_OK;
}

int rt_host_write_png(const char* path, uint32_t width, uint32_t height, const uint8_t* rgb8) {
  if (!path || !rgb8 || width == 0 || height == 0) return set_error(RT_ERR_INVALID, "bad argument");
  // raw scanlines, filter type 0 each
  const size_t stride = (size_t)width * 3;
  std::vector<uint8_t> raw((stride + 1) * height);
  for (uint32_t y = 0; y < height; ++y) {
    raw[(stride + 1) * y] = 0;
    std::memcpy(&raw[(stride + 1) * y + 1], rgb8 + stride * y, stride);
  }
  uLongf zlen = compressBound((uLong)raw.size());
  std::vector<uint8_t> z(zlen);
  if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 6) != Z_OK)
    return set_error(RT_ERR_IO, "Adding frame to png: deflate failed");
  z.resize(zlen);

  std::vector<uint8_t> out;
  auto be32 = [&](uint32_t v) {
    out.push_back((uint8_t)(v >> 24));
    out.push_back((uint8_t)(v >> 16));
    out.push_back((uint8_t)(v >> 8));
    out.push_back((uint8_t)v);
  };
  auto chunk = [&](const char* type, const uint8_t* data, size_t len) {
    be32((uint32_t)len);
    size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    if (len) out.insert(out.end(), data, data + len);
    uint32_t crc = (uint32_t)crc32(0L, out.data() + start, (uInt)(out.size() - start));
    be32(crc);
  };
  const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  out.insert(out.end(), sig, sig + 8);
  uint8_t ihdr[13];
  ihdr[0] = (uint8_t)(width >> 24);
  ihdr[1] = (uint8_t)(width >> 16);
  ihdr[2] = (uint8_t)(width >> 8);
  ihdr[3] = (uint8_t)width;
  ihdr[4] = (uint8_t)(height >> 24);
  ihdr[5] = (uint8_t)(height >> 16);
  ihdr[6] = (uint8_t)(height >> 8);
  ihdr[7] = (uint8_t)height;
  ihdr[8] = 8;  // bit depth
  ihdr[9] = 2;  // colour type: RGB
  ihdr[10] = 0;
  ihdr[11] = 0;
  ihdr[12] = 0;
  chunk("IHDR", ihdr, 13);
  chunk("IDAT", z.data(), z.size());
  chunk("IEND", nullptr, 0);
  std::ofstream f(path, std::ios::binary);
  if (!f) return set_error(RT_ERR_IO, std::string("Saving image: cannot create ") + path);
  f.write((const char*)out.data(), (std::streamsize)out.size());
  if (!f) return set_error(RT_ERR_IO, std::string("Saving image: short write to ") + path);
  return RT_OK;
}

// human_duration (src/main.rs:146-165)
int rt_host_human_duration(uint64_t millis, char* buf, size_t buf_len) {
  const uint64_t SECOND = 1000, MINUTE = SECOND * 60, HOUR = MINUTE * 60;
  std::string parts;
  int n_parts = 0;
  if (millis > HOUR) {
    parts += std::to_string(millis / HOUR) + "h";
    millis %= HOUR;
    ++n_parts;
  }
  if (millis > MINUTE || n_parts > 0) {
    parts += std::to_string(millis / MINUTE) + "m";
    millis %= MINUTE;
    ++n_parts;
  }
  parts += std::to_string(millis / SECOND);
  millis %= SECOND;
  ++n_parts;
  if (n_parts == 1) {
    char frac[8];
    std::snprintf(frac, sizeof frac, ".%02u", (unsigned)(millis / 10));
    parts += frac;
  }
  parts += "s";
  if (buf && buf_len) {
    std::snprintf(buf, buf_len, "%s", parts.c_str());
  }
  return (int)parts.size();
}

}  // extern "C"
