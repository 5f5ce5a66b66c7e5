// Scene realisation: RON tree -> flat rt_scene.  Mirrors SceneLoader
// (src/scene/loader.rs) and the description types (src/scene/desc.rs), with two
// differences that the render path needs: every load-time random draw comes
// from a seeded generator, and both generations of scene syntax are accepted
// (lower/upper-case operator names, bare 3-tuples where a texture is expected).
#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstring>
#include <fstream>
#include <iterator>
#include <map>
#include <sstream>
#include <stdexcept>

#include "../csrc/philox.h"
#include "host_scene.h"
#include "jpeg.h"
#include "jpeg.h"
#include "ron.h"

void rt_host_scene::refresh_view() {
  view.n_objects = (uint32_t)objects.size();
  view.n_prims = (uint32_t)prims.size();
  view.n_ops = (uint32_t)ops.size();
  view.n_materials = (uint32_t)materials.size();
  view.n_textures = (uint32_t)textures.size();
  view.n_perlin = (uint32_t)perlin.size();
  view.n_images = (uint32_t)images.size();
  view.n_texel_bytes = texels.size();
  view.objects = objects.data();
  view.prims = prims.data();
  view.ops = ops.data();
  view.materials = materials.data();
  view.textures = textures.data();
  view.perlin = perlin.data();
  view.images = images.data();
  view.texels = texels.data();
}

namespace rthost {

// ---------------------------------------------------------------- load RNG

double LoadRng::next_f64() {
  rt::u32x4 c;
  c.x = (uint32_t)counter;
  c.y = (uint32_t)(counter >> 32);
  c.z = 0x5CE11E5Eu;  // domain tag: scene expansion
  c.w = 0;
  ++counter;
  rt::u32x4 r = rt::philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  return rt::u01_53(r.x, r.y);
}

double LoadRng::range(double lo, double hi) {
  // rand 0.8.5 panics on an empty range (src/scene/desc.rs:139).
  if (!(lo < hi)) throw std::runtime_error("cannot sample empty range");
  return lo + (hi - lo) * next_f64();
}

uint64_t LoadRng::below(uint64_t n) {
  if (n == 0) throw std::runtime_error("cannot sample empty range");
  uint64_t v = (uint64_t)(next_f64() * (double)n);
  return v >= n ? n - 1 : v;
}

void make_perlin(LoadRng& rng, rt_perlin& out) {
  // rand_vec: 256 x gen::<Vec3>() = 3 x gen_range(-1.0..1.0) (src/geom/vec3.rs:233-241)
  for (int i = 0; i < 256; ++i)
    for (int k = 0; k < 3; ++k) out.rand_vec[i][k] = rng.range(-1.0, 1.0);
  // perm_x/y/z: identity shuffled (src/texture/perlin.rs:32-36); Fisher-Yates from the top.
  uint32_t* perms[3] = {out.perm_x, out.perm_y, out.perm_z};
  for (uint32_t* p : perms) {
    for (uint32_t i = 0; i < 256; ++i) p[i] = i;
    for (uint32_t i = 255; i >= 1; --i) {
      uint32_t j = (uint32_t)rng.below((uint64_t)i + 1);
      std::swap(p[i], p[j]);
    }
  }
}

// ----------------------------------------------------------------- camera

namespace {
struct V3 {
  double x, y, z;
};
// Operator spellings follow src/geom/vec3.rs:141-225 (a - b is a + (-b); v / s is v * (1/s)).
inline V3 add(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 neg(V3 a) { return {-a.x, -a.y, -a.z}; }
inline V3 sub(V3 a, V3 b) { return add(a, neg(b)); }
inline V3 mul(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 divs(V3 a, double s) { return mul(a, 1.0 / s); }
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline double length(V3 a) { return std::sqrt(dot(a, a)); }
inline V3 unit(V3 a) { return divs(a, length(a)); }
inline V3 from(const double* p) { return {p[0], p[1], p[2]}; }
inline void store(double* p, V3 v) {
  p[0] = v.x;
  p[1] = v.y;
  p[2] = v.z;
}
}  // namespace

rt_camera build_camera(const rt_camera_desc& d, uint32_t width, uint32_t height) {
  const double kPi = 3.14159265358979323846264338327950288;
  V3 look_from = from(d.look_from);
  V3 look_at = d.has_look_at ? from(d.look_at) : V3{0.0, 0.0, 0.0};
  V3 look_vector = sub(look_from, look_at);
  const double lim = 0.001;  // Vec3::is_near_zero (src/geom/vec3.rs:118-121)
  if (std::fabs(look_vector.x) < lim && std::fabs(look_vector.y) < lim &&
      std::fabs(look_vector.z) < lim)
    throw std::runtime_error("look_at and look_from are too close together");
  double focus_dist = d.has_focus_distance ? d.focus_distance : length(sub(look_at, look_from));
  double aspect_ratio = (double)width / (double)height;
  double theta = d.vertical_fov * (kPi / 180.0);  // f64::to_radians
  double h = std::tan(theta / 2.0);
  double viewport_height = 2.0 * h;
  double viewport_width = aspect_ratio * viewport_height;
  V3 v_up = d.has_v_up ? from(d.v_up) : V3{0.0, 1.0, 0.0};
  V3 w = unit(look_vector);
  V3 u = unit(cross(v_up, w));
  V3 v = cross(w, u);
  V3 horizontal = mul(u, focus_dist * viewport_width);
  V3 vertical = mul(v, focus_dist * viewport_height);
  V3 llc = sub(sub(sub(look_from, divs(horizontal, 2.0)), divs(vertical, 2.0)), mul(w, focus_dist));
  rt_camera c{};
  store(c.origin, look_from);
  store(c.lower_left, llc);
  store(c.horizontal, horizontal);
  store(c.vertical, vertical);
  store(c.u, u);
  store(c.v, v);
  c.lens_radius = d.aperture / 2.0;
  c.time0 = d.has_shutter ? d.shutter0 : 0.0;
  c.time1 = d.has_shutter ? d.shutter1 : 0.0;
  return c;
}

// ------------------------------------------------------------------ assets

std::vector<std::string>& AssetDirs::list() {
  static std::vector<std::string> dirs;
  return dirs;
}

bool read_ppm(const std::string& path, uint32_t& w, uint32_t& h, std::vector<uint8_t>& rgb,
              std::string& err) {
  std::ifstream f(path, std::ios::binary);
  if (!f) {
    err = "cannot open " + path;
    return false;
  }
  auto token = [&](std::string& out) -> bool {
    out.clear();
    int c;
    for (;;) {
      c = f.get();
      if (c == EOF) return false;
      if (c == '#') {
        while (c != '\n' && c != EOF) c = f.get();
        continue;
      }
      if (!std::isspace(c)) break;
    }
    while (c != EOF && !std::isspace(c)) {
      out.push_back((char)c);
      c = f.get();
    }
    return true;
  };
  std::string magic, sw, sh, smax;
  if (!token(magic) || magic != "P6" || !token(sw) || !token(sh) || !token(smax)) {
    err = path + ": not a binary PPM (P6)";
    return false;
  }
  w = (uint32_t)std::strtoul(sw.c_str(), nullptr, 10);
  h = (uint32_t)std::strtoul(sh.c_str(), nullptr, 10);
  if (w == 0 || h == 0 || std::strtoul(smax.c_str(), nullptr, 10) != 255) {
    err = path + ": unsupported PPM header";
    return false;
  }
  rgb.resize((size_t)w * h * 3);
  f.read((char*)rgb.data(), (std::streamsize)rgb.size());
  if ((size_t)f.gcount() != rgb.size()) {
    err = path + ": truncated PPM";
    return false;
  }
  return true;
}

// ------------------------------------------------------------------ loader

namespace {

std::string lower(std::string s) {
  for (auto& c : s) c = (char)std::tolower((unsigned char)c);
  return s;
}

struct ObjBuild {
  std::vector<rt_prim> prims;
  std::vector<rt_op> ops;        // outermost first; outside the medium when is_medium
  std::vector<rt_op> inner_ops;  // between medium and base
  bool is_medium = false;
  uint32_t medium_material = 0;
  double neg_inv_density = 0.0;
  bool is_cuboid = false;
  double cuboid_center[3] = {0, 0, 0};
  double cuboid_half[3] = {0, 0, 0};
};

class Loader {
 public:
  Loader(rt_host_scene& s, const std::string& base_dir, uint64_t seed)
      : scene_(s), base_dir_(base_dir), rng_(seed) {}

  void load(const RonNode& root, const rt_host_options& opt) {
    if (root.kind != RonNode::STRUCT) fail(root, "expected Scene(...)");
    if (!root.text.empty() && root.text != "Scene") fail(root, "expected struct `Scene`");

    // materials first (src/scene/loader.rs:54-57); the reference iterates a HashMap,
    // so its order is unspecified: file order here.
    const RonNode* mats = need(root, "materials");
    if (mats->kind != RonNode::MAP) fail(*mats, "materials must be a map");
    for (const auto& kv : mats->fields) shared_[kv.first] = realize_material(kv.second);

    const RonNode* objs = need(root, "objects");
    if (objs->kind != RonNode::LIST) fail(*objs, "objects must be a list");
    std::vector<ObjBuild> accum;
    for (const RonNode& o : objs->items) realize_hittable(o, accum);

    const RonNode* img = need(root, "image");
    scene_.view.width = (uint32_t)integer(*need(*img, "width"));
    scene_.view.height = (uint32_t)integer(*need(*img, "height"));
    scene_.view.samples_per_pixel = (uint32_t)integer(*need(*img, "samples_per_pixel"));
    scene_.view.max_depth = (uint32_t)integer(*need(*img, "max_depth"));
    if (opt.width) scene_.view.width = opt.width;
    if (opt.height) scene_.view.height = opt.height;
    if (opt.samples_per_pixel) scene_.view.samples_per_pixel = opt.samples_per_pixel;
    if (opt.max_depth) scene_.view.max_depth = opt.max_depth;
    if (scene_.view.width < 2 || scene_.view.height < 2)
      fail(*img, "image must be at least 2x2");  // u,v divide by (w-1),(h-1): src/main.rs:91-93

    const RonNode* cam = need(root, "camera");
    rt_camera_desc& cd = scene_.camera_desc;
    plain_vec3(*need(*cam, "look_from"), cd.look_from);
    if (const RonNode* n = cam->field("look_at")) {
      plain_vec3(*n, cd.look_at);
      cd.has_look_at = 1;
    }
    if (const RonNode* n = cam->field("v_up")) {
      plain_vec3(*n, cd.v_up);
      cd.has_v_up = 1;
    }
    cd.vertical_fov = number(*need(*cam, "vertical_fov"));
    cd.aperture = number(*need(*cam, "aperture"));
    if (const RonNode* n = cam->field("focus_distance")) {
      cd.focus_distance = number(*n);
      cd.has_focus_distance = 1;
    }
    if (const RonNode* n = cam->field("shutter_time")) {
      if (n->kind != RonNode::TUPLE || n->items.size() != 2) fail(*n, "shutter_time: (start, end)");
      cd.shutter0 = number(n->items[0]);
      cd.shutter1 = number(n->items[1]);
      cd.has_shutter = 1;
    }
    scene_.view.camera = build_camera(cd, scene_.view.width, scene_.view.height);

    // background: default black (src/scene/loader.rs:89-91)
    scene_.view.background[0] = scene_.view.background[1] = scene_.view.background[2] = 0.0;
    if (const RonNode* bg = root.field("background")) eval_vec3(*bg, scene_.view.background);

    flatten(accum);
    scene_.refresh_view();
  }

 private:
  rt_host_scene& scene_;
  std::string base_dir_;
  LoadRng rng_;
  std::map<std::string, int32_t> vars_;
  std::map<std::string, uint32_t> shared_;
  std::map<std::string, uint32_t> image_cache_;

  [[noreturn]] static void fail(const RonNode& n, const std::string& what) {
    throw std::runtime_error("loading scene file: line " + std::to_string(n.line) + ": " + what);
  }
  static const RonNode* need(const RonNode& n, const char* name) {
    const RonNode* f = n.field(name);
    if (!f) fail(n, std::string("missing field `") + name + "`");
    return f;
  }
  static double number(const RonNode& n) {
    if (n.kind != RonNode::NUMBER) fail(n, "expected a number");
    return n.number;
  }
  static int64_t integer(const RonNode& n) {
    if (n.kind != RonNode::NUMBER || !n.is_integer) fail(n, "expected an integer");
    return (int64_t)n.number;
  }
  static void plain_vec3(const RonNode& n, double* out) {
    if (n.kind != RonNode::TUPLE || n.items.size() != 3) fail(n, "expected (x, y, z)");
    for (int i = 0; i < 3; ++i) out[i] = number(n.items[i]);
  }

  // Value::eval (src/scene/desc.rs:124-147)
  double eval(const RonNode& n) {
    if (n.kind == RonNode::NUMBER) return n.number;
    if (n.kind == RonNode::STRING) {
      auto it = vars_.find(n.text);
      if (it == vars_.end()) throw std::runtime_error("Variable " + n.text + " not found");
      return (double)it->second;
    }
    if (n.kind == RonNode::TUPLE && n.items.size() == 3 && n.text.empty()) {
      const RonNode& opn = n.items[0];
      std::string op;
      if (opn.kind == RonNode::STRING)
        op = lower(opn.text);
      else if (opn.is_unit())
        op = lower(opn.text);
      else
        fail(n, "expected (op, a, b)");
      double a = eval(n.items[1]);
      double b = eval(n.items[2]);
      if (op == "rand") return rng_.range(a, b);
      if (op == "add") return a + b;
      if (op == "mult") return a * b;
      fail(opn, "unknown operator `" + opn.text + "` (Add, Mult, Rand)");
    }
    fail(n, "expected a value: number, \"variable\" or (op, a, b)");
  }

  void eval_vec3(const RonNode& n, double* out) {
    if (n.kind != RonNode::TUPLE || n.items.size() != 3 || !n.text.empty())
      fail(n, "expected (x, y, z)");
    for (int i = 0; i < 3; ++i) out[i] = eval(n.items[i]);
  }

  uint32_t add_texture(const rt_texture& t) {
    scene_.textures.push_back(t);
    return (uint32_t)scene_.textures.size() - 1;
  }
  uint32_t add_material(const rt_material& m) {
    scene_.materials.push_back(m);
    return (uint32_t)scene_.materials.size() - 1;
  }

  // realize_texture (src/scene/loader.rs:356-377)
  uint32_t realize_texture(const RonNode& n) {
    rt_texture t{};
    if (n.kind == RonNode::TUPLE && n.text.empty() && n.items.size() == 3) {
      // older scene syntax: a bare (r, g, b) of values where a texture is expected
      t.kind = RT_TEX_SOLID;
      eval_vec3(n, t.color);
      return add_texture(t);
    }
    if (n.kind != RonNode::TUPLE) fail(n, "expected a texture");
    if (n.text == "Solid") {
      if (n.items.size() != 3) fail(n, "Solid(r, g, b)");
      t.kind = RT_TEX_SOLID;
      for (int i = 0; i < 3; ++i) t.color[i] = eval(n.items[i]);
      return add_texture(t);
    }
    if (n.text == "Checkerboard") {
      if (n.items.size() != 2) fail(n, "Checkerboard(even, odd)");
      t.kind = RT_TEX_CHECKER;
      t.a = realize_texture(n.items[0]);
      t.b = realize_texture(n.items[1]);
      return add_texture(t);
    }
    if (n.text == "Perlin") {
      if (!n.items.empty()) fail(n, "Perlin takes no arguments");
      t.kind = RT_TEX_PERLIN;
      t.scale = 4.0;  // Perlin::default (src/texture/perlin.rs:103-107)
      scene_.perlin.emplace_back();
      make_perlin(rng_, scene_.perlin.back());
      t.a = (uint32_t)scene_.perlin.size() - 1;
      return add_texture(t);
    }
    if (n.text == "Image") {
      if (n.items.size() != 1 || n.items[0].kind != RonNode::STRING) fail(n, "Image(\"path\")");
      t.kind = RT_TEX_IMAGE;
      t.a = load_image(n.items[0].text, n);
      return add_texture(t);
    }
    fail(n, "unknown texture `" + n.text + "`");
  }

  uint32_t load_image(const std::string& rel, const RonNode& at) {
    auto it = image_cache_.find(rel);
    if (it != image_cache_.end()) return it->second;
    std::string adjusted = base_dir_.empty() ? rel : base_dir_ + "/" + rel;
    std::string file = rel;
    size_t slash = file.find_last_of('/');
    if (slash != std::string::npos) file = file.substr(slash + 1);
    std::string stem = file;
    size_t dotp = stem.find_last_of('.');
    if (dotp != std::string::npos) stem = stem.substr(0, dotp);

    std::vector<std::string> candidates;
    if (file.size() > 4 && lower(file.substr(file.size() - 4)) == ".ppm") candidates.push_back(adjusted);
    {
      std::string dir = adjusted;
      size_t s2 = dir.find_last_of('/');
      dir = s2 == std::string::npos ? std::string(".") : dir.substr(0, s2);
      candidates.push_back(dir + "/" + stem + ".ppm");
    }
    for (const auto& d : AssetDirs::list()) candidates.push_back(d + "/" + stem + ".ppm");

    // the file itself when it is a baseline JPEG, else an already decoded PPM next to it
    // or in an asset directory
    std::vector<std::string> originals{adjusted};
    for (const auto& d : AssetDirs::list()) originals.push_back(d + "/" + file);

    uint32_t w = 0, h = 0;
    std::vector<uint8_t> rgb;
    std::string err, tried;
    auto keep = [&]() {
      rt_image im{};
      im.width = w;
      im.height = h;
      im.offset = scene_.texels.size();
      scene_.texels.insert(scene_.texels.end(), rgb.begin(), rgb.end());
      scene_.images.push_back(im);
      uint32_t idx = (uint32_t)scene_.images.size() - 1;
      image_cache_[rel] = idx;
      return idx;
    };
    for (const auto& c : originals) {
      std::ifstream f(c, std::ios::binary);
      if (!f) {
        tried += (tried.empty() ? "" : ", ") + c;
        continue;
      }
      std::vector<uint8_t> bytes((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
      if (bytes.size() >= 2 && bytes[0] == 0xFF && bytes[1] == 0xD8) {
        if (rt::decode_jpeg(bytes.data(), bytes.size(), w, h, rgb, err)) return keep();
        tried += (tried.empty() ? "" : ", ") + c + " (" + err + ")";
      } else {
        tried += (tried.empty() ? "" : ", ") + c + " (not a JPEG)";
      }
    }
    for (const auto& c : candidates) {
      if (read_ppm(c, w, h, rgb, err)) return keep();
      tried += (tried.empty() ? "" : ", ") + c;
    }
    fail(at, "Adjusted original path " + rel + " to " + adjusted +
                 ": no baseline JPEG or decoded PPM found (tried " + tried + ")");
  }

  // realize_material (src/scene/loader.rs:313-354)
  uint32_t realize_material(const RonNode& n) {
    if (n.kind != RonNode::TUPLE && n.kind != RonNode::STRUCT) fail(n, "expected a material");
    rt_material m{};
    if (n.text == "Shared") {
      if (n.items.size() != 1 || n.items[0].kind != RonNode::STRING) fail(n, "Shared(\"name\")");
      auto it = shared_.find(n.items[0].text);
      if (it == shared_.end())
        throw std::runtime_error("Material " + n.items[0].text + " not defined");
      return it->second;
    }
    if (n.text == "Lambertian") {
      m.kind = RT_MAT_LAMBERTIAN;
      m.texture = realize_texture(*need(n, "albedo"));
      return add_material(m);
    }
    if (n.text == "Metal") {
      m.kind = RT_MAT_METAL;
      eval_vec3(*need(n, "albedo"), m.albedo);
      m.param = eval(*need(n, "fuzziness"));
      return add_material(m);
    }
    if (n.text == "Dielectric") {
      m.kind = RT_MAT_DIELECTRIC;
      m.param = eval(*need(n, "index_of_refraction"));
      return add_material(m);
    }
    if (n.text == "DiffuseLight") {
      m.kind = RT_MAT_DIFFUSE_LIGHT;
      m.texture = realize_texture(*need(n, "color"));
      return add_material(m);
    }
    if (n.text == "RandomChoice") {
      if (n.items.size() != 1 || n.items[0].kind != RonNode::LIST) fail(n, "RandomChoice([..])");
      const auto& options = n.items[0].items;
      uint64_t idx = rng_.below(options.size());
      return realize_material(options[idx]);
    }
    if (n.text == "RandomChoiceWeighted") {
      if (n.items.size() != 1 || n.items[0].kind != RonNode::LIST)
        fail(n, "RandomChoiceWeighted([(w, m), ..])");
      const auto& options = n.items[0].items;
      std::vector<double> cumulative;
      double total = 0.0;
      for (const RonNode& o : options) {
        if (o.kind != RonNode::TUPLE || o.items.size() != 2) fail(o, "expected (weight, material)");
        double w = number(o.items[0]);
        if (!(w >= 0.0)) throw std::runtime_error("generating weighted distribution: invalid weight");
        total += w;
        cumulative.push_back(total);
      }
      if (options.empty() || !(total > 0.0))
        throw std::runtime_error("generating weighted distribution: no usable weights");
      double chosen = rng_.range(0.0, total);
      size_t idx = 0;
      while (idx + 1 < cumulative.size() && cumulative[idx] <= chosen) ++idx;
      return realize_material(options[idx].items[1]);
    }
    fail(n, "unknown material `" + n.text + "`");
  }

  static uint32_t axis_of(const RonNode& n) {
    if (n.is_unit()) {
      if (n.text == "X") return 0;
      if (n.text == "Y") return 1;
      if (n.text == "Z") return 2;
    }
    fail(n, "expected axis X, Y or Z");
  }

  // realize_hittable / realize_pattern (src/scene/loader.rs:95-311)
  void realize_hittable(const RonNode& n, std::vector<ObjBuild>& out) {
    if (n.kind != RonNode::STRUCT) fail(n, "expected an object");
    const std::string& k = n.text;
    if (k == "Sphere") {
      ObjBuild o;
      rt_prim p{};
      p.kind = RT_PRIM_SPHERE;
      eval_vec3(*need(n, "center"), p.center);
      p.radius = eval(*need(n, "radius"));
      p.material = realize_material(*need(n, "material"));
      o.prims.push_back(p);
      out.push_back(std::move(o));
    } else if (k == "MovingSphere") {
      ObjBuild o;
      rt_prim p{};
      p.kind = RT_PRIM_MOVING_SPHERE;
      const RonNode* c = need(n, "center");
      if (c->kind != RonNode::TUPLE || c->items.size() != 2) fail(*c, "center: (start, end)");
      eval_vec3(c->items[0], p.center);
      eval_vec3(c->items[1], p.center1);
      const RonNode* t = need(n, "time");
      if (t->kind != RonNode::TUPLE || t->items.size() != 2) fail(*t, "time: (start, end)");
      p.time0 = eval(t->items[0]);
      p.time1 = eval(t->items[1]);
      p.radius = eval(*need(n, "radius"));
      p.material = realize_material(*need(n, "material"));
      o.prims.push_back(p);
      out.push_back(std::move(o));
    } else if (k == "AARect") {
      ObjBuild o;
      rt_prim p{};
      p.kind = RT_PRIM_RECT;
      eval_vec3(*need(n, "center"), p.center);
      p.width = eval(*need(n, "width"));
      p.height = eval(*need(n, "height"));
      p.axis = axis_of(*need(n, "axis"));
      p.material = realize_material(*need(n, "material"));
      o.prims.push_back(p);
      out.push_back(std::move(o));
    } else if (k == "Cuboid") {
      ObjBuild o;
      double center[3] = {0.0, 0.0, 0.0}, size[3];
      if (const RonNode* c = n.field("center")) eval_vec3(*c, center);
      eval_vec3(*need(n, "size"), size);
      uint32_t mat = realize_material(*need(n, "material"));
      // Cuboid::new (src/hittable/cuboid.rs:18-37)
      double half[3];
      for (int i = 0; i < 3; ++i) half[i] = size[i] * (1.0 / 2.0);
      for (uint32_t axis = 0; axis < 3; ++axis) {
        for (double mult : {-1.0, 1.0}) {
          rt_prim p{};
          p.kind = RT_PRIM_RECT;
          p.axis = axis;
          p.material = mat;
          for (int i = 0; i < 3; ++i) p.center[i] = center[i];
          // center + half.project(axis) * mult: only the axis component moves
          p.center[axis] = center[axis] + (half[axis] * mult);
          p.width = size[(axis + 1) % 3];   // size[axis.next()]
          p.height = size[(axis + 2) % 3];  // size[axis.prev()]
          o.prims.push_back(p);
        }
      }
      o.is_cuboid = true;
      for (int i = 0; i < 3; ++i) {
        o.cuboid_center[i] = center[i];
        o.cuboid_half[i] = half[i];
      }
      out.push_back(std::move(o));
    } else if (k == "Pattern") {
      const RonNode* var = need(n, "var");
      if (var->kind != RonNode::STRING) fail(*var, "var must be a string");
      const RonNode* range = need(n, "range");
      if (range->kind != RonNode::LIST) fail(*range, "range must be a list");
      std::vector<int64_t> r;
      for (const RonNode& x : range->items) r.push_back(integer(x));
      int64_t start = 0, end = 0, step = 1;
      if (r.size() == 1) {
        end = r[0];
      } else if (r.size() == 2) {
        start = r[0];
        end = r[1];
      } else if (r.size() == 3) {
        start = r[0];
        end = r[1];
        step = r[2] < 0 ? -r[2] : r[2];
        if (step == 0) throw std::runtime_error("Pattern step must not be zero");
      } else {
        throw std::runtime_error("Unexpected format for range");
      }
      const RonNode* object = need(n, "object");
      for (int64_t v = start; v < end; v += step) {
        vars_[var->text] = (int32_t)v;
        realize_hittable(*object, out);
      }
    } else if (k == "Translate") {
      rt_op op{};
      op.kind = RT_OP_TRANSLATE;
      eval_vec3(*need(n, "offset"), op.a);
      std::vector<ObjBuild> inner;
      realize_hittable(*need(n, "hittable"), inner);
      for (auto& o : inner) {
        o.ops.insert(o.ops.begin(), op);
        out.push_back(std::move(o));
      }
    } else if (k == "RotateY") {
      const double kPi = 3.14159265358979323846264338327950288;
      double theta = eval(*need(n, "angle")) * (kPi / 180.0);  // to_radians (loader.rs:164)
      rt_op op{};
      op.kind = RT_OP_ROTATE_Y;
      op.a[0] = std::sin(theta);  // sin_cos (src/hittable/instance.rs:38)
      op.a[1] = std::cos(theta);
      std::vector<ObjBuild> inner;
      realize_hittable(*need(n, "hittable"), inner);
      for (auto& o : inner) {
        o.ops.insert(o.ops.begin(), op);
        out.push_back(std::move(o));
      }
    } else if (k == "ConstantMedium") {
      // order of evaluation: texture, density, boundary (src/scene/loader.rs:175-186)
      uint32_t tex = realize_texture(*need(n, "texture"));
      double density = eval(*need(n, "density"));
      std::vector<ObjBuild> inner;
      realize_hittable(*need(n, "boundary"), inner);
      rt_material iso{};
      iso.kind = RT_MAT_ISOTROPIC;
      iso.texture = tex;
      uint32_t mat = add_material(iso);
      for (auto& o : inner) {
        if (o.is_medium)
          throw std::runtime_error(
              "unsupported: a ConstantMedium whose boundary is itself a ConstantMedium");
        o.is_medium = true;
        o.medium_material = mat;
        o.neg_inv_density = -(1.0 / density);  // -density.recip() (constant_medium.rs:20)
        o.inner_ops = std::move(o.ops);
        o.ops.clear();
        out.push_back(std::move(o));
      }
    } else {
      fail(n, "unknown object `" + k + "`");
    }
  }

  void flatten(const std::vector<ObjBuild>& accum) {
    for (const ObjBuild& b : accum) {
      rt_object o{};
      o.first_prim = (uint32_t)scene_.prims.size();
      o.prim_count = (uint32_t)b.prims.size();
      scene_.prims.insert(scene_.prims.end(), b.prims.begin(), b.prims.end());
      o.first_op = (uint32_t)scene_.ops.size();
      o.op_count = (uint32_t)b.ops.size();
      scene_.ops.insert(scene_.ops.end(), b.ops.begin(), b.ops.end());
      o.inner_first_op = (uint32_t)scene_.ops.size();
      o.inner_op_count = (uint32_t)b.inner_ops.size();
      scene_.ops.insert(scene_.ops.end(), b.inner_ops.begin(), b.inner_ops.end());
      o.is_medium = b.is_medium ? 1u : 0u;
      o.medium_material = b.medium_material;
      o.neg_inv_density = b.neg_inv_density;
      o.is_cuboid = b.is_cuboid ? 1u : 0u;
      for (int i = 0; i < 3; ++i) {
        o.cuboid_center[i] = b.cuboid_center[i];
        o.cuboid_half[i] = b.cuboid_half[i];
      }
      scene_.objects.push_back(o);
    }
  }
};

}  // namespace

rt_host_scene* load_ron_text(const std::string& text, const std::string& base_dir,
                             const rt_host_options& opt) {
  RonNode root;
  try {
    root = ron_parse(text);
  } catch (const std::runtime_error& e) {
    throw std::runtime_error(std::string("loading scene file: ") + e.what());
  }
  auto* s = new rt_host_scene();
  try {
    Loader l(*s, base_dir, opt.scene_seed);
    l.load(root, opt);
  } catch (...) {
    delete s;
    throw;
  }
  return s;
}

}  // namespace rthost
