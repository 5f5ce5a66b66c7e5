// oracle.cpp — CPU restatement of mythmon/raytracer's render path.
//
// TEST INFRASTRUCTURE ONLY.  Nothing in the product (raytracer_amd/, include/)
// may include, link or call this file; only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg do, as the checker.
//
// PARITY UNPINNED: the reference is Rust and cannot be built here, and its own
// tests hold no vector for the render path.  The only reference tests are
// Vec3::project (src/geom/vec3.rs:299-309) and Aabb::surrounding
// (src/geom/aabb.rs:87-104); both are restated and exported below.  Everything
// else is pinned by hand-derived known answers in tests/.
//
// The code keeps the reference's structure (one class per trait impl, virtual
// dispatch, Option-like returns) and its exact operation order in f64
// (a - b is a + (-b), v / s is v * (1/s), powi by repeated multiplication);
// compile with -O2 -ffp-contract=off -fno-fast-math.  Randomness: the reference
// draws from an unseeded thread_rng(); only distributions are reproducible, so
// draws come from Philox4x32-10 addressed by (pixel, sample, bounce, purpose),
// the same addressing the device uses.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <functional>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "../include/mi355rt.h"

namespace {

const double INF = std::numeric_limits<double>::infinity();
const double PI = 3.14159265358979323846264338327950288;

// ------------------------------------------------------------------ Philox
struct U4 {
  uint32_t x, y, z, w;
};

U4 philox4x32_10(U4 c, uint32_t k0, uint32_t k1) {
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c.x;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c.z;
    U4 n;
    n.x = (uint32_t)(p1 >> 32) ^ c.y ^ k0;
    n.y = (uint32_t)p1;
    n.z = (uint32_t)(p0 >> 32) ^ c.w ^ k1;
    n.w = (uint32_t)p0;
    c = n;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return c;
}

enum { DRAW_JITTER = 0, DRAW_LENS = 1, DRAW_TIME = 2, DRAW_SCATTER = 3, DRAW_FRESNEL = 4, DRAW_MEDIUM = 5 };

struct DrawCtx {
  uint64_t seed;
  uint32_t pixel, sample, bounce;
  U4 block(uint32_t purpose, uint32_t seq) const {
    U4 c{pixel, sample, bounce, (purpose << 24) | (seq & 0x00FFFFFFu)};
    return philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  }
};

double u01(uint32_t lo, uint32_t hi) {  // rand `Standard` f64: 53 bits in [0,1)
  uint64_t v = ((uint64_t)hi << 32) | lo;
  return (double)(v >> 11) * (1.0 / 9007199254740992.0);
}

void sym3(U4 b, double& x, double& y, double& z) {  // three 42-bit uniforms in [-1,1)
  uint64_t a = ((uint64_t)b.y << 32) | b.x;
  uint64_t c = ((uint64_t)b.w << 32) | b.z;
  uint64_t ix = a >> 22, iy = c >> 22;
  uint64_t iz = ((a & 0x1FFFFFull) << 21) | (c & 0x1FFFFFull);
  const double s = 2.0 / 4398046511104.0;
  x = (double)ix * s - 1.0;
  y = (double)iy * s - 1.0;
  z = (double)iz * s - 1.0;
}

// -------------------------------------------------------------------- Vec3
// src/geom/vec3.rs
struct Vec3 {
  double e[3];
  Vec3() : e{0, 0, 0} {}
  Vec3(double a, double b, double c) : e{a, b, c} {}
  explicit Vec3(const double* p) : e{p[0], p[1], p[2]} {}
  double x() const { return e[0]; }
  double y() const { return e[1]; }
  double z() const { return e[2]; }
  double operator[](int i) const { return e[i]; }
  double& operator[](int i) { return e[i]; }
};
typedef Vec3 Color;
typedef Vec3 Point3;

Vec3 operator-(Vec3 a) { return Vec3(-a.e[0], -a.e[1], -a.e[2]); }                            // :141-146
Vec3 operator+(Vec3 a, Vec3 b) { return Vec3(a.e[0] + b.e[0], a.e[1] + b.e[1], a.e[2] + b.e[2]); }  // :148-153
Vec3 operator-(Vec3 a, Vec3 b) { return a + -b; }                                             // :163-168
Vec3 operator*(Vec3 a, double s) { return Vec3(a.e[0] * s, a.e[1] * s, a.e[2] * s); }         // :176-181
Vec3 operator*(double s, Vec3 a) { return a * s; }                                            // :183-188
Vec3 operator*(Vec3 a, Vec3 b) { return Vec3(a.e[0] * b.e[0], a.e[1] * b.e[1], a.e[2] * b.e[2]); }  // :198-203
Vec3 operator/(Vec3 a, double s) { return a * (1.0 / s); }                                    // :214-219
double dot(Vec3 a, Vec3 b) { return a.e[0] * b.e[0] + a.e[1] * b.e[1] + a.e[2] * b.e[2]; }   // :70-72
double length_squared(Vec3 a) { return dot(a, a); }
double length(Vec3 a) { return std::sqrt(length_squared(a)); }
Vec3 unit_vector(Vec3 a) { return a / length(a); }  // :94-96
Vec3 cross(Vec3 a, Vec3 b) {                         // :74-80
  return Vec3(a.e[1] * b.e[2] - a.e[2] * b.e[1], a.e[2] * b.e[0] - a.e[0] * b.e[2],
              a.e[0] * b.e[1] - a.e[1] * b.e[0]);
}
Vec3 reflect(Vec3 v, Vec3 n) { return v - 2.0 * dot(v, n) * n; }  // :83-85
double fmin_rust(double a, double b) { return std::fmin(a, b); }  // f64::min ignores a NaN
double fmax_rust(double a, double b) { return std::fmax(a, b); }
Vec3 refract(Vec3 v, Vec3 n, double eta) {  // :87-92
  double cos_theta = fmin_rust(dot(-v, n), 1.0);
  Vec3 r_out_perp = eta * (v + cos_theta * n);
  Vec3 r_out_parallel = std::sqrt(std::fabs(1.0 - length_squared(r_out_perp))) * -n;
  return r_out_perp + r_out_parallel;
}
bool is_near_zero(Vec3 v) {  // :118-121
  const double LIMIT = 0.001;
  return std::fabs(v.e[0]) < LIMIT && std::fabs(v.e[1]) < LIMIT && std::fabs(v.e[2]) < LIMIT;
}
Vec3 project(Vec3 a, Vec3 rhs) {  // :135-138
  Vec3 rhs_unit = unit_vector(rhs);
  return rhs_unit * dot(a, rhs_unit);
}
double powi2(double x) { return x * x; }
double powi5(double x) {  // compiler-rt __powidf2 / LLVM powi expansion: x * (x^2)^2
  double x2 = x * x;
  double x4 = x2 * x2;
  return x * x4;
}

Vec3 rand_unit_vector(const DrawCtx& d) {  // :35-47
  // the loop is capped like the device's (probability of reaching the cap: 2^-4000)
  Vec3 v;
  for (uint32_t it = 0; it < 4096; ++it) {
    double x, y, z;
    sym3(d.block(DRAW_SCATTER, it), x, y, z);
    v = Vec3(x, y, z);
    if (length_squared(v) < 1.0) break;
  }
  return unit_vector(v);
}
Vec3 rand_in_unit_disk(const DrawCtx& d) {  // :49-60
  Vec3 v;
  for (uint32_t it = 0; it < 4096; ++it) {
    double x, y, z;
    sym3(d.block(DRAW_LENS, it), x, y, z);
    v = Vec3(x, y, 0.0);
    if (length_squared(v) < 1.0) break;
  }
  return v;
}

// --------------------------------------------------------------------- Ray
struct Ray {  // src/geom/ray.rs
  Point3 origin;
  Vec3 direction;
  double time;
  Point3 along(double portion) const { return origin + portion * direction; }
};

struct Range {
  double start, end;
  bool contains(double x) const { return start <= x && x < end; }  // Range<f64>::contains
};

// -------------------------------------------------------------------- Aabb
struct Aabb {  // src/geom/aabb.rs
  Point3 min, max;
  static Aabb make(Point3 a, Point3 b) {  // :16-21
    Aabb r;
    for (int i = 0; i < 3; ++i) {
      r.min[i] = fmin_rust(a[i], b[i]);
      r.max[i] = fmax_rust(a[i], b[i]);
    }
    return r;
  }
  bool intersect(const Ray& ray, Range t_range) const {  // :35-50
    for (int a = 0; a < 3; ++a) {
      double inv_dir = 1.0 / ray.direction[a];
      double t0 = (min[a] - ray.origin[a]) * inv_dir;
      double t1 = (max[a] - ray.origin[a]) * inv_dir;
      if (inv_dir < 0.0) std::swap(t0, t1);
      double t_min = fmax_rust(t0, t_range.start);
      double t_max = fmin_rust(t1, t_range.end);
      if (t_max <= t_min) return false;
    }
    return true;
  }
  Vec3 span() const { return max - min; }  // :52-54
};

// OrderedFloat total order: NaN is greatest and equal to itself.
bool of_less(double a, double b) {
  if (std::isnan(a)) return false;
  if (std::isnan(b)) return true;
  return a < b;
}

struct OptAabb {
  bool some = false;
  Aabb box;
};

OptAabb surrounding(const std::vector<Aabb>& parts) {  // :23-33
  OptAabb r;
  if (parts.empty()) return r;  // Iterator::min() of nothing is None
  for (int axis = 0; axis < 3; ++axis) {
    double mn = parts[0].min[axis], mx = parts[0].max[axis];
    for (size_t i = 1; i < parts.size(); ++i) {
      if (of_less(parts[i].min[axis], mn)) mn = parts[i].min[axis];
      if (!of_less(parts[i].max[axis], mx)) mx = parts[i].max[axis];  // max() keeps the last of equals
    }
    r.box.min[axis] = mn;
    r.box.max[axis] = mx;
  }
  r.some = true;
  return r;
}

// --------------------------------------------------------------- Textures
struct SceneData;

struct Texture {  // src/texture/mod.rs:16-19
  virtual ~Texture() {}
  virtual Color value(double u, double v, Point3 p) const = 0;
};

struct SolidColor : Texture {  // src/texture/solid_color.rs
  Color c;
  Color value(double, double, Point3) const override { return c; }
};

struct Checkerboard : Texture {  // src/texture/checkerboard.rs:17-28
  const Texture* even;
  const Texture* odd;
  Color value(double u, double v, Point3 p) const override {
    double sines = 1.0;  // Iterator::product starts from 1.0
    for (int a = 0; a < 3; ++a) sines = sines * std::sin(p[a] * 10.0);
    return sines < 0.0 ? odd->value(u, v, p) : even->value(u, v, p);
  }
};

struct Perlin : Texture {  // src/texture/perlin.rs
  const rt_perlin* t;
  double scale;
  static double interp(const Vec3 c[2][2][2], double u, double v, double w) {  // :79-100
    double uu = u * u * (3.0 - 2.0 * u);
    double vv = v * v * (3.0 - 2.0 * v);
    double ww = w * w * (3.0 - 2.0 * w);
    double accum = 0.0;
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 2; ++j)
        for (int k = 0; k < 2; ++k) {
          double fi = i, fj = j, fk = k;
          Vec3 weight_v(u - fi, v - fj, w - fk);
          accum += dot(c[i][j][k], weight_v) * (fi * uu + (1.0 - fi) * (1.0 - uu)) *
                   (fj * vv + (1.0 - fj) * (1.0 - vv)) * (fk * ww + (1.0 - fk) * (1.0 - ww));
        }
    return accum;
  }
  double noise(Point3 p) const {  // :38-63
    double u = p.x() - std::floor(p.x());
    double v = p.y() - std::floor(p.y());
    double w = p.z() - std::floor(p.z());
    u = u * u * (3.0 - 2.0 * u);
    v = v * v * (3.0 - 2.0 * v);
    w = w * w * (3.0 - 2.0 * w);
    int64_t i = (int64_t)std::floor(p.x());
    int64_t j = (int64_t)std::floor(p.y());
    int64_t k = (int64_t)std::floor(p.z());
    Vec3 corners[2][2][2];
    for (int di = 0; di < 2; ++di)
      for (int dj = 0; dj < 2; ++dj)
        for (int dk = 0; dk < 2; ++dk) {
          uint32_t idx = t->perm_x[(i + di) & 0xFF] ^ t->perm_y[(j + dj) & 0xFF] ^
                         t->perm_z[(k + dk) & 0xFF];
          corners[di][dj][dk] = Vec3(t->rand_vec[idx]);
        }
    return interp(corners, u, v, w);
  }
  double turb(Point3 p, int depth) const {  // :65-76
    double accum = 0.0, weight = 1.0;
    for (int i = 0; i < depth; ++i) {
      accum += weight * noise(p);
      weight *= 0.5;
      p = p * 2.0;
    }
    return std::fabs(accum);
  }
  Color value(double, double, Point3 p) const override {  // :109-116
    return Color(1.0, 1.0, 1.0) * 0.5 * (1.0 + std::sin(scale * p.z() + 10.0 * turb(p, 7)));
  }
};

// Rust `as u32` from f64: saturating, NaN -> 0.
uint32_t as_u32(double x) {
  if (!(x == x)) return 0;
  if (x <= 0.0) return 0;
  if (x >= 4294967295.0) return 4294967295u;
  return (uint32_t)x;
}

struct ImageTex : Texture {  // src/texture/image.rs:24-35
  uint32_t w, h;
  const uint8_t* px;
  Color value(double u, double v, Point3) const override {
    uint32_t i = std::min(as_u32(u * (double)w), w - 1);
    // `h - (v*h) as u32` on u32: wraps in release builds when v > 1; clamp then bounds it
    uint32_t j = std::min((uint32_t)(h - as_u32(v * (double)h)), h - 1);
    const uint8_t* p = px + ((size_t)j * w + i) * 3;
    double scale = 1.0 / 255.0;  // src/geom/vec3.rs:289-297
    return Color((double)p[0] * scale, (double)p[1] * scale, (double)p[2] * scale);
  }
};

// --------------------------------------------------------------- Materials
struct HitRecord {  // src/hittable/mod.rs:32-68
  Point3 p;
  Vec3 normal;
  int material = -1;
  double t = 0.0;
  bool front_face = false;
  double u = 0.0, v = 0.0;
  uint32_t object = 0xFFFFFFFFu;  // not in the reference: top-level object index, for id parity
  void set_face_normal(const Ray& ray, Vec3 outward_normal) {
    front_face = dot(ray.direction, outward_normal) < 0.0;
    normal = front_face ? outward_normal : -outward_normal;
  }
  static HitRecord make(Point3 p, double t, const Ray& ray, Vec3 outward, int material, double u,
                        double v) {
    HitRecord r;
    r.p = p;
    r.material = material;
    r.t = t;
    r.u = u;
    r.v = v;
    r.set_face_normal(ray, outward);
    return r;
  }
};

struct ScatterResult {  // src/material/mod.rs:29-32
  Color attenuation;
  bool scattered = false;
  Ray ray;
};

struct Material {  // src/material/mod.rs:16-27
  int kind = 0;
  virtual ~Material() {}
  virtual ScatterResult scatter(const Ray&, const HitRecord&, const DrawCtx&) const {
    return ScatterResult();  // black, None
  }
  virtual Color emitted(double, double, Point3) const { return Color(); }
};

struct Lambertian : Material {  // src/material/lambertian.rs:13-25
  const Texture* albedo;
  ScatterResult scatter(const Ray& ray_in, const HitRecord& rec, const DrawCtx& d) const override {
    Vec3 dir = rec.normal + rand_unit_vector(d);
    if (is_near_zero(dir)) dir = rec.normal;
    ScatterResult s;
    s.attenuation = albedo->value(rec.u, rec.v, rec.p);
    s.scattered = true;
    s.ray = Ray{rec.p, dir, ray_in.time};
    return s;
  }
};

struct Metal : Material {  // src/material/metal.rs:12-28
  Color albedo;
  double fuzziness;
  ScatterResult scatter(const Ray& ray_in, const HitRecord& rec, const DrawCtx& d) const override {
    Vec3 reflected = reflect(unit_vector(ray_in.direction), rec.normal);
    ScatterResult s;
    if (dot(reflected, rec.normal) > 0.0) {
      Vec3 direction = reflected + fuzziness * rand_unit_vector(d);
      s.scattered = true;
      s.ray = Ray{rec.p, direction, ray_in.time};
    }
    s.attenuation = albedo;
    return s;
  }
};

struct Dielectric : Material {  // src/material/dielectric.rs:14-46
  double ior;
  static double reflectance(double cosine, double ref_idx) {
    double r0 = powi2((1.0 - ref_idx) / (1.0 + ref_idx));
    return r0 + (1.0 - r0) * powi5(1.0 - cosine);
  }
  ScatterResult scatter(const Ray& ray_in, const HitRecord& rec, const DrawCtx& d) const override {
    double ratio = rec.front_face ? 1.0 / ior : ior;
    Vec3 ud = unit_vector(ray_in.direction);
    double cos_theta = fmin_rust(dot(-ud, rec.normal), 1.0);
    double sin_theta = std::sqrt(1.0 - powi2(cos_theta));
    bool cannot_refract = ratio * sin_theta > 1.0;
    U4 b = d.block(DRAW_FRESNEL, 0);
    bool should_reflect = reflectance(cos_theta, ratio) > u01(b.x, b.y);
    Vec3 direction = (cannot_refract || should_reflect) ? reflect(ud, rec.normal)
                                                        : refract(ud, rec.normal, ratio);
    ScatterResult s;
    s.attenuation = Color(1.0, 1.0, 1.0);
    s.scattered = true;
    s.ray = Ray{rec.p, direction, ray_in.time};
    return s;
  }
};

struct DiffuseLight : Material {  // src/material/emissive.rs:10-12
  const Texture* texture;
  Color emitted(double u, double v, Point3 p) const override { return texture->value(u, v, p); }
};

struct Isotropic : Material {  // src/texture/isotropic.rs:10-17
  const Texture* albedo;
  ScatterResult scatter(const Ray& ray_in, const HitRecord& rec, const DrawCtx& d) const override {
    ScatterResult s;
    s.attenuation = albedo->value(rec.u, rec.v, rec.p);
    s.scattered = true;
    s.ray = Ray{rec.p, rand_unit_vector(d), ray_in.time};
    return s;
  }
};

// --------------------------------------------------------------- Hittables
struct Counters {
  uint64_t v[RT_STAT_COUNT] = {0};
};

struct HitCtx {            // what a hit() needs besides the ray
  const DrawCtx* draw;     // null in the first-hit pass
  bool skip_media;         // first-hit pass leaves constant media out
  Counters* count;         // may be null
};

struct OptHit {
  bool some = false;
  HitRecord rec;
};

struct Hittable {  // src/hittable/mod.rs:26-30
  virtual ~Hittable() {}
  virtual OptHit hit(const Ray& ray, Range t_range, const HitCtx& cx) const = 0;
  virtual OptAabb bounding_box(Range time_range) const = 0;
};

struct Sphere : Hittable {  // src/hittable/sphere.rs
  Point3 center;
  double radius;
  int material;
  static void get_uv(Point3 p, double& u, double& v) {  // :20-24
    double theta = std::acos(-p.y());
    double phi = std::atan2(-p.z(), p.x()) + PI;
    u = phi / (2.0 * PI);
    v = theta / PI;
  }
  OptHit hit(const Ray& ray, Range t_range, const HitCtx& cx) const override {  // :28-65
    if (cx.count) cx.count->v[RT_STAT_SPHERE_TESTS]++;
    OptHit none;
    Vec3 oc = ray.origin - center;
    double a = length_squared(ray.direction);
    double half_b = dot(oc, ray.direction);
    double c = length_squared(oc) - powi2(radius);
    double discrim = powi2(half_b) - a * c;
    if (discrim < 0.0) return none;
    if (cx.count) cx.count->v[RT_STAT_SPHERE_ROOTS]++;
    double sqrtd = std::sqrt(discrim);
    double root = (-half_b - sqrtd) / a;
    if (!t_range.contains(root)) {
      root = (-half_b + sqrtd) / a;
      if (!t_range.contains(root)) return none;
    }
    double t = root;
    Point3 p = ray.along(t);
    Vec3 outward = (p - center) / radius;
    double u, v;
    get_uv(outward, u, v);
    OptHit h;
    h.some = true;
    h.rec = HitRecord::make(p, t, ray, outward, material, u, v);
    return h;
  }
  OptAabb bounding_box(Range) const override {  // :67-72
    OptAabb r;
    r.some = true;
    r.box = Aabb::make(center - Vec3(radius, radius, radius), center + Vec3(radius, radius, radius));
    return r;
  }
};

template <typename T>
T lerp(T a, T b, double t) {  // src/interpolate.rs:3-9
  return (1.0 - t) * a + t * b;
}

struct MovingSphere : Hittable {  // src/hittable/moving_sphere.rs
  Point3 c0, c1;
  double t0, t1, radius;
  int material;
  Point3 center_at(double time) const {  // :20-23
    double time_portion = (t0 - time) / (t1 - t0);
    return lerp(c0, c1, time_portion);
  }
  OptHit hit(const Ray& ray, Range t_range, const HitCtx& cx) const override {  // :27-35
    if (cx.count) {
      cx.count->v[RT_STAT_MSPHERE_TESTS]++;
      cx.count->v[RT_STAT_SPHERE_TESTS]--;  // the inner Sphere::hit counts itself
    }
    Sphere fixed;
    fixed.center = center_at(ray.time);
    fixed.radius = radius;
    fixed.material = material;
    return fixed.hit(ray, t_range, cx);
  }
  OptAabb bounding_box(Range tr) const override {  // :37-47
    Vec3 r(radius, radius, radius);
    Aabb b0 = Aabb::make(center_at(tr.start) - r, center_at(tr.start) + r);
    Aabb b1 = Aabb::make(center_at(tr.end) - r, center_at(tr.end) + r);
    return surrounding({b0, b1});
  }
};

struct AxisAlignedRect : Hittable {  // src/hittable/rect.rs
  Point3 center;
  double width, height;
  int axis;
  int material;
  Aabb bbox() const {  // :20-34
    double thickness = 0.001;
    double w2 = width / 2.0, h2 = height / 2.0;
    Vec3 offset = axis == 0 ? Vec3(thickness, w2, h2)
                            : axis == 1 ? Vec3(w2, thickness, h2) : Vec3(w2, h2, thickness);
    return Aabb::make(center - offset, center + offset);
  }
  OptHit hit(const Ray& ray, Range t_range, const HitCtx& cx) const override {  // :38-69
    if (cx.count) cx.count->v[RT_STAT_RECT_TESTS]++;
    OptHit none;
    Aabb bounds = bbox();
    Vec3 span = bounds.span();
    int d0 = axis, d1 = (d0 + 1) % 3, d2 = (d1 + 1) % 3;
    double t = (center[d0] - ray.origin[d0]) / ray.direction[d0];
    if (!t_range.contains(t)) return none;
    double i = ray.origin[d1] + t * ray.direction[d1];
    double j = ray.origin[d2] + t * ray.direction[d2];
    // RangeInclusive::contains: both ends included
    bool in1 = bounds.min[d1] <= i && i <= bounds.max[d1];
    bool in2 = bounds.min[d2] <= j && j <= bounds.max[d2];
    if (!in1 || !in2) return none;
    double u = (i - bounds.min[d1]) / span[d1];
    double v = (j - bounds.min[d2]) / span[d2];
    Vec3 outward(0, 0, 0);
    outward[d0] = 1.0;
    OptHit h;
    h.some = true;
    h.rec = HitRecord::make(ray.along(t), t, ray, outward, material, u, v);
    return h;
  }
  OptAabb bounding_box(Range) const override {
    OptAabb r;
    r.some = true;
    r.box = bbox();
    return r;
  }
};

struct HittableList : Hittable {  // src/hittable/hittable_list.rs
  std::vector<std::unique_ptr<Hittable>> objects;
  std::vector<uint32_t> ids;  // top-level object index per entry, or empty
  OptHit hit(const Ray& ray, Range t_range, const HitCtx& cx) const override {  // :23-32
    OptHit best;
    for (size_t i = 0; i < objects.size(); ++i) {
      OptHit h = objects[i]->hit(ray, t_range, cx);
      if (!h.some) continue;
      if (!ids.empty()) h.rec.object = ids[i];
      // min_by_key keeps the FIRST of equal minima
      if (!best.some || of_less(h.rec.t, best.rec.t)) best = h;
    }
    return best;
  }
  OptAabb bounding_box(Range tr) const override {  // :34-42
    std::vector<Aabb> boxes;
    for (const auto& o : objects) {
      OptAabb b = o->bounding_box(tr);
      if (!b.some) return OptAabb();
      boxes.push_back(b.box);
    }
    return surrounding(boxes);
  }
};

struct Cuboid : Hittable {  // src/hittable/cuboid.rs
  Point3 center;
  Vec3 half_size;
  HittableList sides;
  OptHit hit(const Ray& ray, Range t_range, const HitCtx& cx) const override {
    return sides.hit(ray, t_range, cx);
  }
  OptAabb bounding_box(Range) const override {  // :45-47
    OptAabb r;
    r.some = true;
    r.box = Aabb::make(center - half_size, center + half_size);
    return r;
  }
};

struct Translate : Hittable {  // src/hittable/instance.rs:5-26
  std::unique_ptr<Hittable> inner;
  Vec3 offset;
  OptHit hit(const Ray& ray, Range t_range, const HitCtx& cx) const override {
    if (cx.count) cx.count->v[RT_STAT_OP_TRANSLATE]++;
    Ray moved{ray.origin - offset, ray.direction, ray.time};
    OptHit h = inner->hit(moved, t_range, cx);
    if (h.some) {
      h.rec.p = h.rec.p + offset;
      h.rec.set_face_normal(moved, h.rec.normal);
    }
    return h;
  }
  OptAabb bounding_box(Range tr) const override {
    OptAabb b = inner->bounding_box(tr);
    if (b.some) b.box = Aabb::make(b.box.min + offset, b.box.max + offset);
    return b;
  }
};

struct RotateY : Hittable {  // src/hittable/instance.rs:28-103
  std::unique_ptr<Hittable> inner;
  double sin_theta, cos_theta;
  OptAabb zero_bbox;
  void finish() { zero_bbox = compute_bbox(Range{0.0, 0.0}); }  // RotateY::new :45
  OptHit hit(const Ray& ray, Range t_range, const HitCtx& cx) const override {  // :51-83
    if (cx.count) cx.count->v[RT_STAT_OP_ROTATE]++;
    Point3 origin = ray.origin;
    Vec3 direction = ray.direction;
    origin[0] = cos_theta * ray.origin.x() - sin_theta * ray.origin.z();
    origin[2] = sin_theta * ray.origin.x() + cos_theta * ray.origin.z();
    direction[0] = cos_theta * ray.direction.x() - sin_theta * ray.direction.z();
    direction[2] = sin_theta * ray.direction.x() + cos_theta * ray.direction.z();
    Ray rotated{origin, direction, ray.time};
    OptHit h = inner->hit(rotated, t_range, cx);
    if (h.some) {
      Point3 p = h.rec.p;
      Vec3 normal = h.rec.normal;
      // the reference uses .x() for BOTH terms (instance.rs:67-73); kept as is
      p[0] = cos_theta * h.rec.p.x() + sin_theta * h.rec.p.x();
      p[2] = -sin_theta * h.rec.p.x() + cos_theta * h.rec.p.x();
      normal[0] = cos_theta * h.rec.normal.x() + sin_theta * h.rec.normal.x();
      normal[2] = -sin_theta * h.rec.normal.x() + cos_theta * h.rec.normal.x();
      h.rec.p = p;
      h.rec.normal = normal;
      h.rec.set_face_normal(rotated, normal);
    }
    return h;
  }
  OptAabb compute_bbox(Range tr) const {  // :88-100
    OptAabb b = inner->bounding_box(tr);
    if (!b.some) return b;
    // Aabb::corners (src/geom/aabb.rs:76-84) then PointCloud::bounding_box (point_cloud.rs:8-16)
    double xs[2] = {b.box.min.x(), b.box.max.x()};
    double ys[2] = {b.box.min.y(), b.box.max.y()};
    double zs[2] = {b.box.min.z(), b.box.max.z()};
    Point3 mn(INF, INF, INF), mx(-INF, -INF, -INF);
    bool first = true;
    for (double x : xs)
      for (double y : ys)
        for (double z : zs) {
          Point3 c(cos_theta * x + sin_theta * z, y, -sin_theta * x + cos_theta * z);
          for (int a = 0; a < 3; ++a) {
            if (first || of_less(c[a], mn[a])) mn[a] = c[a];
            if (first || !of_less(c[a], mx[a])) mx[a] = c[a];
          }
          first = false;
        }
    OptAabb r;
    r.some = true;
    r.box.min = mn;
    r.box.max = mx;
    return r;
  }
  OptAabb bounding_box(Range tr) const override {  // :85-102
    if (zero_bbox.some && tr.start == 0.0 && tr.end == 0.0) return zero_bbox;
    return compute_bbox(tr);
  }
};

struct ConstantMedium : Hittable {  // src/hittable/constant_medium.rs
  std::unique_ptr<Hittable> boundary;
  double neg_inv_density;
  int phase_material;
  uint32_t object_index;  // addresses the free-flight draw
  OptHit hit(const Ray& ray, Range t_range, const HitCtx& cx) const override {  // :27-76
    OptHit none;
    if (cx.skip_media) return none;
    if (cx.count) cx.count->v[RT_STAT_MEDIUM_TESTS]++;
    OptHit r1 = boundary->hit(ray, Range{-INF, INF}, cx);
    if (!r1.some) return none;
    OptHit r2 = boundary->hit(ray, Range{r1.rec.t + 0.0001, INF}, cx);
    if (!r2.some) return none;
    double t1 = r1.rec.t, t2 = r2.rec.t;
    if (t1 < t_range.start) t1 = t_range.start;
    if (t2 > t_range.end) t2 = t_range.end;
    if (t1 >= t2) return none;
    if (t1 < 0.0) t1 = 0.0;
    double ray_length = length(ray.direction);
    double distance_inside_boundary = (t2 - t1) * ray_length;
    U4 b = cx.draw->block(DRAW_MEDIUM, object_index);
    double hit_distance = neg_inv_density * std::log10(u01(b.x, b.y));
    if (hit_distance > distance_inside_boundary) return none;
    double t = t1 + hit_distance / ray_length;
    Point3 p = ray.along(t);
    OptHit h;
    h.some = true;
    h.rec = HitRecord::make(p, t, ray, Vec3(1.0, 0.0, 0.0), phase_material, 0.0, 0.0);
    return h;
  }
  OptAabb bounding_box(Range tr) const override { return boundary->bounding_box(tr); }
};

const size_t MAX_PER_LEAF = 256;  // src/hittable/bvh.rs:9

struct BvhNode : Hittable {  // src/hittable/bvh.rs
  Range time_range{0, 0};
  OptAabb bbox;
  bool leaf = true;
  HittableList list;
  std::unique_ptr<BvhNode> left, right;

  struct Entry {
    std::unique_ptr<Hittable> h;
    uint32_t id;
  };

  static std::unique_ptr<BvhNode> build(Range tr, std::vector<Entry> hs, int axis) {  // :33-77
    std::unique_ptr<BvhNode> n(new BvhNode);
    n->time_range = tr;
    if (hs.size() < MAX_PER_LEAF) {
      for (auto& e : hs) {
        n->list.objects.push_back(std::move(e.h));
        n->list.ids.push_back(e.id);
      }
      n->bbox = n->list.bounding_box(tr);
      n->leaf = true;
      return n;
    }
    // sort_by_cached_key: stable, keys computed once
    std::vector<double> keys(hs.size());
    for (size_t i = 0; i < hs.size(); ++i) {
      OptAabb b = hs[i].h->bounding_box(tr);
      keys[i] = b.some ? (b.box.min[axis] + b.box.max[axis]) / 2.0 : 0.0;
    }
    std::vector<size_t> order(hs.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(),
                     [&](size_t a, size_t b) { return of_less(keys[a], keys[b]); });
    std::vector<Entry> sorted;
    sorted.reserve(hs.size());
    for (size_t i : order) sorted.push_back(std::move(hs[i]));
    size_t half = sorted.size() / 2;  // split_off(len/2): the upper part goes right
    std::vector<Entry> upper;
    for (size_t i = half; i < sorted.size(); ++i) upper.push_back(std::move(sorted[i]));
    sorted.resize(half);
    int next_axis = (axis + 1) % 3;
    n->right = build(tr, std::move(upper), next_axis);
    n->left = build(tr, std::move(sorted), next_axis);
    if (n->left->bbox.some && n->right->bbox.some)
      n->bbox = surrounding({n->left->bbox.box, n->right->bbox.box});
    n->leaf = false;
    return n;
  }

  OptHit hit(const Ray& ray, Range t_range, const HitCtx& cx) const override {  // :81-104
    if (bbox.some) {
      if (cx.count) cx.count->v[RT_STAT_BOX_TESTS]++;
      if (!bbox.box.intersect(ray, t_range)) return OptHit();
    }
    if (leaf) return list.hit(ray, t_range, cx);
    OptHit hl = left->hit(ray, t_range, cx);
    OptHit hr = right->hit(ray, t_range, cx);
    if (hl.some && hr.some) return hl.rec.t < hr.rec.t ? hl : hr;
    if (hr.some) return hr;
    return hl;
  }
  OptAabb bounding_box(Range tr) const override {  // :106-112
    if (time_range.start >= tr.start && time_range.end <= tr.end) return bbox;
    return OptAabb();
  }

  // Resolution order among equal-t candidates: right subtree before left, list order in a leaf.
  void ranks(std::vector<uint32_t>& out) const {
    if (leaf) {
      for (uint32_t id : list.ids) out.push_back(id);
    } else {
      right->ranks(out);
      left->ranks(out);
    }
  }
};

// ------------------------------------------------------------------- Scene
struct Camera {  // src/camera.rs:6-36
  Point3 origin, lower_left_corner;
  Vec3 horizontal, vertical, u, v;
  double lens_radius, time0, time1;
  Ray get_ray(double s, double t, const DrawCtx& d) const {
    Vec3 rd = lens_radius * rand_in_unit_disk(d);
    Vec3 offset = u * rd.x() + v * rd.y();
    double time;
    if (time1 > time0) {
      U4 b = d.block(DRAW_TIME, 0);
      time = time0 + (time1 - time0) * u01(b.x, b.y);  // gen_range(t0..t1)
    } else {
      time = time0;
    }
    return Ray{origin + offset,
               lower_left_corner + s * horizontal + t * vertical - origin - offset, time};
  }
  // Parity pass: no lens offset, time = shutter start.
  Ray get_ray_pinhole(double s, double t) const {
    Vec3 offset(0.0, 0.0, 0.0);
    return Ray{origin + offset,
               lower_left_corner + s * horizontal + t * vertical - origin - offset, time0};
  }
};

struct SceneData {
  std::vector<std::unique_ptr<Texture>> textures;
  std::vector<std::unique_ptr<Material>> materials;
  std::unique_ptr<BvhNode> world;
  // one hittable per flat-BVH item kind: [prim index] -> that primitive inside its
  // wrappers; [n_prims + object index] -> the whole object (constant media)
  std::vector<std::unique_ptr<Hittable>> prim_items;
  std::vector<std::unique_ptr<Hittable>> object_items;
  std::vector<uint32_t> prim_owner;
  // flat traversal only: the reference hierarchy's boxes from the root down to the leaf list
  // of every object, and which items answer to them (sides of a Cuboid that reach outside
  // Cuboid::bounding_box: the Y sides when size.x != size.z, cuboid.rs:21-30 and :45-47)
  std::vector<std::vector<Aabb>> chains;
  std::vector<uint32_t> chain_of_object;
  std::vector<uint8_t> prim_answers_to_chain, object_answers_to_chain;
  Camera camera;
  Color background;
  std::vector<uint8_t> texels;
  std::vector<rt_perlin> perlin;
  std::string error;
};

std::unique_ptr<Hittable> make_prim(const rt_prim& p) {
  if (p.kind == RT_PRIM_SPHERE) {
    auto* s = new Sphere;
    s->center = Point3(p.center);
    s->radius = p.radius;
    s->material = (int)p.material;
    return std::unique_ptr<Hittable>(s);
  }
  if (p.kind == RT_PRIM_MOVING_SPHERE) {
    auto* s = new MovingSphere;
    s->c0 = Point3(p.center);
    s->c1 = Point3(p.center1);
    s->t0 = p.time0;
    s->t1 = p.time1;
    s->radius = p.radius;
    s->material = (int)p.material;
    return std::unique_ptr<Hittable>(s);
  }
  auto* r = new AxisAlignedRect;
  r->center = Point3(p.center);
  r->width = p.width;
  r->height = p.height;
  r->axis = (int)p.axis;
  r->material = (int)p.material;
  return std::unique_ptr<Hittable>(r);
}

std::unique_ptr<Hittable> wrap_ops(std::unique_ptr<Hittable> h, const rt_op* ops, uint32_t n) {
  // ops are stored outermost first: apply from the innermost outwards
  for (uint32_t k = n; k-- > 0;) {
    const rt_op& op = ops[k];
    if (op.kind == RT_OP_TRANSLATE) {
      auto* t = new Translate;
      t->inner = std::move(h);
      t->offset = Vec3(op.a);
      h.reset(t);
    } else {
      auto* r = new RotateY;
      r->inner = std::move(h);
      r->sin_theta = op.a[0];
      r->cos_theta = op.a[1];
      r->finish();
      h.reset(r);
    }
  }
  return h;
}

// The base of an object: its single primitive, or the Cuboid of its six rects.
std::unique_ptr<Hittable> make_base(const rt_scene& s, const rt_object& o) {
  if (o.is_cuboid) {
    auto* c = new Cuboid;
    c->center = Point3(o.cuboid_center);
    c->half_size = Vec3(o.cuboid_half);
    for (uint32_t k = 0; k < o.prim_count; ++k)
      c->sides.objects.push_back(make_prim(s.prims[o.first_prim + k]));
    return std::unique_ptr<Hittable>(c);
  }
  if (o.prim_count == 1) return make_prim(s.prims[o.first_prim]);
  auto* l = new HittableList;
  for (uint32_t k = 0; k < o.prim_count; ++k) l->objects.push_back(make_prim(s.prims[o.first_prim + k]));
  return std::unique_ptr<Hittable>(l);
}

std::unique_ptr<Hittable> make_object(const rt_scene& s, uint32_t index) {
  const rt_object& o = s.objects[index];
  std::unique_ptr<Hittable> h = make_base(s, o);
  if (o.is_medium) {
    h = wrap_ops(std::move(h), s.ops + o.inner_first_op, o.inner_op_count);
    auto* m = new ConstantMedium;
    m->boundary = std::move(h);
    m->neg_inv_density = o.neg_inv_density;
    m->phase_material = (int)o.medium_material;
    m->object_index = index;
    h.reset(m);
  }
  return wrap_ops(std::move(h), s.ops + o.first_op, o.op_count);
}

SceneData* build_scene(const rt_scene& s) {
  std::unique_ptr<SceneData> d(new SceneData);
  d->texels.assign(s.texels, s.texels + s.n_texel_bytes);
  d->perlin.assign(s.perlin, s.perlin + s.n_perlin);
  for (uint32_t i = 0; i < s.n_textures; ++i) {
    const rt_texture& t = s.textures[i];
    if (t.kind == RT_TEX_SOLID) {
      auto* x = new SolidColor;
      x->c = Color(t.color);
      d->textures.emplace_back(x);
    } else if (t.kind == RT_TEX_CHECKER) {
      auto* x = new Checkerboard;
      x->even = d->textures[t.a].get();
      x->odd = d->textures[t.b].get();
      d->textures.emplace_back(x);
    } else if (t.kind == RT_TEX_PERLIN) {
      auto* x = new Perlin;
      x->t = &d->perlin[t.a];
      x->scale = t.scale;
      d->textures.emplace_back(x);
    } else {
      auto* x = new ImageTex;
      x->w = s.images[t.a].width;
      x->h = s.images[t.a].height;
      x->px = d->texels.data() + s.images[t.a].offset;
      d->textures.emplace_back(x);
    }
  }
  for (uint32_t i = 0; i < s.n_materials; ++i) {
    const rt_material& m = s.materials[i];
    Material* out = nullptr;
    if (m.kind == RT_MAT_LAMBERTIAN) {
      auto* x = new Lambertian;
      x->albedo = d->textures[m.texture].get();
      out = x;
    } else if (m.kind == RT_MAT_METAL) {
      auto* x = new Metal;
      x->albedo = Color(m.albedo);
      x->fuzziness = m.param;
      out = x;
    } else if (m.kind == RT_MAT_DIELECTRIC) {
      auto* x = new Dielectric;
      x->ior = m.param;
      out = x;
    } else if (m.kind == RT_MAT_DIFFUSE_LIGHT) {
      auto* x = new DiffuseLight;
      x->texture = d->textures[m.texture].get();
      out = x;
    } else {
      auto* x = new Isotropic;
      x->albedo = d->textures[m.texture].get();
      out = x;
    }
    out->kind = (int)m.kind;
    d->materials.emplace_back(out);
  }
  const rt_camera& c = s.camera;
  d->camera = Camera{Point3(c.origin), Point3(c.lower_left), Vec3(c.horizontal), Vec3(c.vertical),
                     Vec3(c.u),        Vec3(c.v),          c.lens_radius,      c.time0,
                     c.time1};
  d->background = Color(s.background);

  std::vector<BvhNode::Entry> entries;
  for (uint32_t i = 0; i < s.n_objects; ++i) {
    BvhNode::Entry e;
    e.h = make_object(s, i);
    e.id = i;
    entries.push_back(std::move(e));
  }
  // BvhNode::new(camera.shutter_time, hittables) (src/scene/loader.rs:86)
  d->world = BvhNode::build(Range{c.time0, c.time1}, std::move(entries), 0);

  // box chains of the reference hierarchy, for the flat traversal
  d->chain_of_object.assign(s.n_objects, 0);
  {
    std::vector<Aabb> path;
    std::function<void(const BvhNode&)> walk = [&](const BvhNode& n) {
      bool pushed = n.bbox.some;
      if (pushed) path.push_back(n.bbox.box);
      if (n.leaf) {
        for (uint32_t id : n.list.ids) d->chain_of_object[id] = (uint32_t)d->chains.size();
        d->chains.push_back(path);
      } else {
        walk(*n.right);
        walk(*n.left);
      }
      if (pushed) path.pop_back();
    };
    walk(*d->world);
  }
  d->prim_answers_to_chain.assign(s.n_prims, 0);
  d->object_answers_to_chain.assign(s.n_objects, 0);
  for (uint32_t i = 0; i < s.n_objects; ++i) {
    const rt_object& o = s.objects[i];
    if (!o.is_cuboid || o.cuboid_half[0] == o.cuboid_half[2]) continue;
    if (o.is_medium) {
      d->object_answers_to_chain[i] = 1;
    } else {
      for (uint32_t k = 0; k < o.prim_count; ++k)
        if (s.prims[o.first_prim + k].kind == RT_PRIM_RECT && s.prims[o.first_prim + k].axis == 1)
          d->prim_answers_to_chain[o.first_prim + k] = 1;
    }
  }

  // per-item hittables for the flat traversal
  d->prim_items.resize(s.n_prims);
  d->prim_owner.assign(s.n_prims, 0xFFFFFFFFu);
  d->object_items.resize(s.n_objects);
  for (uint32_t i = 0; i < s.n_objects; ++i) {
    const rt_object& o = s.objects[i];
    if (o.is_medium) {
      d->object_items[i] = make_object(s, i);
    } else {
      for (uint32_t k = 0; k < o.prim_count; ++k) {
        d->prim_items[o.first_prim + k] =
            wrap_ops(make_prim(s.prims[o.first_prim + k]), s.ops + o.first_op, o.op_count);
        d->prim_owner[o.first_prim + k] = i;
      }
    }
  }
  return d.release();
}

// ------------------------------------------------------------- render loop

// ray_color (src/main.rs:120-140), literally recursive.
Color ray_color(const SceneData& sc, const Ray& ray, uint32_t depth_budget, DrawCtx d, Counters* cnt) {
  if (depth_budget == 0) return Color();
  HitCtx cx{&d, false, cnt};
  if (cnt) cnt->v[RT_STAT_RAYS]++;
  OptHit h = sc.world->hit(ray, Range{0.001, INF}, cx);
  if (!h.some) return sc.background;
  const Material& m = *sc.materials[h.rec.material];
  ScatterResult sr = m.scatter(ray, h.rec, d);
  Color emitted = m.emitted(h.rec.u, h.rec.v, h.rec.p);
  if (sr.scattered) {
    DrawCtx next = d;
    next.bounce = d.bounce + 1;
    Color bounce_color = ray_color(sc, sr.ray, depth_budget - 1, next, cnt);
    return emitted + sr.attenuation * bounce_color;
  }
  return emitted;
}

struct FlatBvh {
  const rt_bvh_node* nodes = nullptr;
  const rt_bvh_item* items = nullptr;
  uint32_t n_nodes = 0, n_items = 0;
};

// Closest hit over the flattened (up to four-wide) BVH, walked the way the device
// walks it: a visit of an interior record slab-tests its consecutive children in
// f32 against [t_start, best t]; the hit children are ordered by (entry parameter
// with its low bits cleared, entry code); the walk goes on with the first and
// stacks the others, last first; a popped entry whose (truncated) entry parameter
// lies beyond the best t by now is dropped.  The root's own box is not tested.
// Ties on t resolve to the lower rank.
OptHit flat_hit(const SceneData& sc, const FlatBvh& bvh, const Ray& ray, Range t_range,
                const HitCtx& cx) {
  OptHit best;
  uint32_t best_rank = 0xFFFFFFFFu;
  if (bvh.n_nodes == 0) return best;
  // f32 slab constants per axis; every step is a correctly rounded IEEE f32
  // operation, repeated here exactly as the device performs it
  // (the plane a ray leaves a slab through carries the widening factor in its constants)
  const float widen = 1.000003814697265625f;  // 1 + 2^-18
  float inv_lo[3], add_lo[3], inv_hi[3], add_hi[3];
  for (int a = 0; a < 3; ++a) {
    float d32 = (float)ray.direction[a], o32 = (float)ray.origin[a];
    float iv = 1.0f / d32;
    if (!(std::fabs(iv) <= 1e25f)) iv = std::copysign(1e25f, d32);
    float n = -(o32 * iv);
    float e = std::fabs(n) * 4.76837158203125e-07f;  // 2^-21
    if (iv >= 0.0f) {
      inv_lo[a] = iv;
      add_lo[a] = n - e;
      inv_hi[a] = iv * widen;
      add_hi[a] = (n + e) * widen;
    } else {
      inv_lo[a] = iv * widen;
      add_lo[a] = (n + e) * widen;
      inv_hi[a] = iv;
      add_hi[a] = n - e;
    }
  }
  const float tmin32 = (float)t_range.start * 0.99f;
  const float finf = std::numeric_limits<float>::infinity();
  float best32w = finf;  // best t so far, times the widening factor
  // a stack word keeps only the leading bits of the entry parameter that fit beside the
  // entry code: the low (ebits - 1) float bits are cleared (rounds down)
  uint32_t ebits;
  {
    uint32_t hi = (bvh.n_nodes << 3) > (bvh.n_items << 4) ? (bvh.n_nodes << 3) : (bvh.n_items << 4);
    ebits = 1;
    while (ebits < 31 && (hi >> ebits) != 0) ++ebits;
  }
  auto truncated = [&](float t) {
    uint32_t u;
    std::memcpy(&u, &t, 4);
    u = (u >> (ebits - 1u)) << (ebits - 1u);
    std::memcpy(&t, &u, 4);
    return t;
  };
  struct Entry {
    uint32_t node;  // record index
    float t;
  };
  // what a record stands for, as one number: the device's second sort key
  auto entry_code = [](const rt_bvh_node& c) -> uint32_t {
    if (c.b & 0x80000000u) return ((c.a << 2) | (c.b & 3u)) << 1;
    return (((c.a << 3) | (c.b - 1u)) << 1) | 1u;
  };
  auto float_bits = [](float t) {
    uint32_t u;
    std::memcpy(&u, &t, 4);
    return u;
  };
  std::vector<Entry> stack;
  auto pop = [&](uint32_t& out) -> bool {
    while (!stack.empty()) {
      Entry e = stack.back();
      stack.pop_back();
      if (e.t <= best32w) {
        out = e.node;
        return true;
      }
    }
    return false;
  };
  uint32_t cur = 0;
  for (;;) {
    const rt_bvh_node& n = bvh.nodes[cur];
    bool have_next = false;
    uint32_t next = 0;
    if (n.b & 0x80000000u) {
      uint32_t kids = (n.b & 3u) + 1u;
      if (cx.count) cx.count->v[RT_STAT_NODE_VISITS]++;
      struct Hit {
        uint64_t key;
        Entry e;
      };
      Hit hits[4];
      uint32_t n_hits = 0;
      for (uint32_t k = 0; k < kids; ++k) {
        const rt_bvh_node& c = bvh.nodes[n.a + k];
        if (cx.count) cx.count->v[RT_STAT_BOX_TESTS]++;
        float lo3[3], hi3[3];
        for (int a = 0; a < 3; ++a) {
          float ta = std::fmaf(c.lo[a], inv_lo[a], add_lo[a]);
          float tb = std::fmaf(c.hi[a], inv_hi[a], add_hi[a]);
          lo3[a] = std::fmin(ta, tb);
          hi3[a] = std::fmax(ta, tb);
        }
        float t0 = std::fmax(std::fmax(lo3[0], lo3[1]), std::fmax(lo3[2], tmin32));
        float t1 = std::fmin(std::fmin(hi3[0], hi3[1]), std::fmin(hi3[2], best32w));
        if (!(t0 <= t1)) continue;
        float tt = truncated(t0);
        hits[n_hits].key = ((uint64_t)float_bits(tt) << 32) | entry_code(c);
        hits[n_hits].e = Entry{n.a + k, tt};
        ++n_hits;
      }
      std::sort(hits, hits + n_hits, [](const Hit& x, const Hit& y) { return x.key < y.key; });
      for (uint32_t k = n_hits; k > 1; --k) stack.push_back(hits[k - 1].e);
      if (n_hits) {
        next = hits[0].e.node;
        have_next = true;
      }
    } else {
      for (uint32_t k = 0; k < n.b; ++k) {
        const rt_bvh_item& it = bvh.items[n.a + k];
        const Hittable* h = it.prim == 0xFFFFFFFFu ? sc.object_items[it.object].get()
                                                   : sc.prim_items[it.prim].get();
        if (!h) continue;
        OptHit r = h->hit(ray, t_range, cx);
        if (!r.some) continue;
        // geometry outside its object's box: the reference finds it only through its own boxes
        bool answers = it.prim == 0xFFFFFFFFu ? sc.object_answers_to_chain[it.object]
                                              : sc.prim_answers_to_chain[it.prim];
        if (answers) {
          bool pass = true;
          for (const Aabb& b : sc.chains[sc.chain_of_object[it.object]])
            pass = pass && b.intersect(ray, t_range);
          if (!pass) continue;
        }
        r.rec.object = it.object;
        if (!best.some || r.rec.t < best.rec.t || (r.rec.t == best.rec.t && it.rank < best_rank)) {
          best = r;
          best_rank = it.rank;
          best32w = (float)r.rec.t * widen;
        }
      }
    }
    if (!have_next && !pop(next)) break;
    cur = next;
  }
  return best;
}

struct World {
  const SceneData* sc;
  const FlatBvh* flat;  // null: the reference topology
  OptHit hit(const Ray& ray, Range r, const HitCtx& cx) const {
    return flat ? flat_hit(*sc, *flat, ray, r, cx) : sc->world->hit(ray, r, cx);
  }
};

// The same estimator unrolled the way the device evaluates it:
// L = sum_k beta_k * emitted_k (+ beta * background on a miss), beta_k the running
// product of attenuations.  Differs from the recursion only in association.
Color ray_color_iterative(const World& w, Ray ray, uint32_t max_depth, DrawCtx d, Counters* cnt) {
  const SceneData& sc = *w.sc;
  Color L(0, 0, 0), beta(1, 1, 1);
  for (uint32_t bounce = 0; bounce < max_depth; ++bounce) {
    d.bounce = bounce;
    HitCtx cx{&d, false, cnt};
    if (cnt) cnt->v[RT_STAT_RAYS]++;
    OptHit h = w.hit(ray, Range{0.001, INF}, cx);
    if (!h.some) {
      if (cnt) cnt->v[RT_STAT_SHADE_MISS]++;
      L = L + beta * sc.background;
      break;
    }
    const Material& m = *sc.materials[h.rec.material];
    if (cnt) {
      cnt->v[RT_STAT_HITS]++;
      cnt->v[RT_STAT_SHADE_LAMBERTIAN + m.kind]++;
    }
    if (m.kind == RT_MAT_DIFFUSE_LIGHT) L = L + beta * m.emitted(h.rec.u, h.rec.v, h.rec.p);
    ScatterResult sr = m.scatter(ray, h.rec, d);
    if (!sr.scattered) break;
    beta = beta * sr.attenuation;
    ray = sr.ray;
  }
  return L;
}

struct RenderJob {
  const SceneData* sc;
  const FlatBvh* flat;
  rt_render_params p;
  bool recursive;
};

Color sample_radiance(const RenderJob& job, uint32_t x, uint32_t y, uint32_t s, Counters* cnt) {
  const rt_render_params& p = job.p;
  DrawCtx d{p.seed, y * p.width + x, s, 0};
  U4 b = d.block(DRAW_JITTER, 0);
  double jx = u01(b.x, b.y), jy = u01(b.z, b.w);
  // src/main.rs:91-93
  double u = ((double)x + jx) / ((double)p.width - 1.0);
  double v = ((double)(p.height - y) + jy) / ((double)p.height - 1.0);
  Ray ray = job.sc->camera.get_ray(u, v, d);
  if (job.recursive && !job.flat) return ray_color(*job.sc, ray, p.max_depth, d, cnt);
  World w{job.sc, job.flat};
  return ray_color_iterative(w, ray, p.max_depth, d, cnt);
}

void window_of(const rt_render_params& p, uint32_t& x0, uint32_t& y0, uint32_t& w, uint32_t& h) {
  if (p.window.w == 0 || p.window.h == 0) {
    x0 = y0 = 0;
    w = p.width;
    h = p.height;
  } else {
    x0 = p.window.x0;
    y0 = p.window.y0;
    w = p.window.w;
    h = p.window.h;
  }
}

int run_render(const RenderJob& job, double* rgb_sum, uint64_t* stats, int n_threads) {
  uint32_t x0, y0, w, h;
  window_of(job.p, x0, y0, w, h);
  if (x0 + w > job.p.width || y0 + h > job.p.height) return RT_ERR_INVALID;
  if (n_threads < 1) n_threads = 1;
  std::atomic<uint32_t> next_row(0);
  std::vector<Counters> counters((size_t)n_threads);
  auto work = [&](int tid) {
    Counters* cnt = stats ? &counters[(size_t)tid] : nullptr;
    for (;;) {
      uint32_t row = next_row.fetch_add(1);
      if (row >= h) break;
      for (uint32_t col = 0; col < w; ++col) {
        // .sum() over samples in order (src/main.rs:97; Sum = reduce(a + b), vec3.rs:227-231)
        Color acc;
        bool first = true;
        for (uint32_t s = job.p.sample_begin; s < job.p.sample_end; ++s) {
          Color c = sample_radiance(job, x0 + col, y0 + row, s, cnt);
          acc = first ? c : acc + c;
          first = false;
        }
        double* out = rgb_sum + ((size_t)row * w + col) * 3;
        out[0] = acc.x();
        out[1] = acc.y();
        out[2] = acc.z();
      }
    }
  };
  if (n_threads == 1) {
    work(0);
  } else {
    std::vector<std::thread> ts;
    for (int t = 0; t < n_threads; ++t) ts.emplace_back(work, t);
    for (auto& t : ts) t.join();
  }
  if (stats) {
    for (int k = 0; k < RT_STAT_COUNT; ++k) {
      stats[k] = 0;
      for (const auto& c : counters) stats[k] += c.v[k];
    }
  }
  return RT_OK;
}

}  // namespace

// ------------------------------------------------------------------- C API
extern "C" {

struct oracle_scene {
  SceneData* d;
};

// A HitRecord plus the object id, as plain data.
struct oracle_hit {
  double p[3];
  double normal[3];
  double t, u, v;
  uint32_t object;
  int32_t material;
  uint32_t front_face;
  uint32_t hit;
};

int oracle_create(const rt_scene* scene, oracle_scene** out) {
  if (!scene || !out) return RT_ERR_INVALID;
  *out = new oracle_scene{build_scene(*scene)};
  return RT_OK;
}

void oracle_destroy(oracle_scene* o) {
  if (o) {
    delete o->d;
    delete o;
  }
}

static void fill_hit(const OptHit& h, oracle_hit* out) {
  std::memset(out, 0, sizeof *out);
  out->object = 0xFFFFFFFFu;
  out->material = -1;
  if (!h.some) return;
  out->hit = 1;
  for (int i = 0; i < 3; ++i) {
    out->p[i] = h.rec.p[i];
    out->normal[i] = h.rec.normal[i];
  }
  out->t = h.rec.t;
  out->u = h.rec.u;
  out->v = h.rec.v;
  out->object = h.rec.object;
  out->material = h.rec.material;
  out->front_face = h.rec.front_face ? 1 : 0;
}

// world.hit(ray, t_start..t_end) through the reference-topology BVH.  `draw` may be
// null when no constant medium can be reached (then media are skipped).
int oracle_world_hit(const oracle_scene* o, const double origin[3], const double direction[3],
                     double time, double t_start, double t_end, uint64_t seed, uint32_t pixel,
                     uint32_t sample, uint32_t bounce, int with_media, oracle_hit* out) {
  if (!o || !out) return RT_ERR_INVALID;
  Ray ray{Point3(origin), Vec3(direction), time};
  DrawCtx d{seed, pixel, sample, bounce};
  HitCtx cx{&d, with_media == 0, nullptr};
  fill_hit(o->d->world->hit(ray, Range{t_start, t_end}, cx), out);
  return RT_OK;
}

// One Material::scatter + emitted for material index `material`.
int oracle_scatter(const oracle_scene* o, int32_t material, const double origin[3],
                   const double direction[3], double time, const oracle_hit* rec, uint64_t seed,
                   uint32_t pixel, uint32_t sample, uint32_t bounce, double attenuation[3],
                   double emitted[3], double out_origin[3], double out_direction[3],
                   int* scattered) {
  if (!o || material < 0 || (size_t)material >= o->d->materials.size()) return RT_ERR_INVALID;
  Ray ray{Point3(origin), Vec3(direction), time};
  HitRecord r;
  r.p = Point3(rec->p);
  r.normal = Vec3(rec->normal);
  r.t = rec->t;
  r.u = rec->u;
  r.v = rec->v;
  r.front_face = rec->front_face != 0;
  r.material = material;
  DrawCtx d{seed, pixel, sample, bounce};
  const Material& m = *o->d->materials[(size_t)material];
  ScatterResult s = m.scatter(ray, r, d);
  Color e = m.emitted(r.u, r.v, r.p);
  for (int i = 0; i < 3; ++i) {
    attenuation[i] = s.attenuation[i];
    emitted[i] = e[i];
    out_origin[i] = s.ray.origin[i];
    out_direction[i] = s.ray.direction[i];
  }
  *scattered = s.scattered ? 1 : 0;
  return RT_OK;
}

int oracle_texture_value(const oracle_scene* o, uint32_t texture, double u, double v,
                         const double p[3], double out[3]) {
  if (!o || texture >= o->d->textures.size()) return RT_ERR_INVALID;
  Color c = o->d->textures[texture]->value(u, v, Point3(p));
  for (int i = 0; i < 3; ++i) out[i] = c[i];
  return RT_OK;
}

// Camera::get_ray for pixel (x, y), sample s (jitter included), as the render loop calls it.
int oracle_camera_ray(const oracle_scene* o, const rt_render_params* p, uint32_t x, uint32_t y,
                      uint32_t s, double origin[3], double direction[3], double* time) {
  if (!o || !p) return RT_ERR_INVALID;
  DrawCtx d{p->seed, y * p->width + x, s, 0};
  U4 b = d.block(DRAW_JITTER, 0);
  double jx = u01(b.x, b.y), jy = u01(b.z, b.w);
  double u = ((double)x + jx) / ((double)p->width - 1.0);
  double v = ((double)(p->height - y) + jy) / ((double)p->height - 1.0);
  Ray r = o->d->camera.get_ray(u, v, d);
  for (int i = 0; i < 3; ++i) {
    origin[i] = r.origin[i];
    direction[i] = r.direction[i];
  }
  *time = r.time;
  return RT_OK;
}

// Deterministic first-hit pass (pixel-centre pinhole rays, media skipped), rows dealt out to
// n_threads threads (each pixel is independent, so the thread count cannot change a result).
int oracle_first_hit_mt(const oracle_scene* o, uint32_t width, uint32_t height,
                        const rt_window* window, int n_threads, uint32_t* id, double* t,
                        double* normal) {
  if (!o || !id || !t || !normal) return RT_ERR_INVALID;
  rt_render_params p{};
  p.width = width;
  p.height = height;
  if (window) p.window = *window;
  uint32_t x0, y0, w, h;
  window_of(p, x0, y0, w, h);
  if (x0 + w > width || y0 + h > height) return RT_ERR_INVALID;
  std::atomic<uint32_t> next_row{0};
  auto work = [&]() {
    HitCtx cx{nullptr, true, nullptr};
    for (uint32_t row = next_row.fetch_add(1); row < h; row = next_row.fetch_add(1))
      for (uint32_t col = 0; col < w; ++col) {
        uint32_t x = x0 + col, y = y0 + row;
        double u = ((double)x + 0.5) / ((double)width - 1.0);
        double v = ((double)(height - y) + 0.5) / ((double)height - 1.0);
        Ray ray = o->d->camera.get_ray_pinhole(u, v);
        OptHit hr = o->d->world->hit(ray, Range{0.001, INF}, cx);
        size_t i = (size_t)row * w + col;
        id[i] = hr.some ? hr.rec.object : 0xFFFFFFFFu;
        t[i] = hr.some ? hr.rec.t : 0.0;
        for (int k = 0; k < 3; ++k) normal[i * 3 + k] = hr.some ? hr.rec.normal[k] : 0.0;
      }
  };
  std::vector<std::thread> pool;
  for (int k = 1; k < n_threads; ++k) pool.emplace_back(work);
  work();
  for (std::thread& th : pool) th.join();
  return RT_OK;
}
int oracle_first_hit(const oracle_scene* o, uint32_t width, uint32_t height,
                     const rt_window* window, uint32_t* id, double* t, double* normal) {
  return oracle_first_hit_mt(o, width, height, window, 1, id, t, normal);
}

// mode 0: reference topology, recursive ray_color (the literal restatement)
// mode 1: reference topology, iterative estimator
// mode 2: flattened BVH (nodes/items required), iterative estimator
int oracle_render(const oracle_scene* o, const rt_render_params* p, int mode,
                  const rt_bvh_node* nodes, uint32_t n_nodes, const rt_bvh_item* items,
                  uint32_t n_items, int n_threads, double* rgb_sum, uint64_t* stats) {
  if (!o || !p || !rgb_sum) return RT_ERR_INVALID;
  if (p->sample_end < p->sample_begin) return RT_ERR_INVALID;
  FlatBvh flat;
  RenderJob job{o->d, nullptr, *p, mode == 0};
  if (mode == 2) {
    if (!nodes || !items) return RT_ERR_INVALID;
    for (uint32_t i = 0; i < n_items; ++i) {
      const rt_bvh_item& it = items[i];
      if (it.object >= o->d->object_items.size()) return RT_ERR_INVALID;
      if (it.prim != 0xFFFFFFFFu && it.prim >= o->d->prim_items.size()) return RT_ERR_INVALID;
    }
    flat.nodes = nodes;
    flat.items = items;
    flat.n_nodes = n_nodes;
    flat.n_items = n_items;
    job.flat = &flat;
  }
  return run_render(job, rgb_sum, stats, n_threads);
}

// Tie-break ranks of the reference topology: rank_of_object[i] = position of object i
// in the order in which the reference would prefer equal-t hits.
int oracle_ranks(const oracle_scene* o, uint32_t* rank_of_object, uint32_t n) {
  if (!o || !rank_of_object) return RT_ERR_INVALID;
  std::vector<uint32_t> order;
  o->d->world->ranks(order);
  if (order.size() != n) return RT_ERR_INVALID;
  for (uint32_t r = 0; r < n; ++r) rank_of_object[order[r]] = r;
  return RT_OK;
}

// Shape of the reference BVH: node count, leaf count, largest leaf.
int oracle_bvh_shape(const oracle_scene* o, uint32_t* n_nodes, uint32_t* n_leaves,
                     uint32_t* max_leaf) {
  if (!o) return RT_ERR_INVALID;
  uint32_t nodes = 0, leaves = 0, biggest = 0;
  std::vector<const BvhNode*> st{o->d->world.get()};
  while (!st.empty()) {
    const BvhNode* n = st.back();
    st.pop_back();
    ++nodes;
    if (n->leaf) {
      ++leaves;
      biggest = std::max<uint32_t>(biggest, (uint32_t)n->list.objects.size());
    } else {
      st.push_back(n->left.get());
      st.push_back(n->right.get());
    }
  }
  *n_nodes = nodes;
  *n_leaves = leaves;
  *max_leaf = biggest;
  return RT_OK;
}

// ---- small pieces exported for known-answer tests ----

void oracle_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  U4 r = philox4x32_10(U4{ctr[0], ctr[1], ctr[2], ctr[3]}, key[0], key[1]);
  out[0] = r.x;
  out[1] = r.y;
  out[2] = r.z;
  out[3] = r.w;
}

// Vec3::project, the reference's own unit test subject (src/geom/vec3.rs:299-309).
void oracle_vec3_project(const double a[3], const double b[3], double out[3]) {
  Vec3 r = project(Vec3(a), Vec3(b));
  for (int i = 0; i < 3; ++i) out[i] = r[i];
}

// Aabb::new + Aabb::surrounding, the reference's other unit test (src/geom/aabb.rs:87-104).
// boxes: n x 6 doubles (corner a, corner b); out: min xyz, max xyz.  Returns 1 for Some.
int oracle_aabb_surrounding(const double* boxes, uint32_t n, double out[6]) {
  std::vector<Aabb> parts;
  for (uint32_t i = 0; i < n; ++i)
    parts.push_back(Aabb::make(Point3(boxes + 6 * i), Point3(boxes + 6 * i + 3)));
  OptAabb r = surrounding(parts);
  if (!r.some) return 0;
  for (int i = 0; i < 3; ++i) {
    out[i] = r.box.min[i];
    out[3 + i] = r.box.max[i];
  }
  return 1;
}

int oracle_aabb_intersect(const double bmin[3], const double bmax[3], const double origin[3],
                          const double direction[3], double t_start, double t_end) {
  Aabb b;
  b.min = Point3(bmin);
  b.max = Point3(bmax);
  Ray r{Point3(origin), Vec3(direction), 0.0};
  return b.intersect(r, Range{t_start, t_end}) ? 1 : 0;
}

void oracle_reflect(const double v[3], const double n[3], double out[3]) {
  Vec3 r = reflect(Vec3(v), Vec3(n));
  for (int i = 0; i < 3; ++i) out[i] = r[i];
}
void oracle_refract(const double v[3], const double n[3], double eta, double out[3]) {
  Vec3 r = refract(Vec3(v), Vec3(n), eta);
  for (int i = 0; i < 3; ++i) out[i] = r[i];
}
double oracle_reflectance(double cosine, double ref_idx) {
  return Dielectric::reflectance(cosine, ref_idx);
}

// MovingSphere::center_at for a single-primitive scene's prim 0 style parameters.
void oracle_moving_center(const double c0[3], const double c1[3], double t0, double t1, double time,
                          double out[3]) {
  MovingSphere m;
  m.c0 = Point3(c0);
  m.c1 = Point3(c1);
  m.t0 = t0;
  m.t1 = t1;
  m.radius = 1.0;
  m.material = 0;
  Point3 c = m.center_at(time);
  for (int i = 0; i < 3; ++i) out[i] = c[i];
}

// Color::into_srgb8 (src/geom/vec3.rs:123-133) for one pixel.
void oracle_into_srgb8(const double rgb_sum[3], uint32_t samples_per_pixel, uint8_t out[3]) {
  double scale = 1.0 / (double)samples_per_pixel;
  double max = 255.0 / 256.0;
  for (int i = 0; i < 3; ++i) {
    double c = rgb_sum[i] * scale;
    // f64::clamp: NaN stays NaN
    if (c < 0.0) c = 0.0;
    if (c > max) c = max;
    double q = c * 256.0;
    // `as u8`: saturating, NaN -> 0
    uint8_t b;
    if (!(q == q)) b = 0;
    else if (q <= 0.0) b = 0;
    else if (q >= 255.0) b = 255;
    else b = (uint8_t)q;
    out[i] = b;
  }
}

// Camera::build()...done() (src/camera.rs:52-86), independent of the host library's copy.
int oracle_camera_build(const rt_camera_desc* d, uint32_t width, uint32_t height, rt_camera* out) {
  if (!d || !out) return RT_ERR_INVALID;
  Point3 look_from(d->look_from);
  Point3 look_at = d->has_look_at ? Point3(d->look_at) : Point3();
  Vec3 look_vector = look_from - look_at;
  if (is_near_zero(look_vector)) return RT_ERR_INVALID;
  double focus_dist = d->has_focus_distance ? d->focus_distance : length(look_at - look_from);
  double aspect_ratio = (double)width / (double)height;
  double theta = d->vertical_fov * (PI / 180.0);
  double h = std::tan(theta / 2.0);
  double viewport_height = 2.0 * h;
  double viewport_width = aspect_ratio * viewport_height;
  Vec3 v_up = d->has_v_up ? Vec3(d->v_up) : Vec3(0.0, 1.0, 0.0);
  Vec3 w = unit_vector(look_vector);
  Vec3 u = unit_vector(cross(v_up, w));
  Vec3 v = cross(w, u);
  Vec3 horizontal = focus_dist * viewport_width * u;
  Vec3 vertical = focus_dist * viewport_height * v;
  Point3 llc = look_from - horizontal / 2.0 - vertical / 2.0 - focus_dist * w;
  for (int i = 0; i < 3; ++i) {
    out->origin[i] = look_from[i];
    out->lower_left[i] = llc[i];
    out->horizontal[i] = horizontal[i];
    out->vertical[i] = vertical[i];
    out->u[i] = u[i];
    out->v[i] = v[i];
  }
  out->lens_radius = d->aperture / 2.0;
  out->time0 = d->has_shutter ? d->shutter0 : 0.0;
  out->time1 = d->has_shutter ? d->shutter1 : 0.0;
  return RT_OK;
}

int oracle_hardware_threads(void) {
  unsigned n = std::thread::hardware_concurrency();
  return n ? (int)n : 1;
}

}  // extern "C"
