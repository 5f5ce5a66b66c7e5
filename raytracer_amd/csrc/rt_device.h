// Device-side geometry, traversal, materials and textures.
//
// Arithmetic that decides WHAT the reference would compute (primitive tests,
// hit records, scattering, textures) is written operation for operation in f64
// after the cited reference lines and compiled without FMA contraction; the
// box culling in front of it is free to use FMA because it only has to be
// conservative.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_scene.h"
#include "philox.h"

namespace rt {

#define RT_DEV __device__ __forceinline__

struct DeviceScene {
  const float4* nodes;   // 2 x float4 per node
  const DPrim* prims;
  const uint32_t* ranks;
  const DPrim* boundary;
  const DOp* ops;
  const DAux* aux;
  const DMaterial* materials;
  const DTexture* textures;
  const rt_perlin* perlin;
  const DImage* images;
  const uint8_t* texels;
  const double* chain_boxes;      // {min xyz, max xyz} per box of the reference's hierarchy
  const uint32_t* chain_ranges;   // {first, count} per chain
  uint32_t n_nodes;
  uint32_t has_media;
  uint32_t has_lights;  // some material is a DiffuseLight (else a path's L stays 0 until its last event)
  uint32_t root_code;  // traversal entry code of the root record
  // wrapper ops [0, n_ops_lds) have a copy in the workgroup's dynamic LDS, ops_lds_offset bytes in
  // (set by a kernel after it staged them; 0 ops for kernels that stage nothing)
  uint32_t ops_lds_offset, n_ops_lds;
  rt_camera cam;
  double bg[3];
};

// ------------------------------------------------------------------- Vec3
// Operators spelled as src/geom/vec3.rs:141-225 spells them.
struct V3 {
  double x, y, z;
};
RT_DEV V3 v3(double x, double y, double z) { return V3{x, y, z}; }
RT_DEV V3 v3(const double* p) { return V3{p[0], p[1], p[2]}; }
RT_DEV V3 vneg(V3 a) { return V3{-a.x, -a.y, -a.z}; }
RT_DEV V3 vadd(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
RT_DEV V3 vsub(V3 a, V3 b) { return vadd(a, vneg(b)); }  // a + (-b)
RT_DEV V3 vscale(V3 a, double s) { return V3{a.x * s, a.y * s, a.z * s}; }
RT_DEV V3 vmul(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
RT_DEV V3 vdiv(V3 a, double s) { return vscale(a, 1.0 / s); }  // a * s.recip()
RT_DEV double vdot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RT_DEV double vlen2(V3 a) { return vdot(a, a); }
RT_DEV double vlen(V3 a) { return sqrt(vlen2(a)); }
RT_DEV V3 vunit(V3 a) { return vdiv(a, vlen(a)); }
RT_DEV V3 vreflect(V3 v, V3 n) { return vsub(v, vscale(n, 2.0 * vdot(v, n))); }  // vec3.rs:83-85
RT_DEV V3 vrefract(V3 v, V3 n, double eta) {                                      // vec3.rs:87-92
  double cos_theta = fmin(vdot(vneg(v), n), 1.0);
  V3 perp = vscale(vadd(v, vscale(n, cos_theta)), eta);
  V3 par = vscale(vneg(n), sqrt(fabs(1.0 - vlen2(perp))));
  return vadd(perp, par);
}
RT_DEV bool vnear_zero(V3 v) {  // vec3.rs:118-121
  const double LIMIT = 0.001;
  return fabs(v.x) < LIMIT && fabs(v.y) < LIMIT && fabs(v.z) < LIMIT;
}
RT_DEV double vget(V3 v, uint32_t axis) { return axis == 0 ? v.x : (axis == 1 ? v.y : v.z); }
RT_DEV double powi5(double x) {  // powi(5) as compiler-rt expands it: x * (x^2)^2
  double x2 = x * x;
  double x4 = x2 * x2;
  return x * x4;
}

// ---------------------------------------------------------------- drawing
struct DrawKey {
  uint64_t seed;
  uint32_t pixel, sample, bounce;
};
RT_DEV u32x4 draw(const DrawKey& k, uint32_t purpose, uint32_t seq) {
  return draw_block(k.seed, k.pixel, k.sample, k.bounce, purpose, seq);
}
#ifndef RT_MAX_REJECTIONS
#define RT_MAX_REJECTIONS 4096
#endif
const uint32_t kMaxRejections = RT_MAX_REJECTIONS;  // the loop ends with probability 1 - 2^-4000 long before

// (rand_unit_vector, vec3.rs:35-47, is wave_rand_in_unit_sphere + shade_finish below)
RT_DEV V3 rand_in_unit_disk(const DrawKey& k) {  // vec3.rs:49-60
  V3 v = v3(0.0, 0.0, 0.0);
  for (uint32_t it = 0; it < kMaxRejections; ++it) {
    double unused;
    sym3_42(draw(k, DRAW_LENS, it), v.x, v.y, unused);
    v.z = 0.0;
    if (vlen2(v) < 1.0) break;
  }
  return v;
}

// ------------------------------------------------------------- statistics
#ifdef RT_LANE_PROFILE
// Development build only: per loop, how many wave-level trips ran and how many
// lanes were active in them (trips[2k] = lane trips, trips[2k+1] = wave trips).
// LP_T_*: phases of one trip of trace_kernel; there trips[2k] = clock ticks summed over waves.
enum { LP_RAY = 0, LP_OUTER, LP_NODE, LP_LEAFPRIM, LP_POP, LP_SHADE, LP_REJECT, LP_ROOTS,
       LP_T_SETUP, LP_T_TRAVERSE, LP_T_SHADE, LP_T_VECTOR, LP_T_STORE, LP_COUNT };
#endif
struct Stats {
  uint32_t v[RT_STAT_COUNT];
#ifdef RT_LANE_PROFILE
  uint32_t trips[2 * LP_COUNT];
#endif
  RT_DEV void clear() {
#pragma unroll
    for (int i = 0; i < RT_STAT_COUNT; ++i) v[i] = 0;
#ifdef RT_LANE_PROFILE
#pragma unroll
    for (int i = 0; i < 2 * LP_COUNT; ++i) trips[i] = 0;
#endif
  }
};
#ifdef RT_LANE_PROFILE
RT_DEV void lane_trip(Stats& st, int loop) {
  unsigned long long m = __ballot(1);
  st.trips[2 * loop]++;
  if ((int)__lane_id() == __ffsll((long long)m) - 1) st.trips[2 * loop + 1]++;
}
RT_DEV void lane_tick(Stats& st, int phase, unsigned long long& t0) {
  unsigned long long t1 = __builtin_readcyclecounter();
  unsigned long long m = __ballot(1);
  if ((int)__lane_id() == __ffsll((long long)m) - 1) {
    st.trips[2 * phase] += (uint32_t)(t1 - t0);
    st.trips[2 * phase + 1]++;
  }
  t0 = t1;
}
#define RT_LP(st, loop) lane_trip(st, loop)
#define RT_TICK(st, phase, t0) lane_tick(st, phase, t0)
#else
#define RT_LP(st, loop) ((void)0)
#define RT_TICK(st, phase, t0) ((void)0)
#endif
template <bool ON>
RT_DEV void stat(Stats& s, int which) {
  if (ON) s.v[which]++;
}

// ---------------------------------------------------------------- records
RT_DEV uint32_t ops_first(uint32_t w) { return w & 0x00FFFFFFu; }
RT_DEV uint32_t ops_count(uint32_t w) { return w >> 24; }

// The pointers carry the LDS address space: a generic pointer that may be LDS or
// global turns every access into a flat instruction, several times slower to
// return than a ds_read.
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) f32x4 lds_f32x4;
typedef __attribute__((address_space(3))) uint32_t lds_u32;

// Wrapper op `index`: from the workgroup's LDS copy when the scene's ops were staged there
// (scenes with a few dozen ops: a leaf test behind two wrappers otherwise waits for two
// dependent reads of global memory inside the traversal loop; cornell 113.2 -> 103.9 ms per
// 32 spp), else from global memory.
RT_DEV DOp load_op(const DeviceScene& sc, uint32_t index) {
  union {
    f32x4 q[2];
    DOp op;
  } u;
  if (index < sc.n_ops_lds) {
    extern __shared__ __align__(16) char rt_dynamic_lds[];
    const lds_f32x4* l = (const lds_f32x4*)(rt_dynamic_lds + sc.ops_lds_offset) + 2u * index;
    u.q[0] = l[0];
    u.q[1] = l[1];
  } else {
    const f32x4* g = (const f32x4*)(sc.ops + index);
    u.q[0] = g[0];
    u.q[1] = g[1];
  }
  return u.op;
}

// Forward (world -> innermost frame) transform of a ray through wrapper ops,
// outermost first: Translate::hit instance.rs:13, RotateY::hit instance.rs:52-61.
template <bool STATS>
RT_DEV void ops_forward(const DeviceScene& sc, uint32_t ops_word, uint32_t upto, V3& o, V3& d,
                        Stats& st) {
  const uint32_t first = ops_first(ops_word);
  for (uint32_t k = 0; k < upto; ++k) {
    const DOp op = load_op(sc, first + k);
    if (op.kind == RT_OP_TRANSLATE) {
      stat<STATS>(st, RT_STAT_OP_TRANSLATE);
      o = vsub(o, v3(op.a));
    } else {
      stat<STATS>(st, RT_STAT_OP_ROTATE);
      double s = op.a[0], c = op.a[1];
      V3 no = o, nd = d;
      no.x = c * o.x - s * o.z;
      no.z = s * o.x + c * o.z;
      nd.x = c * d.x - s * d.z;
      nd.z = s * d.x + c * d.z;
      o = no;
      d = nd;
    }
  }
}

// Direction of the ray as seen by wrapper `level` (after ops [0, level]).
RT_DEV V3 ops_direction(const DeviceScene& sc, uint32_t ops_word, uint32_t level, V3 d) {
  const uint32_t first = ops_first(ops_word);
  for (uint32_t k = 0; k <= level; ++k) {
    const DOp op = load_op(sc, first + k);
    if (op.kind == RT_OP_ROTATE_Y) {
      double s = op.a[0], c = op.a[1];
      V3 nd = d;
      nd.x = c * d.x - s * d.z;
      nd.z = s * d.x + c * d.z;
      d = nd;
    }
  }
  return d;
}

struct Record {  // HitRecord, src/hittable/mod.rs:32-40 (material and t live elsewhere)
  V3 p, normal;
  double u, v;
  bool front;
};

RT_DEV void set_face_normal(Record& r, V3 dir, V3 outward) {  // mod.rs:65-68
  r.front = vdot(dir, outward) < 0.0;
  r.normal = r.front ? outward : vneg(outward);
}

// Backward pass of the wrappers over a record, innermost first
// (Translate instance.rs:14-18, RotateY instance.rs:63-82 with its x-for-z slip kept).
RT_DEV void ops_backward(const DeviceScene& sc, uint32_t ops_word, V3 world_dir, Record& r) {
  const uint32_t first = ops_first(ops_word);
  for (uint32_t k = ops_count(ops_word); k-- > 0;) {
    const DOp op = load_op(sc, first + k);
    V3 dir = ops_direction(sc, ops_word, k, world_dir);
    if (op.kind == RT_OP_TRANSLATE) {
      r.p = vadd(r.p, v3(op.a));
      set_face_normal(r, dir, r.normal);
    } else {
      double s = op.a[0], c = op.a[1];
      V3 p = r.p, n = r.normal;
      p.x = c * r.p.x + s * r.p.x;
      p.z = -s * r.p.x + c * r.p.x;
      n.x = c * r.normal.x + s * r.normal.x;
      n.z = -s * r.normal.x + c * r.normal.x;
      r.p = p;
      set_face_normal(r, dir, n);
    }
  }
}

RT_DEV V3 moving_center(const DeviceScene& sc, const DPrim& pr, double time) {  // moving_sphere.rs:20-23
  const DAux ax = sc.aux[(uint32_t)pr.p5];
  double t0 = pr.p[4], t1 = ax.v[3];
  double portion = (t0 - time) / (t1 - t0);
  V3 c0 = v3(pr.p[0], pr.p[1], pr.p[2]), c1 = v3(ax.v[0], ax.v[1], ax.v[2]);
  return vadd(vscale(c0, 1.0 - portion), vscale(c1, portion));  // lerp, interpolate.rs:3-9
}

RT_DEV bool in_range(double t, double lo, double hi) { return lo <= t && t < hi; }  // Range::contains

// Sphere::hit up to the root (sphere.rs:32-51).
// `a` is ray.direction.length_squared(): the same value for every sphere a ray meets,
// so callers compute it once per ray (or once per wrapper frame).
template <bool STATS>
RT_DEV bool sphere_t(V3 c, double radius, V3 o, V3 d, double a, double tmin, double tmax,
                     double& t, Stats& st) {
  V3 oc = vsub(o, c);
  double half_b = vdot(oc, d);
  double cc = vlen2(oc) - radius * radius;
  double discrim = half_b * half_b - a * cc;
  if (discrim < 0.0) return false;
  stat<STATS>(st, RT_STAT_SPHERE_ROOTS);
  RT_LP(st, LP_ROOTS);
  double sqrtd = sqrt(discrim);
  double root = (-half_b - sqrtd) / a;
  if (!in_range(root, tmin, tmax)) {
    root = (-half_b + sqrtd) / a;
    if (!in_range(root, tmin, tmax)) return false;
  }
  t = root;
  return true;
}

// AxisAlignedRect::hit up to the containment test (rect.rs:39-51).
// `overhang`: for a record tied to the reference's box chain (kChainBit), is the hit point
// outside, or within rounding of the edge of, its cuboid's bounding box?  The Y side of a
// cuboid spans the box's z half-extent along x and its x half-extent along z (cuboid.rs:21-30),
// so the box's extent along one in-plane axis is the side's extent along the other.  A point
// well inside the box is inside every box of the chain, which then cannot reject the ray.
RT_DEV bool rect_t(const DPrim& pr, V3 o, V3 d, double tmin, double tmax, double& t, bool& overhang) {
  uint32_t d0 = (pr.kind >> 4) & 3u;
  uint32_t d1 = d0 == 2 ? 0 : d0 + 1;
  uint32_t d2 = d1 == 2 ? 0 : d1 + 1;
  double tt = (pr.p[0] - vget(o, d0)) / vget(d, d0);
  if (!in_range(tt, tmin, tmax)) return false;
  double i = vget(o, d1) + tt * vget(d, d1);
  double j = vget(o, d2) + tt * vget(d, d2);
  if (!(pr.p[1] <= i && i <= pr.p[2]) || !(pr.p[3] <= j && j <= pr.p[4])) return false;
  t = tt;
  if (pr.kind & kChainBit) {
    const double keep = 0.999999;
    double half_i = 0.5 * (pr.p[4] - pr.p[3]) * keep, half_j = 0.5 * (pr.p[2] - pr.p[1]) * keep;
    double ci = 0.5 * (pr.p[1] + pr.p[2]), cj = 0.5 * (pr.p[3] + pr.p[4]);
    overhang = !(fabs(i - ci) < half_i && fabs(j - cj) < half_j);
  }
  return true;
}

// hit().t of one base primitive inside its wrappers, for a ray given in the frame
// outside those wrappers.
// `dd` = d.d of the ray as given (used as is when the primitive has no wrappers).
template <bool STATS>
RT_DEV bool base_prim_t(const DeviceScene& sc, const DPrim& pr, V3 o, V3 d, double dd, double time,
                        double tmin, double tmax, double& t, Stats& st, bool& overhang) {
  uint32_t n_ops = ops_count(pr.ops);
  if (n_ops) {
    ops_forward<STATS>(sc, pr.ops, n_ops, o, d, st);
    dd = vlen2(d);  // a RotateY in the chain changes the components (not the length, up to rounding)
  }
  uint32_t kind = pr.kind & 15u;
  if (kind == DP_SPHERE) {
    stat<STATS>(st, RT_STAT_SPHERE_TESTS);
    return sphere_t<STATS>(v3(pr.p[0], pr.p[1], pr.p[2]), pr.p[3], o, d, dd, tmin, tmax, t, st);
  }
  if (kind == DP_MOVING_SPHERE) {
    stat<STATS>(st, RT_STAT_MSPHERE_TESTS);
    return sphere_t<STATS>(moving_center(sc, pr, time), pr.p[3], o, d, dd, tmin, tmax, t, st);
  }
  stat<STATS>(st, RT_STAT_RECT_TESTS);
  return rect_t(pr, o, d, tmin, tmax, t, overhang);
}

// boundary.hit(ray, range).t for a medium: first minimum over its primitives
// (HittableList::hit, hittable_list.rs:28-31).
template <bool STATS>
RT_DEV bool boundary_t(const DeviceScene& sc, uint32_t first, uint32_t count, V3 o, V3 d,
                       double time, double tmin, double tmax, double& t, Stats& st) {
  bool any = false;
  for (uint32_t k = 0; k < count; ++k) {
    const DPrim pr = sc.boundary[first + k];
    double tk;
    bool unused = false;
    if (base_prim_t<STATS>(sc, pr, o, d, vlen2(d), time, tmin, tmax, tk, st, unused)) {
      if (!any || tk < t) t = tk;
      any = true;
    }
  }
  return any;
}

// ConstantMedium::hit (constant_medium.rs:27-76) up to t.
template <bool STATS>
RT_DEV bool medium_t(const DeviceScene& sc, const DPrim& pr, V3 o, V3 d, double time, double tmin,
                     double tmax, const DrawKey& key, double& t, Stats& st) {
  stat<STATS>(st, RT_STAT_MEDIUM_TESTS);
  uint32_t n_ops = ops_count(pr.ops);
  if (n_ops) ops_forward<STATS>(sc, pr.ops, n_ops, o, d, st);
  const double inf = __longlong_as_double(0x7FF0000000000000ll);
  uint32_t first = (uint32_t)pr.p[1], count = (uint32_t)pr.p[2];
  double t1, t2;
  const DPrim b0 = sc.boundary[first];
  const uint32_t bkind = b0.kind & 15u;
  if (count == 1u && ops_count(b0.ops) == 0u && (bkind == DP_SPHERE || bkind == DP_MOVING_SPHERE)) {
    // A sphere as the boundary (every medium of the reference's scenes): boundary.hit(-inf..inf) and
    // boundary.hit(t1 + 0.0001..inf) are Sphere::hit twice on the same ray, so the discriminant, its
    // root and both quotients (sphere.rs:32-47) are the same numbers both times; they are formed
    // once and each call's choice between them (sphere.rs:42-50) is made as that call would.
    const int test_stat = bkind == DP_SPHERE ? RT_STAT_SPHERE_TESTS : RT_STAT_MSPHERE_TESTS;
    stat<STATS>(st, test_stat);  // (counted as the two calls would count)
    const V3 c = bkind == DP_SPHERE ? v3(b0.p[0], b0.p[1], b0.p[2]) : moving_center(sc, b0, time);
    const double a = vlen2(d);
    V3 oc = vsub(o, c);
    double half_b = vdot(oc, d);
    double cc = vlen2(oc) - b0.p[3] * b0.p[3];
    double discrim = half_b * half_b - a * cc;
    if (discrim < 0.0) return false;
    stat<STATS>(st, RT_STAT_SPHERE_ROOTS);
    double sqrtd = sqrt(discrim);
    const double near = (-half_b - sqrtd) / a, far = (-half_b + sqrtd) / a;
    if (in_range(near, -inf, inf)) t1 = near;
    else if (in_range(far, -inf, inf)) t1 = far;
    else return false;
    stat<STATS>(st, test_stat);
    stat<STATS>(st, RT_STAT_SPHERE_ROOTS);
    const double lo2 = t1 + 0.0001;
    if (in_range(near, lo2, inf)) t2 = near;
    else if (in_range(far, lo2, inf)) t2 = far;
    else return false;
  } else {
    if (!boundary_t<STATS>(sc, first, count, o, d, time, -inf, inf, t1, st)) return false;
    if (!boundary_t<STATS>(sc, first, count, o, d, time, t1 + 0.0001, inf, t2, st)) return false;
  }
  if (t1 < tmin) t1 = tmin;
  if (t2 > tmax) t2 = tmax;
  if (t1 >= t2) return false;
  if (t1 < 0.0) t1 = 0.0;
  double ray_length = vlen(d);
  double distance_inside = (t2 - t1) * ray_length;
  u32x4 b = draw(key, DRAW_MEDIUM, pr.object);
  double hit_distance = pr.p[0] * log10(u01_53(b.x, b.y));
  if (hit_distance > distance_inside) return false;
  t = t1 + hit_distance / ray_length;
  return true;
}

RT_DEV void sphere_uv(V3 n, double& u, double& v) {  // Sphere::get_uv, sphere.rs:20-24
  const double PI = 3.14159265358979323846264338327950288;
  double theta = acos(-n.y);
  double phi = atan2(-n.z, n.x) + PI;
  u = phi / (2.0 * PI);
  v = theta / PI;
}

// The HitRecord the reference would return for leaf slot `pr` hit at parameter t
// by the world-frame ray (o, d, time).
// SPHERES: the record is known to be a plain sphere without wrappers.
template <bool UV_POSSIBLE = true, bool SPHERES = false>
RT_DEV Record make_record(const DeviceScene& sc, const DPrim& pr, V3 o, V3 d, double time, double t,
                          bool want_uv) {
  if (!UV_POSSIBLE) want_uv = false;  // no texture of the scene reads (u, v): drops acos/atan2
  Record r;
  r.u = 0.0;
  r.v = 0.0;
  if (SPHERES) {  // sphere.rs:52-63
    r.p = vadd(o, vscale(d, t));
    V3 outward = vscale(vsub(r.p, v3(pr.p[0], pr.p[1], pr.p[2])), pr.p[4]);
    if (want_uv) sphere_uv(outward, r.u, r.v);
    set_face_normal(r, d, outward);
    return r;
  }
  Stats unused;
  uint32_t kind = pr.kind & 15u;
  uint32_t n_ops = ops_count(pr.ops);
  V3 lo = o, ld = d;
  if (n_ops) ops_forward<false>(sc, pr.ops, n_ops, lo, ld, unused);
  if (kind == DP_MEDIUM) {  // constant_medium.rs:63-75
    r.p = vadd(lo, vscale(ld, t));
    set_face_normal(r, ld, v3(1.0, 0.0, 0.0));
  } else if (kind == DP_RECT) {  // rect.rs:53-64
    uint32_t d0 = (pr.kind >> 4) & 3u;
    r.p = vadd(lo, vscale(ld, t));
    if (want_uv) {
      uint32_t d1 = d0 == 2 ? 0 : d0 + 1;
      uint32_t d2 = d1 == 2 ? 0 : d1 + 1;
      double i = vget(lo, d1) + t * vget(ld, d1);
      double j = vget(lo, d2) + t * vget(ld, d2);
      r.u = (i - pr.p[1]) / (pr.p[2] + -pr.p[1]);
      r.v = (j - pr.p[3]) / (pr.p[4] + -pr.p[3]);
    }
    V3 outward = v3(d0 == 0 ? 1.0 : 0.0, d0 == 1 ? 1.0 : 0.0, d0 == 2 ? 1.0 : 0.0);
    set_face_normal(r, ld, outward);
  } else {  // sphere.rs:52-63
    V3 c = kind == DP_SPHERE ? v3(pr.p[0], pr.p[1], pr.p[2]) : moving_center(sc, pr, time);
    r.p = vadd(lo, vscale(ld, t));
    // (p - center) / radius = (p - center) * radius.recip(); a sphere's record carries the reciprocal
    V3 outward = kind == DP_SPHERE ? vscale(vsub(r.p, c), pr.p[4]) : vdiv(vsub(r.p, c), pr.p[3]);
    if (want_uv) sphere_uv(outward, r.u, r.v);
    set_face_normal(r, ld, outward);
  }
  if (n_ops) ops_backward(sc, pr.ops, d, r);
  return r;
}

// -------------------------------------------------------------- traversal
// Per-ray constants of the f32 slab test, one set per axis.
//
// The test only culls, so it runs in f32 with FMA; what makes it safe is that it
// is CONSERVATIVE against the exact slab parameters T = (b - o) / d:
//   inv = fl(1 / fl32(d)) carries a relative error < 2^-22, each FMA another
//   2^-24: both scale T and are absorbed by widening the exit side by 1 + 2^-18;
//   rounding o to f32 and rounding -o*inv shift T by at most 2^-23 |o inv|: the
//   pad E = 2^-21 |o inv| is subtracted on the entry side and added on the exit
//   side (which plane is which follows the sign of inv).
// The widening is folded into the constants: the plane that is the exit for this
// direction sign gets (inv, -o*inv) both multiplied by 1 + 2^-18 (two more
// roundings of 2^-24 each, inside the same budgets), so a box test is
//   t_lo = fma(lo, inv_lo, add_lo), t_hi = fma(hi, inv_hi, add_hi),
//   max(min(t_lo, t_hi) per axis, tmin) <= min(max(t_lo, t_hi) per axis, best * (1 + 2^-18)).
// A zero or denormal d would give inv = inf and inf - inf = NaN; inv is clamped
// to +-1e25, which still puts both planes beyond every real t.
// Every operation is a correctly rounded IEEE f32 operation, so a CPU can
// reproduce the decisions bit for bit.
const float kWiden = 1.000003814697265625f;  // 1 + 2^-18
struct CullAxis {
  float inv_lo, add_lo, inv_hi, add_hi;
};
RT_DEV CullAxis cull_axis(double o, double d) {
  float d32 = (float)d, o32 = (float)o;
  float inv = __fdiv_rn(1.0f, d32);
  if (!(fabsf(inv) <= 1e25f)) inv = copysignf(1e25f, d32);
  float noi = -(o32 * inv);
  float e = fabsf(noi) * 4.76837158203125e-07f;  // 2^-21
  // which plane is the exit follows the sign of inv; the entry side's factor is exactly 1
  const bool pos = inv >= 0.0f;
  const float k_lo = pos ? 1.0f : kWiden, k_hi = pos ? kWiden : 1.0f;
  const float a_entry = noi - e, a_exit = noi + e;
  CullAxis c;
  c.inv_lo = inv * k_lo;
  c.add_lo = (pos ? a_entry : a_exit) * k_lo;
  c.inv_hi = inv * k_hi;
  c.add_hi = (pos ? a_exit : a_entry) * k_hi;
  return c;
}

struct Closest {
  double t;
  uint32_t item;  // leaf slot, or kNoPrim
  uint32_t kind;  // kind word of that slot's record
};

// The head of the node array (breadth-first order, so the top of the tree) and
// the head of the leaf-slot array, staged in LDS by each workgroup, plus the
// traversal stack.  All three share one dynamic LDS allocation laid out by
// LdsLayout on the host.
struct LdsLayout {
  uint32_t n_nodes;      // nodes [0, n_nodes) are in LDS
  uint32_t n_prims;      // leaf slots [0, n_prims) are in LDS
  uint32_t stack_depth;  // stack entries per lane held in LDS
  uint32_t cand_rows;    // rows after the stack: leaf slots whose test waits for the end of the traversal
  uint32_t n_ops;        // wrapper ops [0, n_ops) are in LDS, between the leaf slots and the stack
  uint32_t bytes;        // total dynamic LDS bytes per workgroup
  uint32_t entry_bits;   // bits of a traversal entry code (see entry_code_bits)
  // Hierarchies whose worst-case stack need exceeds stack_depth: entries from index
  // stack_depth on live in global memory, overflow[(index - stack_depth) * overflow_stride +
  // global thread id].  Null when the whole stack fits in LDS.
  uint32_t* overflow;
  uint32_t overflow_stride;  // threads the overflow area was sized for (a launch has no more)
};
struct LdsScene {
  const lds_f32x4* nodes;
  const lds_f32x4* prims;  // the first 3 x 16 bytes of each DPrim
  lds_u32* stack;          // this lane's column of packed entries, `stack_stride` words apart
  uint32_t n_nodes, n_prims, stack_stride, entry_bits;
  uint32_t stack_cap;      // entries of the column that exist in LDS
  uint32_t cand_cap;       // rows [stack_cap, stack_cap + cand_cap) of the column: postponed leaf slots
  uint32_t ops_offset;     // bytes from the start of the dynamic LDS to the staged wrapper ops
  uint32_t* overflow;      // this lane's column of the global overflow area, or null (uniform)
  uint32_t overflow_stride;
};

// Carve the workgroup's dynamic LDS and stage the scene head into it.  Every
// thread of the workgroup must call this (it ends with a barrier).
// PRIM_QUADS: 16-byte pieces of a leaf-slot record that are staged (3: everything the traversal
// reads of any record; 2: centre and radius, all it reads when every record is a plain sphere).
template <uint32_t PRIM_QUADS = 3>
RT_DEV LdsScene lds_stage(const DeviceScene& sc, const LdsLayout& lay, char* smem, uint32_t tid,
                          uint32_t block) {
  lds_f32x4* nodes = (lds_f32x4*)smem;
  lds_f32x4* prims = nodes + 2u * lay.n_nodes;
  lds_f32x4* ops = prims + PRIM_QUADS * lay.n_prims;
  lds_u32* stack = (lds_u32*)(ops + 2u * lay.n_ops);
  const f32x4* gn = (const f32x4*)sc.nodes;
  const f32x4* gp = (const f32x4*)sc.prims;
  const f32x4* go = (const f32x4*)sc.ops;
  for (uint32_t i = tid; i < 2u * lay.n_nodes; i += block) nodes[i] = gn[i];
  for (uint32_t i = tid; i < PRIM_QUADS * lay.n_prims; i += block)
    prims[i] = gp[4u * (i / PRIM_QUADS) + i % PRIM_QUADS];
  for (uint32_t i = tid; i < 2u * lay.n_ops; i += block) ops[i] = go[i];
  __syncthreads();
  LdsScene l;
  l.nodes = nodes;
  l.prims = prims;
  l.stack = stack + tid;
  l.n_nodes = lay.n_nodes;
  l.n_prims = lay.n_prims;
  l.stack_stride = block;
  l.entry_bits = lay.entry_bits;
  l.stack_cap = lay.stack_depth;
  l.cand_cap = lay.cand_rows;
  l.overflow = lay.overflow ? lay.overflow + (blockIdx.x * block + tid) : nullptr;
  l.overflow_stride = lay.overflow_stride;
  l.ops_offset = 32u * lay.n_nodes + 16u * PRIM_QUADS * lay.n_prims;
  return l;
}

// LDS writes of one wave are visible to its own later reads once they have been
// issued in order; this keeps the compiler from moving accesses across.
RT_DEV void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The accepted candidate of rand_unit_vector's rejection loop (vec3.rs:35-47, before
// the normalisation) for every lane with `need`, found by the whole wave together.
//
// A lane on its own repeats "draw candidate `it`, stop at the first one inside the
// unit sphere"; the wave then runs as many trips as its unluckiest lane while most
// lanes sit idle.  Here the n lanes still looking share all 64 lanes: each owner
// gets c = 2^k <= 64/n helpers, helper j of an owner tries that owner's candidate
// number tried + j (the owner's key travels through two rows of 64 words of LDS
// private to the wave, `xch` and `xch + row`: callers lend rows of the idle
// traversal stack), and the owner takes the accepted candidate with the lowest number.
// The candidates, their order and therefore the result are exactly the lone
// lane's; only who evaluates them changes.  Must be called by all 64 lanes.
//
// `need_u` (never together with `need`): the lane wants the uniform of Dielectric::scatter's
// reflect-or-refract choice instead (dielectric.rs:27: block DRAW_FRESNEL, seq 0).  It rides
// along in the first trip as one more owner whose candidate number 0 is always taken, so a wave
// that holds a few glass hits does not run a whole generator block for them alone; `u_out`
// receives the same value a lone draw gives.
RT_DEV V3 wave_rand_in_unit_sphere(bool need, bool need_u, const DrawKey& key, lds_u32* xch, uint32_t row,
                                   Stats& st, double& u_out) {
  const uint32_t lane = __lane_id();
  unsigned long long vpend = __ballot(need), upend = __ballot(need_u);
  uint32_t tried = 0;
  V3 v = v3(0.0, 0.0, 0.0);
  while ((vpend | upend) != 0ull) {
    RT_LP(st, LP_REJECT);
    const uint32_t nu = (uint32_t)__popcll(upend), n = nu + (uint32_t)__popcll(vpend);
    const uint32_t c_log = n == 1u ? 6u : (uint32_t)__clz((int)(n - 1u)) - 26u;
    const bool mine_v = (vpend >> lane) & 1ull, mine_u = (upend >> lane) & 1ull;
    if (c_log == 0u) {
      // more than half of the wave is looking: no helper to share, each lane tries its own next
      // candidate and nothing is exchanged
      double x, y, z;
      const u32x4 b = draw_block(key.seed, key.pixel, key.sample, key.bounce,
                                 mine_u ? (uint32_t)DRAW_FRESNEL : (uint32_t)DRAW_SCATTER, mine_u ? 0u : tried);
      sym3_42(b, x, y, z);
      if (mine_u) u_out = u01_53(b.x, b.y);
      const bool got = mine_v && tried < kMaxRejections &&
                       (x * x + y * y + z * z < 1.0 || tried == kMaxRejections - 1u);
      if (got) v = v3(x, y, z);
      vpend &= ~__ballot(got);
      upend = 0ull;
      tried += 1u;
      continue;
    }
    // owners in rank order: the uniform requests first (nu is the same for every lane)
    const unsigned long long below = (1ull << lane) - 1ull;
    const uint32_t rank = mine_u ? (uint32_t)__popcll(upend & below) : nu + (uint32_t)__popcll(vpend & below);
    if (mine_v || mine_u) {
      xch[rank] = key.pixel;
      xch[row + rank] = key.sample;
    }
    wave_lds_sync();
    const uint32_t q = lane >> c_log, sub = lane & ((1u << c_log) - 1u);
    const bool active = q < n;
    const uint32_t qq = active ? q : 0u;
    const bool for_u = qq < nu;
    const uint32_t seq = for_u ? sub : tried + sub;
    double x, y, z;
    const u32x4 b = draw_block(key.seed, xch[qq], xch[row + qq], key.bounce,
                               for_u ? (uint32_t)DRAW_FRESNEL : (uint32_t)DRAW_SCATTER, seq);
    sym3_42(b, x, y, z);
    // the lone loop gives up after kMaxRejections candidates and keeps the last one
    const bool ok = active && (for_u ? sub == 0u
                                     : (seq < kMaxRejections &&
                                        (x * x + y * y + z * z < 1.0 || seq == kMaxRejections - 1u)));
    const unsigned long long accepted = __ballot(ok);
    const uint32_t shift = rank << c_log;
    const unsigned long long group = c_log == 6u ? ~0ull : ((1ull << (1u << c_log)) - 1ull);
    const unsigned long long m = (mine_v || mine_u) ? ((accepted >> shift) & group) : 0ull;
    const bool got = m != 0ull;
    const int src = got ? (int)(shift + (uint32_t)__ffsll((long long)m) - 1u) : (int)lane;
    // a uniform request gets the block's first two words back, untouched, in the place of x
    const double px = for_u ? __longlong_as_double((long long)(((uint64_t)b.y << 32) | b.x)) : x;
    const double fx = __shfl(px, src), fy = __shfl(y, src), fz = __shfl(z, src);
    if (got && mine_v) v = v3(fx, fy, fz);
    if (got && mine_u) {
      const uint64_t w = (uint64_t)__double_as_longlong(fx);
      u_out = u01_53((uint32_t)w, (uint32_t)(w >> 32));
    }
    vpend &= ~__ballot(got && mine_v);
    upend = 0ull;
    tried += 1u << c_log;
    wave_lds_sync();
  }
  return v;
}
RT_DEV V3 wave_rand_in_unit_sphere(bool need, const DrawKey& key, lds_u32* xch, uint32_t row,
                                   Stats& st) {
  double unused = 0.0;
  return wave_rand_in_unit_sphere(need, false, key, xch, row, st, unused);
}

// Leaf slot `slot` as the traversal needs it: the first 48 bytes from LDS when staged, and
// the rest only for the kinds that read it (a moving sphere's second centre, a medium's
// object id, a chain id).
// FULL: always the whole record (shading reads material and object).
template <bool FULL>
RT_DEV DPrim load_prim(const DeviceScene& sc, const LdsScene& lds, uint32_t slot) {
  union {
    f32x4 q[4];
    DPrim p;
  } u;
  const f32x4* g = (const f32x4*)(sc.prims + slot);
  if (slot < lds.n_prims) {
#pragma unroll
    for (int k = 0; k < 3; ++k) u.q[k] = lds.prims[3u * slot + k];
  } else {
#pragma unroll
    for (int k = 0; k < 3; ++k) u.q[k] = g[k];
  }
  u.q[3] = f32x4{0.f, 0.f, 0.f, 0.f};
  const uint32_t kind = u.p.kind & 15u;  // (a medium draws with its object id)
  if (FULL || kind == DP_MOVING_SPHERE || kind == DP_MEDIUM || (u.p.kind & kChainBit)) u.q[3] = g[3];
  return u.p;
}

// Would the reference's own hierarchy let this ray reach the leaf list of chain `chain`?
// BvhNode::hit (bvh.rs:82-87) at every node from the root down, each with
// Aabb::intersect (aabb.rs:35-50) over the full range tmin..inf.  Asked only for hits
// on geometry outside its object's bounding_box() (records with kChainBit).
RT_DEV bool reference_chain_passes(const DeviceScene& sc, uint32_t chain, V3 o, V3 d, double tmin) {
  const double inf = __longlong_as_double(0x7FF0000000000000ll);
  const uint32_t first = sc.chain_ranges[2u * chain], count = sc.chain_ranges[2u * chain + 1u];
  const double inv[3] = {1.0 / d.x, 1.0 / d.y, 1.0 / d.z};  // f64::recip
  const double org[3] = {o.x, o.y, o.z};
  for (uint32_t k = 0; k < count; ++k) {
    const double* b = sc.chain_boxes + 6u * (size_t)(first + k);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      double t0 = (b[a] + -org[a]) * inv[a];
      double t1 = (b[3 + a] + -org[a]) * inv[a];
      if (inv[a] < 0.0) {
        double t = t0;
        t0 = t1;
        t1 = t;
      }
      // f64::max / f64::min: a NaN operand is ignored
      double t_min = fmax(t0, tmin), t_max = fmin(t1, inf);
      if (t_max <= t_min) return false;
    }
  }
  return true;
}

// world.hit(ray, tmin..inf): closest t over the flattened BVH.
//
// A traversal entry is a code for what a record's (a, b) say:
//   interior: ((first child << 2) | (children - 1)) << 1;  leaf: (((first item << 3) | (count - 1)) << 1) | 1.
// The device copy of a record holds that code in place of `a` (see device_nodes).
// A visit of an interior entry reads its four consecutive child records at once
// (fewer than four children: the extra records are read and ignored), slab-tests
// each against [tmin, best t] and forms one 32-bit word per child: the code in the
// low `entry_bits` bits and, above them, as many leading bits of the (positive)
// f32 entry parameter as fit (truncation rounds the parameter down, which only
// makes later drop tests keep more); a missed child gives 0xFFFFFFFF.  The four
// words are sorted as unsigned integers, which orders them by entry parameter:
// the traversal continues with the smallest and stacks the other hits, farthest
// first.  A popped word whose entry parameter now lies beyond the best t is
// dropped.  The whole visit is free of branches.
// Primitive tests always use the reference's full range tmin..inf so that equal
// t can be resolved by rank.
const uint32_t kNoEntry = 0xFFFFFFFFu;

RT_HD uint32_t make_entry(uint32_t a, uint32_t b) {
  return (b & 0x80000000u) ? (((a << kWideBits) | (b & (kWide - 1u))) << 1) : ((((a << 3) | (b - 1u)) << 1) | 1u);
}
// Bits needed for every entry code of a hierarchy with this many records and items.
RT_HD uint32_t entry_code_bits(uint32_t n_nodes, uint32_t n_items) {
  uint32_t hi = (n_nodes << (kWideBits + 1u)) > (n_items << 4) ? (n_nodes << (kWideBits + 1u)) : (n_items << 4);
  uint32_t bits = 1;
  while (bits < 31 && (hi >> bits) != 0) ++bits;
  return bits;
}
// The entry parameter as a stack word keeps it: low (bits - 1) float bits cleared.
RT_HD uint32_t stack_t_bits(float t, uint32_t bits) {
#if defined(__HIP_DEVICE_COMPILE__)
  uint32_t u = __float_as_uint(t);
#else
  uint32_t u;
  __builtin_memcpy(&u, &t, 4);
#endif
  return (u >> (bits - 1u)) << (bits - 1u);
}

// MEDIA: constant media are tested (else skipped); CHAINS: records tied to the reference's
// box chain are checked against it; OVERFLOW: the hierarchy can need more stack entries than
// LDS holds per lane, the rest live in the global overflow area.
// SPHERES: every leaf slot is a plain sphere without wrappers (the hierarchy's workgroup copy
// then holds 32 bytes per slot, centre and radius); `kind` of the result is left 0, the caller
// reads the record of the slot it shades.
template <bool STATS, bool MEDIA, bool CHAINS, bool OVERFLOW, bool SPHERES = false>
RT_DEV Closest closest_hit(const DeviceScene& sc, const LdsScene& lds, V3 o, V3 d, double time,
                           double tmin, const DrawKey& key, Stats& st) {
  lds_u32* stack = lds.stack;
  const uint32_t stride = lds.stack_stride;
  const uint32_t ebits = lds.entry_bits, emask = (1u << ebits) - 1u;
  const double inf = __longlong_as_double(0x7FF0000000000000ll);
  Closest best;
  best.t = inf;
  best.item = kNoPrim;
  best.kind = 0;
  if (sc.n_nodes == 0) return best;
  float best32w = __int_as_float(0x7F800000);  // best t so far, times kWiden
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const double dd = vlen2(d);
  const f32x4* gnodes = (const f32x4*)sc.nodes;
  uint32_t sp = 0;
  uint32_t cur = sc.root_code;  // the root's own box is not tested
  // stack slot access
  uint32_t* const overflow = lds.overflow;
  const uint32_t cap = lds.stack_cap, ostride = lds.overflow_stride;
  auto put = [&](uint32_t pos, uint32_t word) {
    if (!OVERFLOW || pos < cap)
      stack[pos * stride] = word;
    else
      overflow[(size_t)(pos - cap) * ostride] = word;
  };
  auto get = [&](uint32_t pos) -> uint32_t {
    if (!OVERFLOW || pos < cap) return stack[pos * stride];
    return overflow[(size_t)(pos - cap) * ostride];
  };

  // next entry from the stack that can still matter, or kNoEntry
  auto pop = [&]() -> uint32_t {
    while (sp) {
      RT_LP(st, LP_POP);
      uint32_t w = get(--sp);
      if (__uint_as_float((w >> ebits) << (ebits - 1u)) <= best32w) return w & emask;
    }
    return kNoEntry;
  };

  auto take = [&](double t, uint32_t slot, uint32_t kind) {  // HittableList::hit's first minimum, by rank
    if (t < best.t || (t == best.t && best.item != kNoPrim && sc.ranks[slot] < sc.ranks[best.item])) {
      best.t = t;
      best.item = slot;
      best.kind = kind;
      best32w = (float)t * kWiden;
    }
  };
  uint32_t n_cand = 0;  // leaf slots set aside in rows [cap, cap + cand_cap) of the lane's column
  do {
  // culling arithmetic: once per ray, then one FMA per slab plane.  (Kernels that settle
  // postponed slots between two stretches of the walk form the constants again for each
  // stretch, from coordinates the optimiser cannot see through: held across the medium code
  // they would cost a quarter of the register file for something that is almost never resumed.)
  V3 oc = o, dc = d;
  if (MEDIA || CHAINS) {
    asm volatile("" : "+v"(oc.x), "+v"(oc.y), "+v"(oc.z), "+v"(dc.x), "+v"(dc.y), "+v"(dc.z));
  }
  const CullAxis cx = cull_axis(oc.x, dc.x), cy = cull_axis(oc.y, dc.y), cz = cull_axis(oc.z, dc.z);
  const float tmin32 = (float)tmin * 0.99f;
  // constants paired the way the record's words pair up: (lo.x lo.y) (lo.z hi.x) (hi.y hi.z)
  const f32x2 mul_a = {cx.inv_lo, cy.inv_lo}, add_a = {cx.add_lo, cy.add_lo};
  const f32x2 mul_b = {cz.inv_lo, cx.inv_hi}, add_b = {cz.add_lo, cx.add_hi};
  const f32x2 mul_c = {cy.inv_hi, cz.inv_hi}, add_c = {cy.add_hi, cz.add_hi};
  for (;;) {
    RT_LP(st, LP_OUTER);
    // descend until the current entry is a leaf
    uint32_t leaf = kNoEntry;
    while (cur != kNoEntry) {
      if (cur & 1u) {
        leaf = cur;
        break;
      }
      RT_LP(st, LP_NODE);
      const uint32_t first = cur >> (kWideBits + 1u), n = ((cur >> 1) & (kWide - 1u)) + 1u;
      const bool in_lds = first + kWide <= lds.n_nodes;
      if (STATS) {
        st.v[RT_STAT_BOX_TESTS] += n;
        st.v[RT_STAT_NODE_VISITS]++;
      }
      uint32_t w[kWide];
#pragma unroll
      for (uint32_t half = 0; half < kWide / 4u; ++half) {  // four children at a time (their records: 32 registers)
        f32x4 r[8];
        if (in_lds) {
#pragma unroll
          for (uint32_t k = 0; k < 8; ++k) r[k] = lds.nodes[2 * first + 8 * half + k];
        } else {
#pragma unroll
          for (uint32_t k = 0; k < 8; ++k) r[k] = gnodes[2 * first + 8 * half + k];
        }
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) {
          const f32x4 n0 = r[2 * k], n1 = r[2 * k + 1];
          // three packed FMAs: (ax ay) (az bx) (by bz)
          const f32x2 pa = __builtin_elementwise_fma(f32x2{n0.x, n0.y}, mul_a, add_a);
          const f32x2 pb = __builtin_elementwise_fma(f32x2{n0.z, n0.w}, mul_b, add_b);
          const f32x2 pc = __builtin_elementwise_fma(f32x2{n1.x, n1.y}, mul_c, add_c);
          const float ax = pa.x, ay = pa.y, az = pb.x, bx = pb.y, by = pc.x, bz = pc.y;
          float t0 = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), tmin32));
          float t1 = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), best32w));
          const uint32_t word = ((__float_as_uint(t0) >> (ebits - 1u)) << ebits) | __float_as_uint(n1.z);
          w[4 * half + k] = (4 * half + k < n && t0 <= t1) ? word : kNoEntry;
        }
      }
      // sort ascending
      auto cx = [&](uint32_t i, uint32_t j) {
        const uint32_t lo = min(w[i], w[j]), hi = max(w[i], w[j]);
        w[i] = lo;
        w[j] = hi;
      };
      if (kWide == 4) {  // five compare-exchanges
        cx(0, 1); cx(2, 3); cx(0, 2); cx(1, 3); cx(1, 2);
      } else {  // nineteen (the optimal network for eight inputs)
        cx(0, 1); cx(2, 3); cx(4, 5); cx(6, 7); cx(0, 2); cx(1, 3); cx(4, 6); cx(5, 7); cx(1, 2); cx(5, 6);
        cx(0, 4); cx(3, 7); cx(1, 5); cx(2, 6); cx(1, 4); cx(3, 6); cx(2, 4); cx(3, 5); cx(3, 4);
      }
      // stack the far hits, farthest first; a miss is overwritten by the next store
#pragma unroll
      for (uint32_t k = kWide - 1u; k >= 1u; --k) {
        put(sp, w[k]);
        sp += w[k] != kNoEntry ? 1u : 0u;
      }
      cur = w[0] != kNoEntry ? (w[0] & emask) : pop();
    }
    if (leaf == kNoEntry) break;
    const uint32_t leaf_first = leaf >> 4, leaf_count = ((leaf >> 1) & 7u) + 1u;
    bool stopped = false;
    for (uint32_t k = 0; k < leaf_count; ++k) {
      RT_LP(st, LP_LEAFPRIM);
      uint32_t slot = leaf_first + k;
      if (SPHERES) {
        union {
          f32x4 q[2];
          double p[4];
        } u;
        if (slot < lds.n_prims) {
          u.q[0] = lds.prims[2u * slot];
          u.q[1] = lds.prims[2u * slot + 1u];
        } else {
          const f32x4* g = (const f32x4*)(sc.prims + slot);
          u.q[0] = g[0];
          u.q[1] = g[1];
        }
        stat<STATS>(st, RT_STAT_SPHERE_TESTS);
        double t;
        if (sphere_t<STATS>(v3(u.p[0], u.p[1], u.p[2]), u.p[3], o, d, dd, tmin, inf, t, st)) take(t, slot, 0u);
        continue;
      }
      const DPrim pr = load_prim<false>(sc, lds, slot);
      double t;
      bool hit;
      bool overhang = true;
      const bool medium = (pr.kind & 15u) == DP_MEDIUM;
      if (medium)
        hit = MEDIA;  // (a scene's media are skipped by the kernels that do not carry them)
      else
        hit = base_prim_t<STATS>(sc, pr, o, d, dd, time, tmin, inf, t, st, overhang);
      // A constant medium, or a hit outside its object's box (answerable only by the reference's
      // own box chain), is set aside: its slot goes to the lane's list and is settled once the
      // hierarchy has been walked (the loop over the list below).
      if ((MEDIA && medium) || (CHAINS && hit && (pr.kind & kChainBit) && overhang)) {
        if (n_cand == lds.cand_cap) {  // list full: settle it first, then come back to the rest of this leaf
          cur = (((slot << 3) | (leaf_count - k - 1u)) << 1) | 1u;
          stopped = true;
          break;
        }
        stack[(cap + n_cand) * stride] = slot;
        ++n_cand;
        hit = false;
      }
      if (hit) take(t, slot, pr.kind);
    }
    if (stopped) break;
    cur = pop();  // after the leaf: its hits may already rule the next entries out
  }
    if (!(MEDIA || CHAINS)) break;
    // the postponed slots: ConstantMedium::hit, or the flagged record's test again followed by the
    // reference's box chain.  The closest hit is a minimum over (t, rank): the order of the tests
    // does not matter.
    for (uint32_t c = 0; c < n_cand; ++c) {
      const uint32_t slot = stack[(cap + c) * stride];
      const DPrim pr = load_prim<false>(sc, lds, slot);
      double t;
      bool hit;
      if (MEDIA && (pr.kind & 15u) == DP_MEDIUM) {
        hit = medium_t<STATS>(sc, pr, o, d, time, tmin, inf, key, t, st);
        if (CHAINS && hit && (pr.kind & kChainBit)) hit = reference_chain_passes(sc, (uint32_t)pr.p[3], o, d, tmin);
      } else {
        bool overhang = true;
        Stats unused;  // (the test was counted when the traversal made it)
        hit = base_prim_t<false>(sc, pr, o, d, dd, time, tmin, inf, t, unused, overhang);
        hit = hit && reference_chain_passes(sc, (uint32_t)pr.p5, o, d, tmin);
      }
      if (hit) take(t, slot, pr.kind);
    }
    n_cand = 0u;
  } while (cur != kNoEntry);
  return best;
}

// ---------------------------------------------------------------- textures
RT_DEV double perlin_noise(const rt_perlin* t, V3 p) {  // perlin.rs:38-63 and :79-100
  double fx = floor(p.x), fy = floor(p.y), fz = floor(p.z);
  double u = p.x - fx, v = p.y - fy, w = p.z - fz;
  u = u * u * (3.0 - 2.0 * u);
  v = v * v * (3.0 - 2.0 * v);
  w = w * w * (3.0 - 2.0 * w);
  long long i = (long long)fx, j = (long long)fy, k = (long long)fz;
  double uu = u * u * (3.0 - 2.0 * u);
  double vv = v * v * (3.0 - 2.0 * v);
  double ww = w * w * (3.0 - 2.0 * w);
  double accum = 0.0;
#pragma unroll
  for (int di = 0; di < 2; ++di)
#pragma unroll
    for (int dj = 0; dj < 2; ++dj)
#pragma unroll
      for (int dk = 0; dk < 2; ++dk) {
        uint32_t idx = t->perm_x[(i + di) & 0xFF] ^ t->perm_y[(j + dj) & 0xFF] ^
                       t->perm_z[(k + dk) & 0xFF];
        V3 c = v3(t->rand_vec[idx][0], t->rand_vec[idx][1], t->rand_vec[idx][2]);
        double fi = (double)di, fj = (double)dj, fk = (double)dk;
        V3 weight_v = v3(u - fi, v - fj, w - fk);
        accum += vdot(c, weight_v) * (fi * uu + (1.0 - fi) * (1.0 - uu)) *
                 (fj * vv + (1.0 - fj) * (1.0 - vv)) * (fk * ww + (1.0 - fk) * (1.0 - ww));
      }
  return accum;
}

RT_DEV double perlin_turb(const rt_perlin* t, V3 p, int depth) {  // perlin.rs:65-76
  double accum = 0.0, weight = 1.0;
  for (int i = 0; i < depth; ++i) {
    accum += weight * perlin_noise(t, p);
    weight *= 0.5;
    p = vscale(p, 2.0);
  }
  return fabs(accum);
}

// Rust `as u32` from f64: saturating, NaN -> 0.
RT_DEV uint32_t as_u32_sat(double x) {
  if (!(x == x)) return 0u;
  if (x <= 0.0) return 0u;
  if (x >= 4294967295.0) return 4294967295u;
  return (uint32_t)x;
}

template <bool STATS>
RT_DEV V3 texture_value(const DeviceScene& sc, uint32_t tex, double u, double v, V3 p, Stats& st) {
  DTexture t = sc.textures[tex];
  // Checkerboard::value (checkerboard.rs:17-28); children have lower indices, so this ends
  while (t.kind == RT_TEX_CHECKER) {
    stat<STATS>(st, RT_STAT_TEX_CHECKER);
    double sines = 1.0;
    sines = sines * sin(p.x * 10.0);
    sines = sines * sin(p.y * 10.0);
    sines = sines * sin(p.z * 10.0);
    t = sc.textures[sines < 0.0 ? t.b : t.a];
  }
  if (t.kind == RT_TEX_SOLID) return v3(t.color);
  if (t.kind == RT_TEX_PERLIN) {  // perlin.rs:109-116
    stat<STATS>(st, RT_STAT_TEX_PERLIN);
    // Color::white() * 0.5 * (1 + sin(..)): per channel (1.0 * 0.5) * (..)
    double half = 1.0 * 0.5;
    double val = half * (1.0 + sin(t.scale * p.z + 10.0 * perlin_turb(sc.perlin + t.a, p, 7)));
    return v3(val, val, val);
  }
  // Image::value (image.rs:24-35)
  stat<STATS>(st, RT_STAT_TEX_IMAGE);
  const DImage im = sc.images[t.a];
  uint32_t w = im.width, h = im.height;
  uint32_t i = min(as_u32_sat(u * (double)w), w - 1u);
  uint32_t j = min(h - as_u32_sat(v * (double)h), h - 1u);  // u32 wrap, then clamp, as in release Rust
  const uint8_t* px = sc.texels + im.offset + ((size_t)j * w + i) * 3u;
  double scale = 1.0 / 255.0;  // vec3.rs:289-297
  return v3((double)px[0] * scale, (double)px[1] * scale, (double)px[2] * scale);
}

// --------------------------------------------------------------- materials
struct Scatter {
  V3 attenuation;
  V3 direction;
  bool scattered;
};

RT_DEV double reflectance(double cosine, double ref_idx) {  // dielectric.rs:42-46
  double q = (1.0 - ref_idx) / (1.0 + ref_idx);
  double r0 = q * q;
  return r0 + (1.0 - r0) * powi5(1.0 - cosine);
}

}  // namespace rt
