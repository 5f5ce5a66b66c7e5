// Host side of rt_upload_scene: flattens an rt_scene into device records and
// builds the acceleration structure.
//
// Two hierarchies are involved.  (1) The reference's own BVH
// (BvhNode::new_along_axis, src/hittable/bvh.rs:33-77: stable median split on
// box centroids, leaves below 256 objects) is rebuilt only for its ORDER: when
// several objects are hit at exactly the same t the reference returns the one
// its recursion prefers (right subtree over left, bvh.rs:93-97; first in a
// leaf's list, hittable_list.rs:31), and that preference is handed to the
// device as a per-item rank.  (2) The hierarchy the device walks is a binned-SAH
// binary BVH over individual primitives with small leaves and 32-byte nodes
// whose float boxes are rounded outwards; it only has to be conservative,
// because the reference's result is "closest t over everything whose boxes
// were passed" and its box test (src/geom/aabb.rs:35-50) never rejects a true
// hit.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <limits>
#include <numeric>
#include <thread>

#include "device_scene.h"
#include "scene_validate.h"

namespace rt {
namespace {

const double INF = std::numeric_limits<double>::infinity();

struct Box {
  double lo[3] = {INF, INF, INF};
  double hi[3] = {-INF, -INF, -INF};
  void grow(const double* p) {
    for (int a = 0; a < 3; ++a) {
      lo[a] = std::fmin(lo[a], p[a]);
      hi[a] = std::fmax(hi[a], p[a]);
    }
  }
  void grow(const Box& b) {  // (an empty b, lo = +inf and hi = -inf, changes nothing)
    for (int a = 0; a < 3; ++a) {
      lo[a] = std::fmin(lo[a], b.lo[a]);
      hi[a] = std::fmax(hi[a], b.hi[a]);
    }
  }
  double area() const {
    double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    if (!(dx >= 0.0) || !(dy >= 0.0) || !(dz >= 0.0)) return 0.0;
    return 2.0 * (dx * dy + dy * dz + dz * dx);
  }
};

// ---- reference bounding boxes (for the rank order only) --------------------
// Operation order follows the cited functions so that centroid keys, and with
// them the stable sort, agree with the reference.

Box aabb_new(const double* a, const double* b) {  // Aabb::new, src/geom/aabb.rs:16-21
  Box r;
  for (int i = 0; i < 3; ++i) {
    r.lo[i] = std::fmin(a[i], b[i]);
    r.hi[i] = std::fmax(a[i], b[i]);
  }
  return r;
}

void moving_center(const rt_prim& p, double time, double* c) {  // moving_sphere.rs:20-23
  double portion = (p.time0 - time) / (p.time1 - p.time0);
  for (int i = 0; i < 3; ++i) c[i] = (1.0 - portion) * p.center[i] + portion * p.center1[i];
}

Box ref_prim_box(const rt_prim& p, double t0, double t1) {
  if (p.kind == RT_PRIM_SPHERE) {  // sphere.rs:67-72
    double a[3], b[3];
    for (int i = 0; i < 3; ++i) {
      a[i] = p.center[i] + -p.radius;
      b[i] = p.center[i] + p.radius;
    }
    return aabb_new(a, b);
  }
  if (p.kind == RT_PRIM_MOVING_SPHERE) {  // moving_sphere.rs:37-47
    Box out;
    for (double t : {t0, t1}) {
      double c[3], a[3], b[3];
      moving_center(p, t, c);
      for (int i = 0; i < 3; ++i) {
        a[i] = c[i] + -p.radius;
        b[i] = c[i] + p.radius;
      }
      out.grow(aabb_new(a, b));
    }
    return out;
  }
  // rect.rs:20-34
  double thickness = 0.001, w2 = p.width / 2.0, h2 = p.height / 2.0;
  double off[3];
  if (p.axis == 0) {
    off[0] = thickness; off[1] = w2; off[2] = h2;
  } else if (p.axis == 1) {
    off[0] = w2; off[1] = thickness; off[2] = h2;
  } else {
    off[0] = w2; off[1] = h2; off[2] = thickness;
  }
  double a[3], b[3];
  for (int i = 0; i < 3; ++i) {
    a[i] = p.center[i] + -off[i];
    b[i] = p.center[i] + off[i];
  }
  return aabb_new(a, b);
}

Box apply_op_box(const rt_op& op, const Box& in) {
  if (op.kind == RT_OP_TRANSLATE) {  // instance.rs:21-25
    double a[3], b[3];
    for (int i = 0; i < 3; ++i) {
      a[i] = in.lo[i] + op.a[i];
      b[i] = in.hi[i] + op.a[i];
    }
    return aabb_new(a, b);
  }
  // RotateY::bounding_box, instance.rs:85-101: rotate the 8 corners back to the outer frame
  double s = op.a[0], c = op.a[1];
  Box out;
  const double xs[2] = {in.lo[0], in.hi[0]}, ys[2] = {in.lo[1], in.hi[1]},
               zs[2] = {in.lo[2], in.hi[2]};
  for (double x : xs)
    for (double y : ys)
      for (double z : zs) {
        double p[3] = {c * x + s * z, y, -s * x + c * z};
        out.grow(p);
      }
  return out;
}

Box ref_object_box(const rt_scene& s, const rt_object& o, double t0, double t1) {
  Box b;
  if (o.is_cuboid) {  // cuboid.rs:45-47
    double lo[3], hi[3];
    for (int i = 0; i < 3; ++i) {
      lo[i] = o.cuboid_center[i] + -o.cuboid_half[i];
      hi[i] = o.cuboid_center[i] + o.cuboid_half[i];
    }
    b = aabb_new(lo, hi);
  } else {
    for (uint32_t k = 0; k < o.prim_count; ++k) b.grow(ref_prim_box(s.prims[o.first_prim + k], t0, t1));
  }
  for (uint32_t k = o.inner_op_count; k-- > 0;) b = apply_op_box(s.ops[o.inner_first_op + k], b);
  // ConstantMedium::bounding_box is its boundary's (constant_medium.rs:78-80)
  for (uint32_t k = o.op_count; k-- > 0;) b = apply_op_box(s.ops[o.first_op + k], b);
  return b;
}

// OrderedFloat order: NaN greatest.
bool of_less(double a, double b) {
  if (std::isnan(a)) return false;
  if (std::isnan(b)) return true;
  return a < b;
}

unsigned build_threads() {
  unsigned threads = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  if (const char* e = getenv("MI355RT_BUILD_THREADS")) threads = (unsigned)std::max(1ul, strtoul(e, nullptr, 10));
  return threads;
}

// fn(begin, end) over [0, n) in contiguous pieces, one per thread.
template <typename F>
void parallel_for(size_t n, F&& fn) {
  const unsigned threads = build_threads();
  if (threads < 2 || n < (1u << 14)) {
    fn((size_t)0, n);
    return;
  }
  std::vector<std::thread> pool;
  for (unsigned k = 1; k < threads; ++k)
    pool.emplace_back([&, k] { fn(n * k / threads, n * (k + 1) / threads); });
  fn((size_t)0, n / threads);
  for (std::thread& t : pool) t.join();
}

// Stable sort of [first, last) on up to `threads` threads: halves sorted side by side, then merged
// (std::inplace_merge keeps equal elements in order, so the result is std::stable_sort's).
template <typename It, typename Less>
void parallel_stable_sort(It first, It last, Less less, unsigned threads) {
  const size_t n = (size_t)(last - first);
  if (threads < 2 || n < (1u << 14)) {
    std::stable_sort(first, last, less);
    return;
  }
  It mid = first + (long)(n / 2);
  std::thread worker([&] { parallel_stable_sort(first, mid, less, threads / 2); });
  parallel_stable_sort(mid, last, less, threads - threads / 2);
  worker.join();
  std::inplace_merge(first, mid, last, less);
}

// Objects in the order the reference prefers among equal-t hits: BvhNode::new_along_axis
// (bvh.rs:33-77) sorts its objects by box centre along the axis (stable, NaN last), gives the
// upper half to the right child and the lower half to the left, and moves on to the next axis;
// among equal t the right subtree wins (bvh.rs:93-97) and, in a leaf list, the earlier object
// (hittable_list.rs:31).  A range of the id array is sorted in place and its two halves are
// ordered side by side: the right half's objects come first in `out`, so where each half
// writes is known before either starts.
struct RefOrder {
  struct Keyed {
    double key;
    uint32_t id;
  };
  const std::vector<Box>& boxes;
  std::vector<uint32_t>& out;
  void run(Keyed* ids, size_t n, int axis, size_t out_at, unsigned threads) {
    const size_t kMaxPerLeaf = 256;  // bvh.rs:9
    if (n < kMaxPerLeaf) {
      for (size_t i = 0; i < n; ++i) out[out_at + i] = ids[i].id;
      return;
    }
    for (size_t i = 0; i < n; ++i) {
      const Box& b = boxes[ids[i].id];
      ids[i].key = (b.lo[axis] + b.hi[axis]) / 2.0;
    }
    parallel_stable_sort(ids, ids + n, [](const Keyed& a, const Keyed& b) { return of_less(a.key, b.key); },
                         threads);
    const size_t half = n / 2;
    const int next = (axis + 1) % 3;
    // the right subtree (upper half) wins ties: it comes first
    if (threads >= 2 && n >= (1u << 14)) {
      std::thread worker([&] { run(ids + half, n - half, next, out_at, threads / 2); });
      run(ids, half, next, out_at + (n - half), threads - threads / 2);
      worker.join();
    } else {
      run(ids + half, n - half, next, out_at, 1);
      run(ids, half, next, out_at + (n - half), 1);
    }
  }
};
void reference_order(const std::vector<Box>& boxes, std::vector<uint32_t>& ids, int axis,
                     std::vector<uint32_t>& out) {
  std::vector<RefOrder::Keyed> keyed(ids.size());
  for (size_t i = 0; i < ids.size(); ++i) keyed[i] = RefOrder::Keyed{0.0, ids[i]};
  const size_t at = out.size();
  out.resize(at + ids.size());
  RefOrder{boxes, out}.run(keyed.data(), keyed.size(), axis, at, build_threads());
}

// The box chains of the reference's hierarchy (bvh.rs:33-77): one chain per leaf list, the
// boxes of the nodes from the root down to it.  A node's box is the union of its objects'
// boxes (Aabb::surrounding is exact, so unions of unions are the same numbers).
struct RefChains {
  std::vector<double> boxes;            // 6 per box
  std::vector<uint32_t> ranges;         // first, count per chain
  std::vector<uint32_t> chain_of_object;
};
void reference_chains(const std::vector<Box>& boxes, std::vector<uint32_t>& ids, int axis,
                      std::vector<Box>& path, RefChains& out) {
  const size_t kMaxPerLeaf = 256;  // bvh.rs:9
  Box all;
  for (uint32_t id : ids) all.grow(boxes[id]);
  path.push_back(all);
  if (ids.size() < kMaxPerLeaf) {
    uint32_t chain = (uint32_t)(out.ranges.size() / 2);
    out.ranges.push_back((uint32_t)(out.boxes.size() / 6));
    out.ranges.push_back((uint32_t)path.size());
    for (const Box& b : path) {
      for (int a = 0; a < 3; ++a) out.boxes.push_back(b.lo[a]);
      for (int a = 0; a < 3; ++a) out.boxes.push_back(b.hi[a]);
    }
    for (uint32_t id : ids) out.chain_of_object[id] = chain;
  } else {
    std::stable_sort(ids.begin(), ids.end(), [&](uint32_t a, uint32_t b) {
      double ka = (boxes[a].lo[axis] + boxes[a].hi[axis]) / 2.0;
      double kb = (boxes[b].lo[axis] + boxes[b].hi[axis]) / 2.0;
      return of_less(ka, kb);
    });
    size_t half = ids.size() / 2;
    std::vector<uint32_t> left(ids.begin(), ids.begin() + (long)half);
    std::vector<uint32_t> right(ids.begin() + (long)half, ids.end());
    int next = (axis + 1) % 3;
    reference_chains(boxes, right, next, path, out);
    reference_chains(boxes, left, next, path, out);
  }
  path.pop_back();
}

// Does a side of this cuboid reach outside Cuboid::bounding_box?  The Y sides take
// size.z along x and size.x along z (cuboid.rs:21-30 with rect.rs:27-31).
bool cuboid_overhangs(const rt_object& o) {
  return o.is_cuboid && o.cuboid_half[0] != o.cuboid_half[2];
}

// ---- true (conservative) boxes for the device hierarchy ---------------------

Box true_prim_box(const rt_prim& p, double t0, double t1) {
  Box b;
  if (p.kind == RT_PRIM_SPHERE) {
    double r = std::fabs(p.radius);
    for (int i = 0; i < 3; ++i) {
      b.lo[i] = p.center[i] - r;
      b.hi[i] = p.center[i] + r;
    }
    return b;
  }
  if (p.kind == RT_PRIM_MOVING_SPHERE) {
    // the centre is affine in time, so the two ends of the shutter bound it
    double r = std::fabs(p.radius);
    for (double t : {t0, t1}) {
      double c[3];
      moving_center(p, t, c);
      double lo[3], hi[3];
      for (int i = 0; i < 3; ++i) {
        lo[i] = c[i] - r;
        hi[i] = c[i] + r;
      }
      b.grow(lo);
      b.grow(hi);
    }
    return b;
  }
  return ref_prim_box(p, t0, t1);  // the rect's padded box contains its extent
}

// Outward float rounding with a relative pad: the slab test runs on these
// values in higher precision than they are stored in, so the pad, not the
// arithmetic, is what keeps it conservative.
void store_box(const Box& b, rt_bvh_node& n) {
  for (int a = 0; a < 3; ++a) {
    double lo = b.lo[a], hi = b.hi[a];
    if (!(lo <= hi)) {  // empty or NaN: never cull
      n.lo[a] = -FLT_MAX;
      n.hi[a] = FLT_MAX;
      continue;
    }
    double scale = std::fmax(std::fmax(std::fabs(lo), std::fabs(hi)), hi - lo);
    double pad = scale * (1.0 / 1048576.0) + 1e-30;
    double l = lo - pad, h = hi + pad;
    float fl = (float)l, fh = (float)h;
    if ((double)fl > l) fl = std::nextafterf(fl, -FLT_MAX);
    if ((double)fh < h) fh = std::nextafterf(fh, FLT_MAX);
    if (!std::isfinite(fl)) fl = -FLT_MAX;
    if (!std::isfinite(fh)) fh = FLT_MAX;
    n.lo[a] = fl;
    n.hi[a] = fh;
  }
}

struct BuildItem {
  Box box;
  double centroid[3];
  uint32_t slot;  // index into the pre-order item list
};

// Node of the binary SAH tree the wide hierarchy is collapsed from.
struct BinNode {
  Box box;
  uint32_t left = 0;   // children left, left + 1 (interior)
  uint32_t first = 0;  // first leaf slot (leaf)
  uint32_t count = 0;  // leaf: item count; interior: 0
};

struct Builder {
  uint32_t max_leaf = 1;
  int n_bins = 64;          // SAH bins per axis (ranges too large for the exact sweep)
  size_t sweep_max = 4096;  // ranges up to this many items try every split position
  bool parallel = true;     // large ranges: the two sides of a split on two threads
  double min_side = 0.0;  // smallest share of the items a side of a split may get
  std::vector<BuildItem>& items;
  std::vector<BinNode>& nodes;
  std::vector<uint32_t>& leaf_order;  // slots in leaf order
  uint32_t max_depth = 0;

  Builder(std::vector<BuildItem>& i, std::vector<BinNode>& n, std::vector<uint32_t>& lo)
      : items(i), nodes(n), leaf_order(lo) {}

  // Can `count` items still be split into leaves within `levels` more interior levels?
  bool fits(uint64_t count, uint32_t levels) const {
    if (levels >= 40) return true;
    return count <= ((uint64_t)max_leaf << levels);
  }

  void build(uint32_t node_index, size_t begin, size_t end, uint32_t depth) {
    max_depth = std::max(max_depth, depth);
    Box bounds, cbounds;
    for (size_t i = begin; i < end; ++i) {
      bounds.grow(items[i].box);
      cbounds.grow(items[i].centroid);
    }
    nodes[node_index].box = bounds;
    size_t count = end - begin;
    if (count <= max_leaf) {
      nodes[node_index].first = (uint32_t)leaf_order.size();
      nodes[node_index].count = (uint32_t)count;
      for (size_t i = begin; i < end; ++i) leaf_order.push_back(items[i].slot);
      return;
    }
    uint32_t levels_left = kMaxBvhDepth - depth - 1;  // levels available to each child
    int axis = 0;
    size_t mid = begin;
    bool have_split = false;

    // small ranges: exact SAH, every split position of the centroid order on every axis.
    // The orders are kept as (centroid, position) pairs: the 80-byte items move once, into the
    // order of the winning axis (the same order a stable sort of the items themselves gives).
    if (count <= sweep_max) {
      double best_cost = INF;
      int best_axis = -1;
      size_t best_left = 0;
      struct Key {
        double c;
        uint32_t at;
      };
      std::vector<double> right_area(count);
      std::vector<Key> order(count), best_order;
      for (int a = 0; a < 3; ++a) {
        if (!(cbounds.hi[a] > cbounds.lo[a])) continue;
        for (size_t i = 0; i < count; ++i) order[i] = Key{items[begin + i].centroid[a], (uint32_t)i};
        std::stable_sort(order.begin(), order.end(), [](const Key& x, const Key& y) { return x.c < y.c; });
        Box acc;
        for (size_t i = count; i-- > 1;) {
          acc.grow(items[begin + order[i].at].box);
          right_area[i] = acc.area();
        }
        Box lacc;
        bool won = false;
        for (size_t l = 1; l < count; ++l) {  // l items on the left
          lacc.grow(items[begin + order[l - 1].at].box);
          size_t r = count - l;
          if ((double)std::min(l, r) < min_side * (double)count) continue;
          if (!fits(l, levels_left) || !fits(r, levels_left)) continue;
          double cost = lacc.area() * (double)l + right_area[l] * (double)r;
          if (cost < best_cost) {
            best_cost = cost;
            best_axis = a;
            best_left = l;
            won = true;
          }
        }
        if (won) best_order = order;
      }
      if (best_axis >= 0) {
        std::vector<BuildItem> moved(count);
        for (size_t i = 0; i < count; ++i) moved[i] = items[begin + best_order[i].at];
        std::copy(moved.begin(), moved.end(), items.begin() + (long)begin);
        mid = begin + best_left;
        have_split = true;
      }
    }

    // binned SAH on the axis/plane with the lowest cost
    const int kMaxBins = 128;
    const int kBins = n_bins;
    double best_cost = INF;
    int best_axis = -1, best_plane = -1;
    for (int a = 0; a < 3 && !have_split && count > sweep_max; ++a) {
      double lo = cbounds.lo[a], hi = cbounds.hi[a];
      if (!(hi > lo) || !std::isfinite(lo) || !std::isfinite(hi)) continue;
      Box bin_box[kMaxBins];
      size_t bin_count[kMaxBins] = {0};
      double scale = (double)kBins / (hi - lo);
      for (size_t i = begin; i < end; ++i) {
        int b = (int)((items[i].centroid[a] - lo) * scale);
        b = std::min(std::max(b, 0), kBins - 1);
        bin_box[b].grow(items[i].box);
        bin_count[b]++;
      }
      double right_area[kMaxBins];
      size_t right_count[kMaxBins];
      Box acc;
      size_t cnt = 0;
      for (int b = kBins - 1; b > 0; --b) {
        acc.grow(bin_box[b]);
        cnt += bin_count[b];
        right_area[b] = acc.area();
        right_count[b] = cnt;
      }
      Box lacc;
      size_t lcnt = 0;
      for (int b = 0; b < kBins - 1; ++b) {
        lacc.grow(bin_box[b]);
        lcnt += bin_count[b];
        size_t rcnt = right_count[b + 1];
        if (lcnt == 0 || rcnt == 0) continue;
        if ((double)std::min(lcnt, rcnt) < min_side * (double)count) continue;
        if (!fits(lcnt, levels_left) || !fits(rcnt, levels_left)) continue;
        double cost = lacc.area() * (double)lcnt + right_area[b + 1] * (double)rcnt;
        if (cost < best_cost) {
          best_cost = cost;
          best_axis = a;
          best_plane = b;
        }
      }
    }
    if (best_axis >= 0) {
      axis = best_axis;
      double lo = cbounds.lo[axis], hi = cbounds.hi[axis];
      double scale = (double)kBins / (hi - lo);
      auto it = std::partition(items.begin() + (long)begin, items.begin() + (long)end,
                               [&](const BuildItem& bi) {
                                 int b = (int)((bi.centroid[axis] - lo) * scale);
                                 b = std::min(std::max(b, 0), kBins - 1);
                                 return b <= best_plane;
                               });
      mid = (size_t)(it - items.begin());
      have_split = mid > begin && mid < end;
    }
    if (!have_split) {
      // median split on the widest centroid axis: always balanced, so the depth bound holds
      double widest = -1.0;
      for (int a = 0; a < 3; ++a) {
        double ext = cbounds.hi[a] - cbounds.lo[a];
        if (ext > widest) {
          widest = ext;
          axis = a;
        }
      }
      mid = begin + count / 2;
      std::nth_element(items.begin() + (long)begin, items.begin() + (long)mid,
                       items.begin() + (long)end, [&](const BuildItem& x, const BuildItem& y) {
                         return x.centroid[axis] < y.centroid[axis];
                       });
    }
    uint32_t left = (uint32_t)nodes.size();
    nodes.emplace_back();
    nodes.emplace_back();
    nodes[node_index].left = left;
    nodes[node_index].count = 0;
    if (parallel && count >= kParallelMin && depth < kParallelDepth) {
      // the two halves are disjoint ranges of `items`: build the left one on another thread into
      // vectors of its own and splice both in, in the order the sequential recursion would have
      // produced (the same tree and the same leaf order; only the numbering of the binary nodes
      // differs, and collapse_wide renumbers them anyway)
      std::vector<BinNode> ln(1), rn(1);
      std::vector<uint32_t> ll, rl;
      Builder lb(items, ln, ll), rb(items, rn, rl);
      for (Builder* b : {&lb, &rb}) {
        b->max_leaf = max_leaf;
        b->n_bins = n_bins;
        b->sweep_max = sweep_max;
        b->min_side = min_side;
        b->parallel = parallel;
      }
      std::thread worker([&] { lb.build(0, begin, mid, depth + 1); });
      rb.build(0, mid, end, depth + 1);
      worker.join();
      splice(left, ln, ll);
      splice(left + 1, rn, rl);
      max_depth = std::max(max_depth, std::max(lb.max_depth, rb.max_depth));
      return;
    }
    build(left, begin, mid, depth + 1);
    build(left + 1, mid, end, depth + 1);
  }

  static constexpr size_t kParallelMin = 1u << 15;  // items
  static constexpr uint32_t kParallelDepth = 4;     // up to sixteen threads

  // Put a subtree built on its own (root at sub[0], leaf slots counted from 0) at node `slot`.
  void splice(uint32_t slot, const std::vector<BinNode>& sub, const std::vector<uint32_t>& sub_leaves) {
    const uint32_t node_base = (uint32_t)nodes.size();  // sub[i], i >= 1, becomes nodes[node_base + i - 1]
    const uint32_t leaf_base = (uint32_t)leaf_order.size();
    auto moved = [&](BinNode n) {
      if (n.count) n.first += leaf_base;
      else n.left = node_base + n.left - 1u;
      return n;
    };
    nodes[slot] = moved(sub[0]);
    for (size_t i = 1; i < sub.size(); ++i) nodes.push_back(moved(sub[i]));
    leaf_order.insert(leaf_order.end(), sub_leaves.begin(), sub_leaves.end());
  }
};

// Stack entries a traversal of the wide hierarchy can need: a visit pushes at most
// (children - 1) entries before descending.  Children have larger indices than their parent,
// so one backward sweep settles it.
uint32_t stack_need(const std::vector<rt_bvh_node>& out) {
  std::vector<uint32_t> need(out.size(), 0);
  for (size_t i = out.size(); i-- > 0;) {
    if (!(out[i].b & 0x80000000u)) continue;
    uint32_t k = (out[i].b & (kWide - 1u)) + 1u, deepest = 0;
    for (uint32_t c = 0; c < k; ++c) deepest = std::max(deepest, need[out[i].a + c]);
    need[i] = (k - 1u) + deepest;
  }
  return out.empty() ? 1u : std::max(1u, need[0]);
}

// Collapse the binary tree into nodes of up to four children.  Output records are
// numbered breadth-first and the children of a node are consecutive, so one visit
// reads up to 128 contiguous bytes and the head of the array is the top of the tree.
// Returns the number of stack entries a traversal can need: a visit pushes at most
// (children - 1) entries before descending.
uint32_t collapse_wide(const std::vector<BinNode>& bin, std::vector<rt_bvh_node>& out,
                       uint32_t& depth_out) {
  out.clear();
  out.emplace_back();
  store_box(bin[0].box, out[0]);
  struct Pending {
    uint32_t bin_index, record, depth;
  };
  std::vector<Pending> queue{{0u, 0u, 0u}};
  depth_out = 0;
  for (size_t q = 0; q < queue.size(); ++q) {
    Pending cur = queue[q];
    depth_out = std::max(depth_out, cur.depth);
    const BinNode& n = bin[cur.bin_index];
    if (n.count) {
      out[cur.record].a = n.first;
      out[cur.record].b = n.count;
      continue;
    }
    // expand the interior child with the largest box until there are four
    uint32_t kids[kWide] = {n.left, n.left + 1};
    uint32_t k = 2;
    while (k < kWide) {
      int pick = -1;
      double area = -1.0;
      for (uint32_t i = 0; i < k; ++i)
        if (bin[kids[i]].count == 0 && bin[kids[i]].box.area() > area) {
          area = bin[kids[i]].box.area();
          pick = (int)i;
        }
      if (pick < 0) break;
      uint32_t expanded = kids[pick];
      kids[pick] = bin[expanded].left;
      kids[k++] = bin[expanded].left + 1;
    }
    uint32_t first = (uint32_t)out.size();
    out[cur.record].a = first;
    out[cur.record].b = 0x80000000u | (k - 1u);
    for (uint32_t i = 0; i < k; ++i) {
      out.emplace_back();
      store_box(bin[kids[i]].box, out.back());
      queue.push_back({kids[i], first + i, cur.depth + 1});
    }
  }
  return stack_need(out);
}

uint32_t pack_material(uint32_t material) {
  return material < (1u << (32u - kMaterialShift)) ? material << kMaterialShift : kMaterialElsewhereBit;
}
uint32_t pack_ops(uint32_t first, uint32_t count) { return (first & 0x00FFFFFFu) | (count << 24); }

}  // namespace

std::string flatten_scene(const rt_scene& s, FlatScene& out, int& code, const WideBuild* wide,
                          const RefOrderFn* ref_order) {
  // MI355RT_DEBUG: where the time of a flatten goes
  const bool timing = getenv("MI355RT_DEBUG") != nullptr;
  auto clock0 = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!timing) return;
    auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "mi355rt: flatten %-22s %8.1f ms\n", what,
            std::chrono::duration<double, std::milli>(now - clock0).count());
    clock0 = now;
  };
  code = RT_ERR_INVALID;
  std::string bad = validate_scene(s);
  if (!bad.empty()) return "invalid scene: " + bad;
  if (s.n_ops >= (1u << 24)) {
    code = RT_ERR_LIMIT;
    return "too many instance ops (limit 16777215)";
  }
  out = FlatScene();
  out.camera = s.camera;
  std::memcpy(out.background, s.background, sizeof out.background);
  const double t0 = s.camera.time0, t1 = s.camera.time1;

  // textures, materials, tables
  out.textures.resize(s.n_textures);
  std::vector<uint8_t> tex_needs_uv(s.n_textures, 0);
  parallel_for(s.n_textures, [&](size_t t_begin, size_t t_end) {
    for (size_t i = t_begin; i < t_end; ++i) {
      const rt_texture& t = s.textures[i];
      DTexture d{};
      d.kind = t.kind;
      d.a = t.a;
      d.b = t.b;
      std::memcpy(d.color, t.color, sizeof d.color);
      d.scale = t.scale;
      out.textures[i] = d;
      if (t.kind == RT_TEX_IMAGE) tex_needs_uv[i] = 1;
    }
  });
  // children precede parents (validated), so one forward pass settles a checkerboard's flag
  for (uint32_t i = 0; i < s.n_textures; ++i)
    if (s.textures[i].kind == RT_TEX_CHECKER)
      tex_needs_uv[i] = tex_needs_uv[s.textures[i].a] | tex_needs_uv[s.textures[i].b];
  out.materials.resize(s.n_materials);
  parallel_for(s.n_materials, [&](size_t m_begin, size_t m_end) {
  for (size_t i = m_begin; i < m_end; ++i) {
    const rt_material& m = s.materials[i];
    DMaterial d{};
    d.kind = m.kind;
    d.texture = m.texture;
    std::memcpy(d.albedo, m.albedo, sizeof d.albedo);
    d.param = m.param;
    bool textured = m.kind == RT_MAT_LAMBERTIAN || m.kind == RT_MAT_DIFFUSE_LIGHT ||
                    m.kind == RT_MAT_ISOTROPIC;
    d.needs_uv = textured ? tex_needs_uv[m.texture] : 0;
    d.textured = textured && s.textures[m.texture].kind != RT_TEX_SOLID ? 1u : 0u;
    if (textured && !d.textured) std::memcpy(d.albedo, s.textures[m.texture].color, sizeof d.albedo);
    // (a Dielectric has no albedo: the slot carries 1 / index_of_refraction, dielectric.rs:16-20)
    if (m.kind == RT_MAT_DIELECTRIC) d.albedo[0] = 1.0 / m.param;
    out.materials[i] = d;
  }
  });
  out.perlin.assign(s.perlin, s.perlin + s.n_perlin);
  out.images.resize(s.n_images);
  for (uint32_t i = 0; i < s.n_images; ++i)
    out.images[i] = DImage{s.images[i].width, s.images[i].height, s.images[i].offset};
  out.texels.assign(s.texels, s.texels + s.n_texel_bytes);
  out.ops.resize(s.n_ops);
  for (uint32_t i = 0; i < s.n_ops; ++i) {
    DOp d{};
    d.kind = s.ops[i].kind;
    std::memcpy(d.a, s.ops[i].a, sizeof d.a);
    out.ops[i] = d;
  }

  lap("materials, ops");
  // reference preference order -> rank of the first primitive of every object
  std::vector<Box> ref_boxes(s.n_objects);
  parallel_for(s.n_objects, [&](size_t b, size_t e) {
    for (size_t i = b; i < e; ++i) ref_boxes[i] = ref_object_box(s, s.objects[i], t0, t1);
  });
  std::vector<uint32_t> order;
  if (ref_order && s.n_objects >= 256) {
    std::vector<double> centres((size_t)3 * s.n_objects);
    parallel_for(s.n_objects, [&](size_t b, size_t e) {
      for (size_t i = b; i < e; ++i)
        for (int a = 0; a < 3; ++a)  // the sort key of bvh.rs:47-53
          centres[(size_t)a * s.n_objects + i] = (ref_boxes[i].lo[a] + ref_boxes[i].hi[a]) / 2.0;
    });
    std::string err = (*ref_order)(centres, s.n_objects, order);
    if (!err.empty()) {  // the host's recursive sorts give the same order; the caller clears the device's error state
      if (getenv("MI355RT_DEBUG")) fprintf(stderr, "mi355rt: device reference order failed (%s): host sorts instead\n", err.c_str());
      std::vector<uint32_t> ids(s.n_objects);
      std::iota(ids.begin(), ids.end(), 0u);
      order.clear();
      order.reserve(s.n_objects);
      reference_order(ref_boxes, ids, 0, order);
    } else if (getenv("MI355RT_CHECK_ORDER")) {  // development: the host's sorts must give the same order
      std::vector<uint32_t> ids(s.n_objects), host_order;
      std::iota(ids.begin(), ids.end(), 0u);
      reference_order(ref_boxes, ids, 0, host_order);
      if (host_order != order) return "device reference order differs from the host's";
    }
  } else {
    std::vector<uint32_t> ids(s.n_objects);
    std::iota(ids.begin(), ids.end(), 0u);
    order.reserve(s.n_objects);
    reference_order(ref_boxes, ids, 0, order);
  }
  std::vector<uint32_t> rank_base(s.n_objects, 0);
  {
    uint64_t running = 0;
    for (uint32_t obj : order) {
      rank_base[obj] = (uint32_t)running;
      running += s.objects[obj].prim_count;
      if (running >= 0xFFFFFFFFull) {
        code = RT_ERR_LIMIT;
        return "too many primitives";
      }
    }
  }

  lap("reference order");
  // box chains of the reference's hierarchy, only for scenes that need them
  RefChains chains;
  {
    bool any = false;
    for (uint32_t i = 0; i < s.n_objects && !any; ++i) any = cuboid_overhangs(s.objects[i]);
    if (any) {
      chains.chain_of_object.assign(s.n_objects, 0);
      std::vector<uint32_t> all(s.n_objects);
      std::iota(all.begin(), all.end(), 0u);
      std::vector<Box> path;
      reference_chains(ref_boxes, all, 0, path, chains);
      out.chain_boxes = chains.boxes;
      out.chain_ranges = chains.ranges;
    }
  }

  // Material kinds shaded out of line whatever their texture (bit k = rt_material_kind k).
  // Off by default: on book_cover, deferring Metal and Dielectric took 2 ms off the fused
  // kernel and cost 6.7 ms in the deferred one (its gathers of path state).
  uint32_t defer_kinds = 0;
  if (const char* e = getenv("MI355RT_DEFER_KINDS")) defer_kinds = (uint32_t)strtoul(e, nullptr, 0) & 7u;

  // `aux_at`: where a moving sphere's second centre goes in out.aux (slots are made on several
  // threads, so every object's share of the side arrays is laid out beforehand)
  auto make_record = [&](const rt_prim& p, uint32_t ops_word, uint32_t object, uint32_t& aux_at, bool& defers) {
    DPrim d{};
    const rt_material& m = s.materials[p.material];
    uint32_t needs_uv = out.materials[p.material].needs_uv ? 1u : 0u;
    d.material = p.material;
    d.object = object;
    d.ops = ops_word;
    if (p.kind == RT_PRIM_SPHERE) {
      d.kind = DP_SPHERE;
      d.p[0] = p.center[0];
      d.p[1] = p.center[1];
      d.p[2] = p.center[2];
      d.p[3] = p.radius;
      d.p[4] = 1.0 / p.radius;  // the reciprocal `(p - center) / radius` multiplies by (vec3.rs:214-219)
    } else if (p.kind == RT_PRIM_MOVING_SPHERE) {
      d.kind = DP_MOVING_SPHERE;
      d.p[0] = p.center[0];
      d.p[1] = p.center[1];
      d.p[2] = p.center[2];
      d.p[3] = p.radius;
      d.p[4] = p.time0;
      d.p5 = (double)aux_at;
      out.aux[aux_at++] = DAux{{p.center1[0], p.center1[1], p.center1[2], p.time1}};
    } else {
      d.kind = DP_RECT | (p.axis << 4);
      Box b = ref_prim_box(p, 0.0, 0.0);  // AxisAlignedRect::bounding_box, rect.rs:20-34
      int d0 = (int)p.axis, d1 = (d0 + 1) % 3, d2 = (d1 + 1) % 3;
      d.p[0] = p.center[d0];
      d.p[1] = b.lo[d1];
      d.p[2] = b.hi[d1];
      d.p[3] = b.lo[d2];
      d.p[4] = b.hi[d2];
      const rt_object& o = s.objects[object];
      if (!o.is_medium && p.axis == 1 && cuboid_overhangs(o)) {
        d.kind |= kChainBit;
        d.p5 = (double)chains.chain_of_object[object];
      }
    }
    d.kind |= (m.kind & 7u) << 8;
    d.kind |= needs_uv << 11;
    if (out.materials[p.material].textured || ((defer_kinds >> (m.kind & 7u)) & 1u)) {
      d.kind |= kDeferBit;
      defers = true;
    }
    d.kind |= pack_material(p.material);
    return d;
  };

  // Leaf slots in pre-order (object order), then the build permutes them.  Where each object's
  // slots, boundary primitives and moving-sphere centres go: one cheap pass; the records and
  // boxes themselves: in parallel.
  std::vector<uint32_t> slot_base(s.n_objects + 1, 0), boundary_base(s.n_objects + 1, 0), aux_base(s.n_objects + 1, 0);
  for (uint32_t oi = 0; oi < s.n_objects; ++oi) {
    const rt_object& o = s.objects[oi];
    uint32_t moving = 0;
    for (uint32_t k = 0; k < o.prim_count; ++k)
      moving += s.prims[o.first_prim + k].kind == RT_PRIM_MOVING_SPHERE ? 1u : 0u;
    slot_base[oi + 1] = slot_base[oi] + (o.is_medium ? 1u : o.prim_count);
    boundary_base[oi + 1] = boundary_base[oi] + (o.is_medium ? o.prim_count : 0u);
    aux_base[oi + 1] = aux_base[oi] + moving;
    if (o.is_medium) out.has_media = true;
  }
  const size_t n_slots = slot_base[s.n_objects];
  std::vector<DPrim> slots(n_slots);
  std::vector<rt_bvh_item> slot_items(n_slots);
  std::vector<BuildItem> build_items(n_slots);
  out.boundary.resize(boundary_base[s.n_objects]);
  out.aux.resize(aux_base[s.n_objects]);
  std::atomic<bool> any_defers{false};
  parallel_for(s.n_objects, [&](size_t o_begin, size_t o_end) {
    bool defers = false;
    for (size_t oi = o_begin; oi < o_end; ++oi) {
      const rt_object& o = s.objects[oi];
      uint32_t aux_at = aux_base[oi];
      auto through_ops = [&](Box b) {
        for (uint32_t k = o.inner_op_count; k-- > 0;) b = apply_op_box(s.ops[o.inner_first_op + k], b);
        for (uint32_t k = o.op_count; k-- > 0;) b = apply_op_box(s.ops[o.first_op + k], b);
        return b;
      };
      auto finish = [&](BuildItem& bi, uint32_t slot) {
        bi.slot = slot;
        for (int a = 0; a < 3; ++a) {
          double c = 0.5 * bi.box.lo[a] + 0.5 * bi.box.hi[a];
          bi.centroid[a] = std::isfinite(c) ? c : 0.0;
        }
      };
      if (o.is_medium) {
        DPrim d{};
        d.kind = DP_MEDIUM | ((uint32_t)RT_MAT_ISOTROPIC << 8);
        d.kind |= (out.materials[o.medium_material].needs_uv ? 1u : 0u) << 11;
        if (out.materials[o.medium_material].textured) {
          d.kind |= kDeferBit;
          defers = true;
        }
        d.kind |= pack_material(o.medium_material);
        d.ops = pack_ops(o.first_op, o.op_count);
        d.material = o.medium_material;
        d.object = (uint32_t)oi;
        d.p[0] = o.neg_inv_density;
        d.p[1] = (double)boundary_base[oi];
        d.p[2] = (double)o.prim_count;
        if (cuboid_overhangs(o)) {  // the medium as a whole answers to its object's chain
          d.kind |= kChainBit;
          d.p[3] = (double)chains.chain_of_object[oi];
        }
        Box b;
        for (uint32_t k = 0; k < o.prim_count; ++k) {
          const rt_prim& p = s.prims[o.first_prim + k];
          out.boundary[boundary_base[oi] + k] =
              make_record(p, pack_ops(o.inner_first_op, o.inner_op_count), (uint32_t)oi, aux_at, defers);
          b.grow(true_prim_box(p, t0, t1));
        }
        const uint32_t slot = slot_base[oi];
        build_items[slot].box = through_ops(b);
        finish(build_items[slot], slot);
        slots[slot] = d;
        slot_items[slot] = rt_bvh_item{(uint32_t)oi, kNoPrim, rank_base[oi], 0};
      } else {
        for (uint32_t k = 0; k < o.prim_count; ++k) {
          const rt_prim& p = s.prims[o.first_prim + k];
          const uint32_t slot = slot_base[oi] + k;
          build_items[slot].box = through_ops(true_prim_box(p, t0, t1));
          finish(build_items[slot], slot);
          slots[slot] = make_record(p, pack_ops(o.first_op, o.op_count), (uint32_t)oi, aux_at, defers);
          slot_items[slot] = rt_bvh_item{(uint32_t)oi, o.first_prim + k, rank_base[oi] + k, 0};
        }
      }
    }
    if (defers) any_defers = true;
  });
  out.defers = any_defers;

  lap("records, boxes");
  if (!build_items.empty()) {
    std::vector<uint32_t> leaf_order;
    bool built = false;
    if (wide && build_items.size() >= 2) {
      std::vector<rt_bvh_node> leaf_boxes(build_items.size());
      parallel_for(build_items.size(), [&](size_t b, size_t e) {
        for (size_t i = b; i < e; ++i) {
          leaf_boxes[i].a = leaf_boxes[i].b = 0;
          store_box(build_items[i].box, leaf_boxes[i]);
        }
      });
      std::string err = (*wide)(leaf_boxes, out.nodes, leaf_order, out.bvh_depth);
      if (err.empty()) {
        lap("device build");
        out.stack_entries = stack_need(out.nodes);
        lap("stack bound");
        built = true;
        // Clustering has no depth bound of its own (nested boxes merge one leaf at a time): a
        // hierarchy whose walk could need more than 64 stack entries is left to the host's
        // build, which bounds its depth by construction.
        if (out.stack_entries > 64) {
          if (timing) fprintf(stderr, "mi355rt: device-built hierarchy needs %u stack entries; building on the host\n",
                              out.stack_entries);
          built = false;
          out.nodes.clear();
          leaf_order.clear();
        }
      } else {  // the host's build serves every scene
        if (timing) fprintf(stderr, "mi355rt: %s; building on the host\n", err.c_str());
        out.nodes.clear();
        leaf_order.clear();
      }
    }
    if (!built) {
      std::vector<BinNode> bin;
      bin.reserve(build_items.size() * 2);
      bin.emplace_back();
      leaf_order.reserve(build_items.size());
      Builder b(build_items, bin, leaf_order);
      if (const char* e = getenv("MI355RT_SAH_MIN_SIDE")) b.min_side = strtod(e, nullptr);
      if (const char* e = getenv("MI355RT_SAH_BINS")) {
        long v = strtol(e, nullptr, 10);
        if (v >= 2 && v <= 128) b.n_bins = (int)v;
      }
      if (const char* e = getenv("MI355RT_SAH_SWEEP")) b.sweep_max = strtoul(e, nullptr, 10);
      if (const char* e = getenv("MI355RT_BUILD_THREADS")) b.parallel = strtoul(e, nullptr, 10) != 1;
      if (const char* e = getenv("MI355RT_LEAF_ITEMS")) {
        unsigned long v = strtoul(e, nullptr, 10);
        if (v >= 1 && v <= kMaxLeafItems) b.max_leaf = (uint32_t)v;
      }
      b.build(0, 0, build_items.size(), 0);
      lap("SAH build");
      out.stack_entries = collapse_wide(bin, out.nodes, out.bvh_depth);
      lap("collapse");
    }
    out.prims.resize(leaf_order.size());
    out.items.resize(leaf_order.size());
    out.ranks.resize(leaf_order.size());
    parallel_for(leaf_order.size(), [&](size_t b, size_t e) {
      for (size_t i = b; i < e; ++i) {
        const uint32_t slot = leaf_order[i];
        out.prims[i] = slots[slot];
        out.items[i] = slot_items[slot];
        out.ranks[i] = slot_items[slot].rank;
      }
    });
  }
  lap("leaf order");
  if (out.items.size() >= (1u << 26) || out.nodes.size() >= (1u << 27)) {
    code = RT_ERR_LIMIT;
    return "acceleration structure too large for the packed traversal entries";
  }
  code = RT_OK;
  return "";
}

}  // namespace rt
