// The wide hierarchy built on the device, for scenes large enough that the host's SAH
// build would dominate an upload (the million-sphere cover scene: 0.6 s on sixteen host threads).
//
// What has to be preserved is results, not topology: the reference's own BVH
// (BvhNode::new_along_axis, src/hittable/bvh.rs:33-77) only decides which of several objects
// hit at exactly the same t is returned, and that order reaches the device as per-slot ranks
// whichever hierarchy it walks (reference_order on the host, device_reference_order below).
// Any conservative hierarchy over the same leaf boxes returns the same hits.
//
//   1. Morton keys (21 bits per axis) of the leaf-box centres, radix-sorted (rocPRIM).
//   2. Binary tree bottom-up by locally-ordered clustering: every cluster, at first the leaves
//      in key order, looks a few places to either side for the neighbour whose union with it
//      has the smallest surface; clusters that choose each other merge; repeat until one is left.
//   3. Collapse to records of up to four children, breadth-first, the children of a record
//      consecutive: the same rule as the host's collapse_wide (expand the interior child with
//      the largest box), one level per step with a prefix sum over the frontier.
// The records come back to the host, which derives the stack bound and uploads them with the
// rest of the flattened scene like a host-built hierarchy.
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <cstdlib>
#include <string>
#include <vector>

#include "bvh_device.h"

namespace rt {
namespace {

const uint32_t kLeafBit = 0x80000000u;
const int kThreads = 256;

__device__ __forceinline__ uint64_t spread21(uint32_t v) {  // bit i of v to bit 3 i
  uint64_t x = v & 0x1FFFFFu;
  x = (x | (x << 32)) & 0x1F00000000FFFFull;
  x = (x | (x << 16)) & 0x1F0000FF0000FFull;
  x = (x | (x << 8)) & 0x100F00F00F00F00Full;
  x = (x | (x << 4)) & 0x10C30C30C30C30C3ull;
  x = (x | (x << 2)) & 0x1249249249249249ull;
  return x;
}

struct Bounds6 {
  float lo[3], hi[3];
};

__global__ void __launch_bounds__(kThreads)
morton_kernel(const rt_bvh_node* leaves, uint32_t n, Bounds6 cb, uint64_t* keys, uint32_t* ids) {
  const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= n) return;
  const rt_bvh_node b = leaves[i];
  uint32_t q[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float c = 0.5f * b.lo[a] + 0.5f * b.hi[a];
    if (!(fabsf(c) <= 3.0e38f)) c = 0.0f;  // a box that never culls has no centre
    const float ext = cb.hi[a] - cb.lo[a];
    float u = ext > 0.0f ? (c - cb.lo[a]) / ext : 0.0f;
    u = fminf(fmaxf(u, 0.0f), 1.0f);
    q[a] = min((uint32_t)(u * 2097152.0f), 2097151u);
  }
  keys[i] = (spread21(q[0]) << 2) | (spread21(q[1]) << 1) | spread21(q[2]);
  ids[i] = i;
}

__device__ __forceinline__ Bounds6 box_of(uint32_t ref, const Bounds6* node_box, const rt_bvh_node* leaves,
                                          const uint32_t* ids) {
  if (ref & kLeafBit) {
    const rt_bvh_node b = leaves[ids[ref & ~kLeafBit]];
    Bounds6 r;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      r.lo[a] = b.lo[a];
      r.hi[a] = b.hi[a];
    }
    return r;
  }
  return node_box[ref];
}

// ------------------------------------------------------------ clustering
// The binary tree bottom-up instead (Meister and Bittner, "Parallel locally-ordered clustering
// for bounding volume hierarchy construction", TVCG 2018): the clusters, at first the leaves in
// Morton order, each look a few places to either side for the neighbour whose union with them
// has the smallest surface; two clusters that choose each other merge into a new node; the
// array is compacted and the step repeats until one cluster is left.  Merging by surface area
// instead of by key prefix gives a tree close to a top-down SAH build's (on the million-sphere
// cover scene 49.5 box tests per ray against 47.1 for the host's sweep build and 55.0 for a radix tree
// over the same keys with boxes fitted afterwards; render time as with the host's build).
const int kMaxRadius = 32;

__global__ void __launch_bounds__(kThreads)
leaf_clusters_kernel(const rt_bvh_node* leaves, const uint32_t* ids, uint32_t n, uint32_t* ref, Bounds6* box) {
  const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= n) return;
  ref[i] = i | kLeafBit;
  const rt_bvh_node b = leaves[ids[i]];
  Bounds6 r;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    r.lo[a] = b.lo[a];
    r.hi[a] = b.hi[a];
  }
  box[i] = r;
}

__device__ __forceinline__ float union_area(const Bounds6& a, const Bounds6& b) {
  const float x = fmaxf(a.hi[0], b.hi[0]) - fminf(a.lo[0], b.lo[0]);
  const float y = fmaxf(a.hi[1], b.hi[1]) - fminf(a.lo[1], b.lo[1]);
  const float z = fmaxf(a.hi[2], b.hi[2]) - fminf(a.lo[2], b.lo[2]);
  return x * y + y * z + z * x;
}

// nearest[i]: the position within `radius` of i whose union with cluster i is smallest.  A block
// works on its 256 clusters from a copy in LDS that reaches `radius` beyond both ends.
__global__ void __launch_bounds__(kThreads)
nearest_kernel(const Bounds6* box, uint32_t n, int radius, uint32_t* nearest) {
  __shared__ Bounds6 s_box[kThreads + 2 * kMaxRadius];
  const int first = (int)(blockIdx.x * kThreads) - radius;
  for (int k = (int)threadIdx.x; k < kThreads + 2 * radius; k += kThreads) {
    const int g = first + k;
    if (g >= 0 && g < (int)n) s_box[k] = box[g];
  }
  __syncthreads();
  const int i = (int)(blockIdx.x * kThreads + threadIdx.x);
  if (i >= (int)n) return;
  const Bounds6 mine = s_box[threadIdx.x + radius];
  // Candidates by growing distance, the partner i ^ 1 first: among equal surfaces the earlier
  // candidate wins, so that identical boxes (coincident objects) pair up two by two instead of
  // all choosing the same lowest position, which would merge one pair per round.
  float best = __int_as_float(0x7F800000);
  int best_j = -1;
  const int toward = (i & 1) ? -1 : 1;
  for (int k = 1; k <= radius; ++k) {
#pragma unroll
    for (int side = 0; side < 2; ++side) {
      const int dj = side == 0 ? toward * k : -toward * k;
      const int j = i + dj;
      if (j < 0 || j >= (int)n) continue;
      float a = union_area(mine, s_box[(int)threadIdx.x + radius + dj]);
      if (!(a < 3.0e38f)) a = 3.0e38f;  // (boxes that never cull have no finite surface: any partner will do)
      if (a < best) {
        best = a;
        best_j = j;
      }
    }
  }
  nearest[i] = (uint32_t)best_j;
}

// Mutual pairs merge: the lower position keeps the new node, the upper one leaves the array.
__global__ void __launch_bounds__(kThreads)
merge_kernel(const uint32_t* ref, const Bounds6* box, const uint32_t* nearest, uint32_t n, uint32_t* node_count,
             uint32_t* left, uint32_t* right, Bounds6* node_box, uint32_t* out_ref, Bounds6* out_box,
             uint32_t* keep) {
  const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= n) return;
  const uint32_t j = nearest[i];
  const bool mutual = j < n && nearest[j] == i;
  if (!mutual) {
    out_ref[i] = ref[i];
    out_box[i] = box[i];
    keep[i] = 1u;
    return;
  }
  if (i > j) {
    keep[i] = 0u;
    return;
  }
  const Bounds6 a = box[i], b = box[j];
  Bounds6 u;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    u.lo[k] = fminf(a.lo[k], b.lo[k]);
    u.hi[k] = fmaxf(a.hi[k], b.hi[k]);
  }
  const uint32_t node = atomicAdd(node_count, 1u);
  left[node] = ref[i];
  right[node] = ref[j];
  node_box[node] = u;
  out_ref[i] = node;
  out_box[i] = u;
  keep[i] = 1u;
}

__global__ void __launch_bounds__(kThreads)
compact_kernel(const uint32_t* in_ref, const Bounds6* in_box, const uint32_t* keep, const uint32_t* offset,
               uint32_t n, uint32_t* ref, Bounds6* box) {
  const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= n || !keep[i]) return;
  ref[offset[i]] = in_ref[i];
  box[offset[i]] = in_box[i];
}

__device__ __forceinline__ float half_area(const Bounds6& b) {
  const float x = b.hi[0] - b.lo[0], y = b.hi[1] - b.lo[1], z = b.hi[2] - b.lo[2];
  return x * y + y * z + z * x;
}

// One frontier entry = one record of the wide hierarchy and the radix-tree node (or leaf) it
// stands for.  Pick its children: start from the node's two, and while there are fewer than
// four expand the interior one with the largest box.
__global__ void __launch_bounds__(kThreads)
pick_children_kernel(const uint32_t* front_ref, uint32_t n_front, const uint32_t* left, const uint32_t* right,
                     const Bounds6* node_box, uint32_t* kids, uint32_t* kid_count) {
  const uint32_t f = blockIdx.x * kThreads + threadIdx.x;
  if (f >= n_front) return;
  const uint32_t ref = front_ref[f];
  if (ref & kLeafBit) {
    kid_count[f] = 0;
    return;
  }
  uint32_t k[4] = {left[ref], right[ref], 0, 0};
  uint32_t n = 2;
  while (n < 4) {
    int pick = -1;
    float area = -1.0f;
    for (uint32_t i = 0; i < n; ++i)
      if (!(k[i] & kLeafBit)) {
        const float a = half_area(node_box[k[i]]);
        if (a > area) {
          area = a;
          pick = (int)i;
        }
      }
    if (pick < 0) break;
    const uint32_t expanded = k[pick];
    k[pick] = left[expanded];
    k[n++] = right[expanded];
  }
  for (uint32_t i = 0; i < 4; ++i) kids[4 * f + i] = k[i];
  kid_count[f] = n;
}

// Write the frontier's records and the next frontier.  `offset` = exclusive prefix sum of
// kid_count; the children of this level's records start at record `next_base`.
__global__ void __launch_bounds__(kThreads)
emit_level_kernel(const uint32_t* front_ref, const uint32_t* front_rec, uint32_t n_front, const uint32_t* kids,
                  const uint32_t* kid_count, const uint32_t* offset, uint32_t next_base, const Bounds6* node_box,
                  const rt_bvh_node* leaves, const uint32_t* ids, rt_bvh_node* records, uint32_t* next_ref,
                  uint32_t* next_rec) {
  const uint32_t f = blockIdx.x * kThreads + threadIdx.x;
  if (f >= n_front) return;
  const uint32_t ref = front_ref[f], rec = front_rec[f];
  if (ref & kLeafBit) {  // leaf record: one item, the leaf's sorted position
    records[rec].a = ref & ~kLeafBit;
    records[rec].b = 1u;
    return;
  }
  const uint32_t n = kid_count[f], first = next_base + offset[f];
  records[rec].a = first;
  records[rec].b = 0x80000000u | (n - 1u);
  for (uint32_t i = 0; i < n; ++i) {
    const uint32_t kid = kids[4 * f + i];
    const Bounds6 b = box_of(kid, node_box, leaves, ids);
    rt_bvh_node r;
    for (int a = 0; a < 3; ++a) {
      r.lo[a] = b.lo[a];
      r.hi[a] = b.hi[a];
    }
    r.a = 0;
    r.b = 0;
    records[first + i] = r;
    next_ref[offset[f] + i] = kid;
    next_rec[offset[f] + i] = first + i;
  }
}

__global__ void root_record_kernel(const Bounds6* node_box, uint32_t root, rt_bvh_node* records) {
  rt_bvh_node r;
  for (int a = 0; a < 3; ++a) {
    r.lo[a] = node_box[root].lo[a];
    r.hi[a] = node_box[root].hi[a];
  }
  r.a = 0;
  r.b = 0;
  records[0] = r;
}

// Device scratch of one build: sizes are collected first (plan), then one allocation is carved
// (256-byte aligned pieces) and freed when the object goes out of scope.
struct Scratch {
  struct Want {
    void** out;
    size_t bytes;
  };
  std::vector<Want> wants;
  void* base = nullptr;
  ~Scratch() {
    if (base) (void)hipFree(base);
  }
  template <typename T>
  void plan(T** out, size_t count) {
    wants.push_back(Want{(void**)out, (std::max<size_t>(count * sizeof(T), 64) + 255u) & ~(size_t)255u});
  }
  hipError_t allocate() {
    size_t total = 0;
    for (const Want& w : wants) total += w.bytes;
    hipError_t e = hipMalloc(&base, total);
    if (e != hipSuccess) return e;
    unsigned char* at = (unsigned char*)base;
    for (const Want& w : wants) {
      *w.out = at;
      at += w.bytes;
    }
    return hipSuccess;
  }
};

#define LB_HIP(call)                                                              \
  do {                                                                            \
    hipError_t e__ = (call);                                                      \
    if (e__ != hipSuccess) return std::string(#call) + ": " + hipGetErrorString(e__); \
  } while (0)

uint32_t blocks(size_t n) { return (uint32_t)((n + kThreads - 1) / kThreads); }

}  // namespace

std::string device_build_wide(hipStream_t stream, const std::vector<rt_bvh_node>& leaf_boxes,
                              std::vector<rt_bvh_node>& records, std::vector<uint32_t>& leaf_order,
                              uint32_t& depth) {
  const size_t n = leaf_boxes.size();
  if (n < 2 || n >= (1u << 26)) return "device build: item count out of range";
  // bounds of the box centres (host: one pass over what is about to be copied anyway)
  Bounds6 cb;
  for (int a = 0; a < 3; ++a) {
    cb.lo[a] = 3.0e38f;
    cb.hi[a] = -3.0e38f;
  }
  for (const rt_bvh_node& b : leaf_boxes)
    for (int a = 0; a < 3; ++a) {
      float c = 0.5f * b.lo[a] + 0.5f * b.hi[a];
      if (!(std::fabs(c) <= 3.0e38f)) c = 0.0f;
      cb.lo[a] = std::min(cb.lo[a], c);
      cb.hi[a] = std::max(cb.hi[a], c);
    }
  Scratch mem;
  rt_bvh_node *d_leaves = nullptr, *d_records = nullptr;
  uint64_t *d_keys = nullptr, *d_keys_sorted = nullptr;
  uint32_t *d_ids = nullptr, *d_ids_sorted = nullptr, *d_left = nullptr, *d_right = nullptr;
  uint32_t* d_arrived = nullptr;  // nodes made so far
  Bounds6* d_node_box = nullptr;
  const size_t max_records = 2 * n + 4;
  mem.plan(&d_leaves, n);
  mem.plan(&d_records, max_records);
  mem.plan(&d_keys, n);
  mem.plan(&d_keys_sorted, n);
  mem.plan(&d_ids, n);
  mem.plan(&d_ids_sorted, n);
  mem.plan(&d_left, n);
  mem.plan(&d_right, n);
  mem.plan(&d_arrived, 1);
  mem.plan(&d_node_box, n);
  // frontier buffers, ping-pong: a level has at most n entries of each kind
  uint32_t *d_ref[2] = {nullptr, nullptr}, *d_rec[2] = {nullptr, nullptr}, *d_kids = nullptr, *d_count = nullptr,
           *d_offset = nullptr;
  for (int k = 0; k < 2; ++k) {
    mem.plan(&d_ref[k], max_records);
    mem.plan(&d_rec[k], max_records);
  }
  uint32_t *d_cref[2] = {nullptr, nullptr}, *d_mref = nullptr, *d_nearest = nullptr;
  Bounds6 *d_cbox[2] = {nullptr, nullptr}, *d_mbox = nullptr;
  for (int k = 0; k < 2; ++k) {
    mem.plan(&d_cref[k], n);
    mem.plan(&d_cbox[k], n);
  }
  mem.plan(&d_mref, n);
  mem.plan(&d_mbox, n);
  mem.plan(&d_nearest, n);
  mem.plan(&d_kids, 4 * max_records);
  mem.plan(&d_count, max_records + 1);
  mem.plan(&d_offset, max_records + 1);
  size_t sort_bytes = 0, scan_bytes = 0;
  LB_HIP(rocprim::radix_sort_pairs(nullptr, sort_bytes, d_keys, d_keys_sorted, d_ids, d_ids_sorted, n, 0, 63, stream));
  LB_HIP(rocprim::exclusive_scan(nullptr, scan_bytes, d_count, d_offset, 0u, max_records + 1,
                                 rocprim::plus<uint32_t>(), stream));
  unsigned char* d_tmp = nullptr;
  mem.plan(&d_tmp, std::max(sort_bytes, scan_bytes));
  LB_HIP(mem.allocate());

  LB_HIP(hipMemcpyAsync(d_leaves, leaf_boxes.data(), n * sizeof(rt_bvh_node), hipMemcpyHostToDevice, stream));
  morton_kernel<<<blocks(n), kThreads, 0, stream>>>(d_leaves, (uint32_t)n, cb, d_keys, d_ids);
  LB_HIP(rocprim::radix_sort_pairs(d_tmp, sort_bytes, d_keys, d_keys_sorted, d_ids, d_ids_sorted, n, 0, 63, stream));
  uint32_t root = 0u;
  {
    // clustering: the cluster arrays ping-pong between (ref, box) pairs; d_arrived counts the nodes
    uint32_t n_cur = (uint32_t)n;
    LB_HIP(hipMemsetAsync(d_arrived, 0, sizeof(uint32_t), stream));
    leaf_clusters_kernel<<<blocks(n), kThreads, 0, stream>>>(d_leaves, d_ids_sorted, n_cur, d_cref[0], d_cbox[0]);
    int cc = 0;
    uint32_t steps = 0;
    int radius = 4;  // (measured on the cover scene: 3-4 gives the host build's box tests per ray, 10 and more 10-20 % more)
    if (const char* e = getenv("MI355RT_CLUSTER_RADIUS")) radius = std::min(std::max(atoi(e), 1), kMaxRadius);
    while (n_cur > 1) {
      nearest_kernel<<<blocks(n_cur), kThreads, 0, stream>>>(d_cbox[cc], n_cur, radius, d_nearest);
      merge_kernel<<<blocks(n_cur), kThreads, 0, stream>>>(d_cref[cc], d_cbox[cc], d_nearest, n_cur, d_arrived, d_left,
                                                           d_right, d_node_box, d_mref, d_mbox, d_count);
      LB_HIP(hipMemsetAsync(d_count + n_cur, 0, sizeof(uint32_t), stream));
      LB_HIP(rocprim::exclusive_scan(d_tmp, scan_bytes, d_count, d_offset, 0u, (size_t)n_cur + 1,
                                     rocprim::plus<uint32_t>(), stream));
      compact_kernel<<<blocks(n_cur), kThreads, 0, stream>>>(d_mref, d_mbox, d_count, d_offset, n_cur, d_cref[cc ^ 1],
                                                             d_cbox[cc ^ 1]);
      uint32_t left_over = 0;
      LB_HIP(hipMemcpyAsync(&left_over, d_offset + n_cur, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
      LB_HIP(hipStreamSynchronize(stream));
      if (left_over == 0 || left_over >= n_cur) return "device build: clustering made no progress";
      n_cur = left_over;
      cc ^= 1;
      if (++steps > 4096u) return "device build: clustering did not end within 4096 rounds";
    }
    root = (uint32_t)n - 2u;  // the last merge made the last node
  }
  root_record_kernel<<<1, 1, 0, stream>>>(d_node_box, root, d_records);
  const uint32_t root_ref = root, root_rec = 0u;
  LB_HIP(hipMemcpyAsync(d_ref[0], &root_ref, sizeof(uint32_t), hipMemcpyHostToDevice, stream));
  LB_HIP(hipMemcpyAsync(d_rec[0], &root_rec, sizeof(uint32_t), hipMemcpyHostToDevice, stream));
  uint32_t n_front = 1, n_records = 1, level = 0;
  int cur = 0;
  while (n_front) {
    pick_children_kernel<<<blocks(n_front), kThreads, 0, stream>>>(d_ref[cur], n_front, d_left, d_right, d_node_box,
                                                                   d_kids, d_count);
    // (one element past the frontier, so that offset[n_front] is the level's total)
    LB_HIP(hipMemsetAsync(d_count + n_front, 0, sizeof(uint32_t), stream));
    LB_HIP(rocprim::exclusive_scan(d_tmp, scan_bytes, d_count, d_offset, 0u, (size_t)n_front + 1,
                                   rocprim::plus<uint32_t>(), stream));
    emit_level_kernel<<<blocks(n_front), kThreads, 0, stream>>>(d_ref[cur], d_rec[cur], n_front, d_kids, d_count,
                                                                d_offset, n_records, d_node_box, d_leaves,
                                                                d_ids_sorted, d_records, d_ref[cur ^ 1],
                                                                d_rec[cur ^ 1]);
    uint32_t total = 0;
    LB_HIP(hipMemcpyAsync(&total, d_offset + n_front, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    LB_HIP(hipStreamSynchronize(stream));
    if ((size_t)n_records + total > max_records) return "device build: record count exceeds its bound";
    n_records += total;
    n_front = total;
    cur ^= 1;
    if (total) ++level;
    if (level > 4096) return "device build: hierarchy deeper than 4096 levels";
  }
  LB_HIP(hipGetLastError());
  depth = level;
  records.resize(n_records);
  leaf_order.resize(n);
  LB_HIP(hipMemcpyAsync(records.data(), d_records, (size_t)n_records * sizeof(rt_bvh_node), hipMemcpyDeviceToHost,
                        stream));
  LB_HIP(hipMemcpyAsync(leaf_order.data(), d_ids_sorted, n * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
  LB_HIP(hipStreamSynchronize(stream));
  return "";
}

// ---------------------------------------------------------------- reference order
// BvhNode::new_along_axis (bvh.rs:33-77) as level-by-level sorts: at level l every range of
// 256 objects or more is stably sorted by box centre along axis l mod 3 (NaN last, -0 = +0)
// and split in halves.  A range is a stretch of positions that depends on the object count
// alone, so all ranges of a level are sorted at once: a stable sort of every object by key,
// then a stable sort by the first position of its range.
namespace {

struct RangeAt {
  uint32_t lo, len, out_at;
};
// The range position p lies in after `levels` splits of [0, n), and where the reference's
// resolution order puts that range (the upper half of every split comes first).
__device__ __forceinline__ RangeAt range_of(uint32_t p, uint32_t n, uint32_t levels) {
  RangeAt r{0u, n, 0u};
  for (uint32_t l = 0; l < levels && r.len >= 256u; ++l) {  // bvh.rs:9: leaves hold fewer than 256
    const uint32_t half = r.len / 2u;
    if (p < r.lo + half) {
      r.out_at += r.len - half;
      r.len = half;
    } else {
      r.lo += half;
      r.len -= half;
    }
  }
  return r;
}

__global__ void __launch_bounds__(kThreads)
order_keys_kernel(const uint32_t* perm, const double* centre, uint32_t n, uint32_t level, uint64_t* keys,
                  uint64_t* vals) {
  const uint32_t p = blockIdx.x * kThreads + threadIdx.x;
  if (p >= n) return;
  const RangeAt r = range_of(p, n, level);
  const uint32_t id = perm[p];
  uint64_t key = 0;  // a range that is a leaf keeps its order
  if (r.len >= 256u) {
    double x = centre[id];
    if (x != x) {
      key = ~0ull;  // OrderedFloat: NaN is greatest, and all NaN are equal
    } else {
      if (x == 0.0) x = 0.0;  // -0 and +0 compare equal
      const uint64_t u = (uint64_t)__double_as_longlong(x);
      key = (u >> 63) ? ~u : (u | 0x8000000000000000ull);
    }
  }
  keys[p] = key;
  vals[p] = ((uint64_t)r.lo << 32) | id;
}

__global__ void __launch_bounds__(kThreads)
order_split_kernel(const uint64_t* vals, uint32_t n, uint32_t* range_lo, uint32_t* ids) {
  const uint32_t p = blockIdx.x * kThreads + threadIdx.x;
  if (p >= n) return;
  range_lo[p] = (uint32_t)(vals[p] >> 32);
  ids[p] = (uint32_t)vals[p];
}

__global__ void __launch_bounds__(kThreads)
order_emit_kernel(const uint32_t* perm, uint32_t n, uint32_t levels, uint32_t* order) {
  const uint32_t p = blockIdx.x * kThreads + threadIdx.x;
  if (p >= n) return;
  const RangeAt r = range_of(p, n, levels);
  order[r.out_at + (p - r.lo)] = perm[p];
}

__global__ void __launch_bounds__(kThreads)
iota_kernel(uint32_t* v, uint32_t n) {
  const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
  if (i < n) v[i] = i;
}

}  // namespace

std::string device_reference_order(hipStream_t stream, const std::vector<double>& centres, uint32_t n,
                                   std::vector<uint32_t>& order) {
  order.resize(n);
  if (n == 0) return "";
  if (centres.size() != (size_t)3 * n) return "device order: three centres per object expected";
  uint32_t levels = 0;
  for (uint32_t longest = n; longest >= 256u; longest -= longest / 2u) ++levels;
  uint32_t pos_bits = 1;
  while (pos_bits < 32 && (n >> pos_bits) != 0) ++pos_bits;
  Scratch mem;
  double* d_centres = nullptr;
  uint64_t *d_keys = nullptr, *d_keys2 = nullptr, *d_vals = nullptr, *d_vals2 = nullptr;
  uint32_t *d_perm = nullptr, *d_lo = nullptr, *d_lo2 = nullptr, *d_ids = nullptr, *d_order = nullptr;
  mem.plan(&d_centres, (size_t)3 * n);
  mem.plan(&d_keys, n);
  mem.plan(&d_keys2, n);
  mem.plan(&d_vals, n);
  mem.plan(&d_vals2, n);
  mem.plan(&d_perm, n);
  mem.plan(&d_lo, n);
  mem.plan(&d_lo2, n);
  mem.plan(&d_ids, n);
  mem.plan(&d_order, n);
  size_t bytes1 = 0, bytes2 = 0;
  LB_HIP(rocprim::radix_sort_pairs(nullptr, bytes1, d_keys, d_keys2, d_vals, d_vals2, n, 0, 64, stream));
  LB_HIP(rocprim::radix_sort_pairs(nullptr, bytes2, d_lo, d_lo2, d_ids, d_perm, n, 0, pos_bits, stream));
  unsigned char* d_tmp = nullptr;
  mem.plan(&d_tmp, std::max(bytes1, bytes2));
  LB_HIP(mem.allocate());
  LB_HIP(hipMemcpyAsync(d_centres, centres.data(), (size_t)3 * n * sizeof(double), hipMemcpyHostToDevice, stream));
  iota_kernel<<<blocks(n), kThreads, 0, stream>>>(d_perm, n);
  for (uint32_t l = 0; l < levels; ++l) {
    order_keys_kernel<<<blocks(n), kThreads, 0, stream>>>(d_perm, d_centres + (size_t)(l % 3u) * n, n, l, d_keys,
                                                          d_vals);
    LB_HIP(rocprim::radix_sort_pairs(d_tmp, bytes1, d_keys, d_keys2, d_vals, d_vals2, n, 0, 64, stream));
    order_split_kernel<<<blocks(n), kThreads, 0, stream>>>(d_vals2, n, d_lo, d_ids);
    LB_HIP(rocprim::radix_sort_pairs(d_tmp, bytes2, d_lo, d_lo2, d_ids, d_perm, n, 0, pos_bits, stream));
  }
  order_emit_kernel<<<blocks(n), kThreads, 0, stream>>>(d_perm, n, levels, d_order);
  LB_HIP(hipGetLastError());
  LB_HIP(hipMemcpyAsync(order.data(), d_order, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
  LB_HIP(hipStreamSynchronize(stream));
  return "";
}

}  // namespace rt
