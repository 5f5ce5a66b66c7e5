// The HIP render path and its C ABI (rt_* in include/mi355rt.h).
//
// Wavefront organisation of the reference's render closure (src/main.rs:82-101):
// a pass takes S samples of every pixel of the window as independent paths
// (path = s_local * n_pixels + pixel, so a wave holds neighbouring pixels of one
// sample).  Two pipelines produce the same numbers:
//
// fused (the default), per pass:
//   repeat max_depth times (ray_color's recursion, main.rs:120-140, unrolled):
//     trace    camera ray (first bounce: jitter + Camera::get_ray, main.rs:87-94,
//              camera.rs:22-36) or ray load, closest hit over the BVH, hit record,
//              emitted, scatter, attenuation; survivors go to the next live queue.
//              Hits on materials with procedural or image textures are queued instead.
//              Once few paths are left, one launch finishes all their bounces.
//     shade    (scenes with such textures) the queued hits, any material kind
//   resolve  per pixel, adds the S path radiances in sample order (main.rs:97)
//
// binned (RT_FLAG_BINNED), per pass:
//   raygen   jitter + Camera::get_ray
//   repeat max_depth times:
//     extend   closest hit, then each live path is appended to the bin of what it
//              hit: miss, or one bin per material kind
//     shade    one launch per bin, so a kernel carries one material's code:
//              emitted, scatter, attenuation; survivors go to the next live queue
//   resolve
//
// Queue appends are staged in LDS and reach the global queues in runs, one atomic
// per run.  All queue sizes live in device memory; the host looks at the live
// count every eighth bounce only to stop early and to shrink the grid.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <limits>
#include <memory>
#include <mutex>
#include <thread>
#include <dlfcn.h>
#include <type_traits>
#include <vector>

#include "bvh_device.h"
#include "rt_device.h"

using namespace rt;

// Path state, queues, counters and the f64 sums of rt_render_device_noalloc: one array per GPU,
// part of the code object, so the entry point never allocates.  Host code must not use this
// name as an address: rt_create asks the runtime for each device's copy (hipGetSymbolAddress)
// and everything below works on that pointer.
static __device__ __align__(256) unsigned char scratch[RT_NOALLOC_SCRATCH_BYTES];

namespace {

const int kBlock = 256;
const uint32_t kBins = 1 + RT_MAT_KINDS;  // miss + one per material kind
// The queue of live paths is kept as kSubQueues queues, each with its own counter in its own
// 128-byte line; a workgroup appends to the one its index selects.  Returning atomics on one
// line are served one after the other (about 90 per microsecond), and the fused kernel's waves
// together come close to that rate; spread over eight lines they no longer wait for each other.
// Counter row of one bounce (words): [0..5] bin sizes, [kAnyBin] deferred hits,
// [32 * (1 + q)] entries of sub-queue q of the next bounce's live queue.
const uint32_t kSubQueues = 8;
const uint32_t kCounterStride = 32 * (1 + kSubQueues);
const int kAnyBin = 7;                    // shade_kernel over the deferred queue: any material kind
// [kWorkWord] of a bounce's row (and of each of its sub-queue lines but the first): how many runs
// of trips the waves of its trace launch have drawn from a stretch of the queue
const uint32_t kWorkWord = 16;
#ifndef RT_TRIPS_PER_GRAB
#define RT_TRIPS_PER_GRAB 4
#endif
#ifndef RT_WORK_SLICES
#define RT_WORK_SLICES 8  // (at most 8: the words come from the sub-queue lines of a bounce's row)
#endif

struct PathBuffers {
  double4* ray_o;     // origin xyz, time
  double4* ray_d;     // direction xyz, hit t (written by extend)
  double4* beta;      // running attenuation product
  double4* radiance;  // running radiance
  uint32_t* hit_item; // leaf slot hit, or kNoPrim
  uint32_t* live[2];  // live-path queues, ping-pong; each kSubQueues regions of `capacity` entries
  uint32_t* bins;     // kBins queues of `capacity` entries (binned pipeline), or the one deferred queue
  uint32_t* counters; // (max_depth) x kCounterStride
  uint64_t capacity;
  // Streamed path state (the fused pipeline): ray and attenuation live at the path's POSITION IN
  // THE LIVE QUEUE, not at its index.  A bounce reads positions of the set
  // (bounce & 1) and writes its survivors to freshly reserved positions of the other set
  // (ray_o, ray_d, beta are set 0; alt_* set 1), so both sides are dense whatever share of
  // the pass is still alive.  Only sub-queue 0 is used; its counter is the end of the position space.
  double4* alt_o;
  double4* alt_d;
  double4* alt_beta;
  uint32_t stream;
};

// Division of a 32-bit number by an invariant divisor as one multiply-high and two shifts
// (Granlund and Montgomery, "Division by invariant integers using multiplication", fig. 4.1):
// q = (t + ((n - t) >> s1)) >> s2 with t = mulhi(m, n).  Exact for every n < 2^32.
struct FastDiv {
  uint32_t d, m, s1, s2;
};
inline FastDiv make_fastdiv(uint32_t d) {
  FastDiv f{};
  f.d = d;
  uint32_t l = 0;
  while (l < 32 && (1ull << l) < d) ++l;  // ceil(log2 d)
  f.m = (uint32_t)((((1ull << l) - d) << 32) / d + 1);
  f.s1 = l < 1 ? l : 1;
  f.s2 = l > 1 ? l - 1 : 0;
  return f;
}
__host__ __device__ inline uint32_t fastdiv(uint32_t n, const FastDiv& f) {
#if defined(__HIP_DEVICE_COMPILE__)
  uint32_t t = __umulhi(f.m, n);
#else
  uint32_t t = (uint32_t)(((uint64_t)f.m * n) >> 32);
#endif
  return (t + ((n - t) >> f.s1)) >> f.s2;
}

struct PassInfo {
  uint32_t width, height;      // full image
  uint32_t x0, y0, ww, wh;     // window
  uint32_t n_pixels;           // ww * wh
  FastDiv by_pixels, by_ww;    // division by n_pixels and by ww
  uint32_t n_paths;            // n_pixels * samples in this pass
  uint32_t sample_base;        // absolute index of the pass's first sample
  uint32_t first_pass;         // 1: resolve starts the sums
  uint32_t samples;            // samples per pixel in this pass
  uint32_t tiled;              // window pixels are numbered 8x8 tile by tile (by_ww divides by tiles per row)
  uint64_t seed;
  // Per 8x8 tile of the window, what its camera rays can meet (tile_list_kernel), or null: tile_words
  // words a tile, [0] the number of leaf slots that follow or kTileWalk; sphere-only scenes: then, from
  // word 16 on, centre and radius of each (so the first bounce reads one block and no record).
  const uint32_t* tile_lists;
  uint32_t tile_words;
};
const uint32_t kTileListWords = 16;          // header: count + up to 15 leaf slots
const uint32_t kTileSphereMax = 14;          // sphere-only scenes: at most this many, 32 bytes of data each
const uint32_t kTileSphereWords = 16 + 8 * kTileSphereMax;  // 128 words = 512 bytes a tile
const uint32_t kTileWalk = 0xFFFFFFFFu;

RT_DEV uint32_t lane_id() { return __lane_id(); }

// Pixel of the window's `local`-th path slot.  Which pixel a slot names changes no result (draws
// are addressed by pixel and sample): windows that divide into 8x8 tiles are numbered tile by
// tile, so that the 64 camera rays of a wave cover a square instead of a 64-pixel row segment
// and stay together longer in the hierarchy.
RT_DEV void pixel_of(const PassInfo& ps, uint32_t local, uint32_t& x, uint32_t& y) {
  if (ps.tiled) {
    const uint32_t tile = local >> 6, in = local & 63u;
    const uint32_t ty = fastdiv(tile, ps.by_ww), tx = tile - ty * (ps.ww >> 3);
    x = ps.x0 + (tx << 3) + (in & 7u);
    y = ps.y0 + (ty << 3) + (in >> 3);
  } else {
    const uint32_t row = fastdiv(local, ps.by_ww);
    x = ps.x0 + (local - row * ps.ww);
    y = ps.y0 + row;
  }
}

// Where workgroup `block` appends the paths that live on after `bounce`.
RT_DEV uint32_t* next_live_counter(const PathBuffers& pb, uint32_t bounce, uint32_t block) {
  return pb.counters + (size_t)bounce * kCounterStride + 32u * (1u + (block % kSubQueues));
}
RT_DEV uint32_t* next_live_queue(const PathBuffers& pb, uint32_t bounce, uint32_t block) {
  return pb.live[(bounce + 1u) & 1u] + (size_t)(block % kSubQueues) * pb.capacity;
}
// The live paths of `bounce` as one index range over the sub-queues the previous bounce
// filled (bounce 0: every path of the pass, in order).
struct LiveIn {
  uint32_t start[kSubQueues + 1];  // start[q] = entries in the sub-queues before q
  const uint32_t* queue;
  uint64_t capacity;
  uint32_t identity;
  RT_DEV uint32_t total() const { return start[kSubQueues]; }
  RT_DEV uint32_t path(uint32_t i) const {
    if (identity) return i;
    uint32_t q = 0;
#pragma unroll
    for (uint32_t k = 1; k < kSubQueues; ++k) q += i >= start[k] ? 1u : 0u;
    uint32_t first = start[0];
#pragma unroll
    for (uint32_t k = 1; k < kSubQueues; ++k) first = q >= k ? start[k] : first;
    return queue[(size_t)q * capacity + (i - first)];
  }
};
RT_DEV LiveIn live_in(const PathBuffers& pb, uint32_t bounce, uint32_t n_paths) {
  LiveIn li;
  li.queue = pb.live[bounce & 1u];
  li.capacity = pb.capacity;
  li.identity = bounce == 0 ? 1u : 0u;
  li.start[0] = 0;
  const uint32_t* row = pb.counters + (size_t)(bounce == 0 ? 0u : bounce - 1u) * kCounterStride;
#pragma unroll
  for (uint32_t q = 0; q < kSubQueues; ++q)
    li.start[q + 1] = bounce == 0 ? n_paths : li.start[q] + row[32u * (1u + q)];
  return li;
}

// The same live paths as trips of 64 for trace_kernel: the index space gives every sub-queue a
// whole number of trips (its count rounded up to 64), so a trip lies in one sub-queue and which
// one is settled with scalar arithmetic on the trip's first index; the lanes past the end of a
// sub-queue's last trip get no path.  (Finding the sub-queue per lane, through eight compares
// and selects against prefix sums, was forty vector instructions of every trip.)
const uint32_t kNoPath = 0xFFFFFFFFu;  // (a pass holds at most 2^30 paths)
struct LiveTrips {
  uint32_t count[kSubQueues];  // entries per sub-queue (bounce 0: unused)
  const uint32_t* queue;
  uint64_t capacity;
  uint32_t n_paths;   // bounce 0: every path of the pass, in order
  uint32_t end;       // end of the trip index space
  // Path of lane `lane` in the trip that starts at index `base` (wave-uniform, a multiple of 64), or kNoPath.
  template <bool IDENTITY>
  RT_DEV uint32_t fetch(uint32_t base, uint32_t lane) const {
    if (IDENTITY) {
      const uint32_t i = base + lane;
      return i < n_paths ? i : kNoPath;
    }
    uint32_t off = base, q = 0;
#pragma unroll
    for (uint32_t k = 0; k + 1 < kSubQueues; ++k) {
      const uint32_t padded = (count[k] + 63u) & ~63u;
      const bool past = q == k && off >= padded;
      off = past ? off - padded : off;
      q = past ? k + 1u : q;
    }
    uint32_t cnt = count[0];
#pragma unroll
    for (uint32_t k = 1; k < kSubQueues; ++k) cnt = q == k ? count[k] : cnt;
    const uint32_t j = off + lane;
    return j < cnt ? queue[(size_t)q * capacity + j] : kNoPath;
  }
};
RT_DEV LiveTrips live_trips(const PathBuffers& pb, uint32_t bounce, uint32_t n_paths) {
  LiveTrips lt;
  lt.queue = pb.live[bounce & 1u];
  lt.capacity = pb.capacity;
  lt.n_paths = n_paths;
  const uint32_t* row = pb.counters + (size_t)(bounce == 0 ? 0u : bounce - 1u) * kCounterStride;
  uint32_t end = 0;
#pragma unroll
  for (uint32_t q = 0; q < kSubQueues; ++q) {
    lt.count[q] = bounce == 0 ? 0u : row[32u * (1u + q)];
    end += (lt.count[q] + 63u) & ~63u;
  }
  lt.end = bounce == 0 ? ((n_paths + 63u) & ~63u) : end;
  return lt;
}

// Queue appends are staged in LDS.  A returning atomic on one global word
// saturates near 90 per microsecond on this chip, far below one per wave per
// bin, so a workgroup collects entries in an LDS buffer (slots handed out by an
// LDS atomic per wave: ballot + popcount) and moves them to the global queue in
// runs of kFlushAt or more, with ONE global atomic per run and coalesced stores.
const uint32_t kFlushAt = 256;               // flush a staged queue at this many entries
const uint32_t kStageCap = kFlushAt + kBlock;  // a round adds at most kBlock entries

// Slot in the LDS stage for each active lane.  Must be reached by whole waves.
RT_DEV uint32_t stage_append(uint32_t* lds_count, bool active) {
  unsigned long long mask = __ballot(active);
  uint32_t base = 0;
  if (mask != 0ull) {
    uint32_t lane = lane_id();
    uint32_t leader = (uint32_t)__ffsll((long long)mask) - 1u;
    if (lane == leader) base = atomicAdd(lds_count, (uint32_t)__popcll(mask));
    base = __shfl(base, (int)leader);
    base += (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
  }
  return base;
}

// Move `n` staged entries to the global queue.  Called by the whole workgroup
// with a uniform n; `lds_word` is scratch for the broadcast of the reserved base.
RT_DEV void stage_flush(const uint32_t* lds_buf, uint32_t* lds_count, uint32_t* lds_word,
                        uint32_t* g_counter, uint32_t* g_queue, uint32_t n) {
  if (threadIdx.x == 0) *lds_word = atomicAdd(g_counter, n);
  __syncthreads();  // every thread has read its copy of n before this point
  const uint32_t base = *lds_word;
  if (threadIdx.x == 0) *lds_count = 0;
  for (uint32_t i = threadIdx.x; i < n; i += kBlock) g_queue[base + i] = lds_buf[i];
  __syncthreads();
}

#ifndef RT_STREAM_MIN_CHUNK
#define RT_STREAM_MIN_CHUNK 128
#endif
const uint32_t kStreamMinChunk = RT_STREAM_MIN_CHUNK, kStreamMaxChunk = 1024;  // (powers of two, multiples of 64)
// 2: a wave of the fused kernel keeps two runs of output positions, one for survivors whose new ray
// points up (before its random part is added) and one for the others, so that the trips of the
// next bounce hold rays of one kind (A/B builds: measured slower, a trip's survivors split over
// two runs write half-filled lines and the walk gains nothing from the class).
#ifndef RT_STREAM_CLASSES
#define RT_STREAM_CLASSES 1
#endif
const uint32_t kStreamClasses = RT_STREAM_CLASSES;
// Output positions of one wave in a streamed bounce (PathBuffers::stream).  The wave reserves a
// chunk of positions at a time with one atomic and hands them to its surviving lanes call by call;
// finish() marks what is left of its last chunk empty.  Every call is made by the whole wave.
struct StreamOut {
  uint32_t chunk, next, end;  // wave-uniform: chunk size; the unused part of the current chunk
  uint32_t* counter;          // end of the position space
  uint32_t* ids;              // path index per position, kNoPath where empty
  RT_DEV void init(uint32_t chunk_, uint32_t* counter_, uint32_t* ids_) {
    chunk = chunk_;
    next = end = 0;
    counter = counter_;
    ids = ids_;
  }
  RT_DEV uint32_t take(bool survives) {  // the position of a surviving lane
    const unsigned long long m = __ballot(survives);
    const uint32_t cnt = (uint32_t)__popcll(m);
    uint32_t pos = 0;
    if (cnt != 0u) {
      const uint32_t rank = (uint32_t)__popcll(m & ((1ull << lane_id()) - 1ull)), room = end - next;
      if (cnt <= room) {
        pos = next + rank;
        next += cnt;
      } else {  // (a chunk holds a trip: one fresh chunk is enough)
        uint32_t v = 0;
        if (lane_id() == 0) v = atomicAdd(counter, chunk);
        const uint32_t fresh = (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
        pos = rank < room ? next + rank : fresh + (rank - room);
        next = fresh + (cnt - room);
        end = fresh + chunk;
      }
    }
    return pos;
  }
  RT_DEV void finish() {
    for (uint32_t i = next + lane_id(); i < end; i += 64u) ids[i] = kNoPath;
  }
};

#ifdef RT_LANE_PROFILE
__device__ unsigned long long g_lane_trips[2 * LP_COUNT];
#endif
RT_DEV void flush_stats(const Stats& st, unsigned long long* g) {
#pragma unroll
  for (int i = 0; i < RT_STAT_COUNT; ++i)
    if (st.v[i]) atomicAdd(g + i, (unsigned long long)st.v[i]);
#ifdef RT_LANE_PROFILE
#pragma unroll
  for (int i = 0; i < 2 * LP_COUNT; ++i)
    if (st.trips[i]) atomicAdd(g_lane_trips + i, (unsigned long long)st.trips[i]);
#endif
}

// --------------------------------------------------------------- raygen
// The camera ray of one path: jitter (main.rs:91-93) and Camera::get_ray (camera.rs:22-36).
RT_DEV void camera_ray(const DeviceScene& sc, const PassInfo& ps, uint32_t x, uint32_t y,
                       const DrawKey& key, V3& origin, V3& dir, double& time) {
  u32x4 j = draw(key, DRAW_JITTER, 0);
  double jx = u01_53(j.x, j.y), jy = u01_53(j.z, j.w);
  double u = ((double)x + jx) / ((double)ps.width - 1.0);
  double v = ((double)(ps.height - y) + jy) / ((double)ps.height - 1.0);
  const rt_camera& c = sc.cam;
  V3 rd = vscale(rand_in_unit_disk(key), c.lens_radius);
  V3 offset = vadd(vscale(v3(c.u), rd.x), vscale(v3(c.v), rd.y));
  if (c.time1 > c.time0) {
    u32x4 tb = draw(key, DRAW_TIME, 0);
    time = c.time0 + (c.time1 - c.time0) * u01_53(tb.x, tb.y);
  } else {
    time = c.time0;
  }
  origin = vadd(v3(c.origin), offset);
  dir = vsub(vsub(vadd(vadd(v3(c.lower_left), vscale(v3(c.horizontal), u)),
                       vscale(v3(c.vertical), v)),
                  v3(c.origin)),
             offset);
}

__global__ void __launch_bounds__(kBlock)
raygen_kernel(DeviceScene sc, PathBuffers pb, PassInfo ps) {
  uint32_t path = blockIdx.x * kBlock + threadIdx.x;
  if (path >= ps.n_paths) return;
  uint32_t s_local = fastdiv(path, ps.by_pixels), local = path - s_local * ps.n_pixels;
  uint32_t x, y;
  pixel_of(ps, local, x, y);
  DrawKey key{ps.seed, y * ps.width + x, ps.sample_base + s_local, 0u};
  V3 origin, dir;
  double time;
  camera_ray(sc, ps, x, y, key, origin, dir, time);
  pb.ray_o[path] = make_double4(origin.x, origin.y, origin.z, time);
  pb.ray_d[path] = make_double4(dir.x, dir.y, dir.z, 0.0);
  pb.beta[path] = make_double4(1.0, 1.0, 1.0, 0.0);
  pb.radiance[path] = make_double4(0.0, 0.0, 0.0, 0.0);
}

// --------------------------------------------------------------- extend
#ifndef RT_EXTEND_MEDIA_WAVES
#define RT_EXTEND_MEDIA_WAVES 3
#endif
template <bool STATS, bool MEDIA, bool OVERFLOW>
__global__ void __launch_bounds__(kBlock, STATS ? 1 : MEDIA ? RT_EXTEND_MEDIA_WAVES : 4)  // waves per SIMD to fit registers for
extend_kernel(DeviceScene sc, LdsLayout lay, PathBuffers pb, PassInfo ps, uint32_t bounce,
              unsigned long long* g_stats) {
  extern __shared__ __align__(16) char smem[];
  __shared__ uint32_t s_bin[kBins * kStageCap];
  __shared__ uint32_t s_bin_count[kBins];
  __shared__ uint32_t s_word;
  const LiveIn live = live_in(pb, bounce, ps.n_paths);
  uint32_t* counters = pb.counters + (size_t)bounce * kCounterStride;
  const uint32_t n_live = live.total();
  Stats st;
  if (STATS) st.clear();
  const uint32_t stride = gridDim.x * kBlock;
  if (blockIdx.x * kBlock >= n_live) return;  // nothing for this workgroup: skip the staging
  if (threadIdx.x < kBins) s_bin_count[threadIdx.x] = 0;
  const LdsScene lds = lds_stage(sc, lay, smem, threadIdx.x, kBlock);  // ends with a barrier
  sc.ops_lds_offset = lds.ops_offset;
  sc.n_ops_lds = lay.n_ops;
  // whole workgroups iterate together so that every lane reaches the ballots and barriers
  for (uint32_t base = blockIdx.x * kBlock; base < n_live; base += stride) {
    uint32_t i = base + threadIdx.x;
    bool valid = i < n_live;
    uint32_t path = 0, bin = 0;
    if (valid) {
      path = live.path(i);
      double4 ro = pb.ray_o[path];
      double4 rdv = pb.ray_d[path];
      uint32_t s_local = fastdiv(path, ps.by_pixels), local = path - s_local * ps.n_pixels;
      uint32_t x, y;
      pixel_of(ps, local, x, y);
      DrawKey key{ps.seed, y * ps.width + x, ps.sample_base + s_local, bounce};
      stat<STATS>(st, RT_STAT_RAYS);
      Closest h = closest_hit<STATS, MEDIA, MEDIA, OVERFLOW>(sc, lds, v3(ro.x, ro.y, ro.z), v3(rdv.x, rdv.y, rdv.z),
                                            ro.w, 0.001, key, st);
      pb.hit_item[path] = h.item;
      if (h.item != kNoPrim) {
        pb.ray_d[path].w = h.t;
        bin = 1u + ((h.kind >> 8) & 7u);
        stat<STATS>(st, RT_STAT_HITS);
      }
    }
#pragma unroll
    for (uint32_t k = 0; k < kBins; ++k) {
      bool mine = valid && bin == k;
      uint32_t slot = stage_append(&s_bin_count[k], mine);
      if (mine) s_bin[k * kStageCap + slot] = path;
    }
    __syncthreads();
    // every wave reads the same counts: none may append again before all have read them
    uint32_t staged[kBins];
#pragma unroll
    for (uint32_t k = 0; k < kBins; ++k) staged[k] = s_bin_count[k];
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < kBins; ++k) {
      if (staged[k] >= kFlushAt)
        stage_flush(s_bin + k * kStageCap, &s_bin_count[k], &s_word, counters + k,
                    pb.bins + (size_t)k * pb.capacity, staged[k]);
    }
  }
#pragma unroll
  for (uint32_t k = 0; k < kBins; ++k) {
    uint32_t n = s_bin_count[k];
    if (n) stage_flush(s_bin + k * kStageCap, &s_bin_count[k], &s_word, counters + k,
                       pb.bins + (size_t)k * pb.capacity, n);
  }
  if (STATS) flush_stats(st, g_stats);
}

// ---------------------------------------------------------------- shade
// What one hit contributes, after Material::scatter / Material::emitted of the
// material kind `kind` (src/material/*.rs, src/texture/isotropic.rs).
struct ShadeOut {
  V3 attenuation, direction, emitted, p;
  V3 alt;           // dielectric whose choice is still open: the refracted direction (`direction`: the mirrored one)
  double fuzz;      // metal: scale of the random vector; open dielectric: reflect iff fuzz > the uniform
  uint32_t kind;
  bool scattered, emits;
  bool needs_vector;  // direction still lacks its random unit vector (shade_finish)
  bool needs_u;       // dielectric: the reflect-or-refract uniform is still to be drawn (shade_finish)
};

// First half: everything up to the material's random unit vector.  For Lambertian,
// Metal and Isotropic hits `direction` holds what the vector is added to.
// Material index of a record (`pr` must have been loaded whole when the kind word does not hold it).
RT_DEV uint32_t material_of(const DPrim& pr) {
  return (pr.kind & kMaterialElsewhereBit) ? pr.material : pr.kind >> kMaterialShift;
}

// SOLID: the material's texture (if it has one) is known to be a SolidColor, whose colour the
// material record carries: no texture walk, no (u, v).
// SPHERES: `pr` is a plain sphere without wrappers.
// LATER: a Dielectric hit leaves its uniform draw to the caller (needs_u; both directions are
// prepared and shade_finish picks), so that the draw can share a generator block's trip with the
// wave's random vectors.
template <bool STATS, bool SOLID, bool SPHERES = false, bool LATER = false>
RT_DEV ShadeOut shade_begin(const DeviceScene& sc, uint32_t kind, const DPrim& pr, uint32_t material,
                            V3 o, V3 d, double time, double t, const DrawKey& key, Stats& st) {
  ShadeOut out;
  out.attenuation = v3(0.0, 0.0, 0.0);
  out.direction = v3(0.0, 0.0, 0.0);
  out.emitted = v3(0.0, 0.0, 0.0);
  out.fuzz = 0.0;
  out.kind = kind;
  out.scattered = false;
  out.emits = false;
  out.needs_vector = false;
  out.needs_u = false;
  out.alt = v3(0.0, 0.0, 0.0);
  stat<STATS>(st, RT_STAT_SHADE_LAMBERTIAN + (int)kind);
  const DMaterial m = sc.materials[material];
  const bool textured = kind == RT_MAT_LAMBERTIAN || kind == RT_MAT_DIFFUSE_LIGHT ||
                        kind == RT_MAT_ISOTROPIC;
  bool want_uv = textured && ((pr.kind >> 11) & 1u);
  Record rec = make_record<!SOLID, SPHERES>(sc, pr, o, d, time, t, want_uv);
  out.p = rec.p;
  // Metal and Dielectric both start from the unit direction and its mirror image; a wave that
  // holds both kinds computes them in one go instead of once per branch.
  V3 ud = v3(0.0, 0.0, 0.0), mirrored = ud;
  if (kind == RT_MAT_METAL || kind == RT_MAT_DIELECTRIC) {
    ud = vunit(d);
    mirrored = vreflect(ud, rec.normal);
  }
  if (kind == RT_MAT_LAMBERTIAN) {  // lambertian.rs:13-25
    out.direction = rec.normal;
    out.attenuation = SOLID ? v3(m.albedo) : texture_value<STATS>(sc, m.texture, rec.u, rec.v, rec.p, st);
    out.scattered = true;
    out.needs_vector = true;
  } else if (kind == RT_MAT_METAL) {  // metal.rs:12-28
    if (vdot(mirrored, rec.normal) > 0.0) {
      out.direction = mirrored;
      out.fuzz = m.param;
      out.scattered = true;
      out.needs_vector = true;
    }
    out.attenuation = v3(m.albedo);
  } else if (kind == RT_MAT_DIELECTRIC) {  // dielectric.rs:14-38
    double ratio = rec.front ? m.albedo[0] : m.param;  // albedo[0] of a Dielectric: 1.0 / param
    double cos_theta = fmin(vdot(vneg(ud), rec.normal), 1.0);
    double sin_theta = sqrt(1.0 - cos_theta * cos_theta);
    bool cannot_refract = ratio * sin_theta > 1.0;
    if (LATER) {
      out.direction = mirrored;
      out.alt = vrefract(ud, rec.normal, ratio);
      out.fuzz = cannot_refract ? 2.0 : reflectance(cos_theta, ratio);  // (2 > every uniform: reflect)
      out.needs_u = true;
    } else {
      u32x4 fb = draw(key, DRAW_FRESNEL, 0);
      bool should_reflect = reflectance(cos_theta, ratio) > u01_53(fb.x, fb.y);
      out.direction = (cannot_refract || should_reflect) ? mirrored : vrefract(ud, rec.normal, ratio);
    }
    out.attenuation = v3(1.0, 1.0, 1.0);
    out.scattered = true;
  } else if (kind == RT_MAT_DIFFUSE_LIGHT) {  // emissive.rs:10-12; never scatters
    out.emitted = SOLID ? v3(m.albedo) : texture_value<STATS>(sc, m.texture, rec.u, rec.v, rec.p, st);
    out.emits = true;
  } else {  // isotropic.rs:10-17
    out.attenuation = SOLID ? v3(m.albedo) : texture_value<STATS>(sc, m.texture, rec.u, rec.v, rec.p, st);
    out.scattered = true;
    out.needs_vector = true;
  }
  return out;
}

// Second half: `inside` is the accepted candidate of rand_unit_vector's rejection loop, `u` the
// uniform of an open Dielectric choice.
RT_DEV void shade_finish(ShadeOut& out, V3 inside, double u = 0.0) {
  if (out.needs_u && !(out.fuzz > u)) out.direction = out.alt;
  if (!out.needs_vector) return;
  const V3 ruv = vunit(inside);
  if (out.kind == RT_MAT_LAMBERTIAN) {
    V3 dir = vadd(out.direction, ruv);
    if (vnear_zero(dir)) dir = out.direction;
    out.direction = dir;
  } else if (out.kind == RT_MAT_METAL) {
    out.direction = vadd(out.direction, vscale(ruv, out.fuzz));
  } else {
    out.direction = ruv;
  }
}

// One instantiation per bin (BIN 0 = misses, BIN 1 + k = material kind k), each
// launched over its own queue: a kernel carries only its material's code and
// registers.  SOLID: every texture of the scene is a SolidColor (no texture walk,
// no (u, v)).  BIN kAnyBin: the queue of hits trace_kernel deferred (materials with
// procedural or image textures, whatever their kind).
template <int BIN, bool STATS, bool SOLID>
__global__ void __launch_bounds__(kBlock)
shade_kernel(DeviceScene sc, PathBuffers pb, PassInfo ps, uint32_t bounce, uint32_t max_depth,
             unsigned long long* g_stats) {
  __shared__ uint32_t s_next[kStageCap];
  __shared__ uint32_t s_next_count;
  __shared__ uint32_t s_word;
  __shared__ uint32_t s_xch[(kBlock / 64) * 128];
  lds_u32* xch = (lds_u32*)s_xch + (threadIdx.x / 64u) * 128u;
  // material kinds whose scatter draws a random unit vector
  constexpr bool kVector = BIN == 1 + RT_MAT_LAMBERTIAN || BIN == 1 + RT_MAT_METAL ||
                           BIN == 1 + RT_MAT_ISOTROPIC || BIN == kAnyBin;
  uint32_t* counters = pb.counters + (size_t)bounce * kCounterStride;
  // Streamed state (only ever with BIN == kAnyBin): a queue entry is the position the fused kernel
  // read the path at; ray and attenuation are there, in the set of this bounce, and the survivors
  // go to fresh positions of the other set like the fused kernel's.
  const bool stream = BIN == kAnyBin && pb.stream != 0u;
  uint32_t* next_live = next_live_queue(pb, bounce, stream ? 0u : blockIdx.x);
  uint32_t* next_count = next_live_counter(pb, bounce, stream ? 0u : blockIdx.x);
  const bool in_alt = stream && (bounce & 1u);
  const double4* const in_o = in_alt ? pb.alt_o : pb.ray_o;
  const double4* const in_d = in_alt ? pb.alt_d : pb.ray_d;
  const double4* const in_beta = in_alt ? pb.alt_beta : pb.beta;
  double4* const out_o = stream && !in_alt ? pb.alt_o : pb.ray_o;
  double4* const out_d = stream && !in_alt ? pb.alt_d : pb.ray_d;
  double4* const out_beta = stream && !in_alt ? pb.alt_beta : pb.beta;
  const uint32_t* const live_in = pb.live[bounce & 1u];
  const uint32_t total = counters[BIN];
  if (blockIdx.x * kBlock >= total) return;
  const uint32_t* queue = pb.bins + (BIN == kAnyBin ? (size_t)0 : (size_t)BIN * pb.capacity);
  if (threadIdx.x == 0) s_next_count = 0;
  __syncthreads();
  StreamOut appends;
  appends.init(kStreamMinChunk, next_count, next_live);
  Stats st;
  if (STATS) st.clear();
  const uint32_t stride = gridDim.x * kBlock;
  for (uint32_t base = blockIdx.x * kBlock; base < total; base += stride) {
    uint32_t i = base + threadIdx.x;
    bool valid = i < total;
    bool survives = false;
    uint32_t path = 0, at = 0;  // `at`: where the path's ray and attenuation are (its index, or its queue position)
    DrawKey key{ps.seed, 0u, 0u, bounce};
    ShadeOut so;
    so.needs_vector = false;
    so.needs_u = false;
    V3 nb = v3(0.0, 0.0, 0.0);  // a survivor's new attenuation, its (pixel, sample) word and its time
    double tag = 0.0, time = 0.0;
    if (valid) {
      at = queue[i];
      path = stream && bounce != 0u ? live_in[at] : at;
      double4 bt = in_beta[at];
      V3 beta = v3(bt.x, bt.y, bt.z);
      tag = bt.w;
      if (BIN == 0) {
        // miss: the recursion returns `background` (main.rs:137-139)
        stat<STATS>(st, RT_STAT_SHADE_MISS);
        double4 L = pb.radiance[path];
        V3 add = vmul(beta, v3(sc.bg));
        pb.radiance[path] = make_double4(L.x + add.x, L.y + add.y, L.z + add.z, 0.0);
      } else {
        double4 ro = in_o[at];
        double4 rdv = in_d[at];
        time = ro.w;
        const DPrim pr = sc.prims[pb.hit_item[at]];
        uint32_t s_local = fastdiv(path, ps.by_pixels), local = path - s_local * ps.n_pixels;
        uint32_t x, y;
        pixel_of(ps, local, x, y);
        key.pixel = y * ps.width + x;
        key.sample = ps.sample_base + s_local;
        const uint32_t kind = BIN == kAnyBin ? (pr.kind >> 8) & 7u : (uint32_t)BIN - 1u;
        so = shade_begin<STATS, SOLID>(sc, kind, pr, material_of(pr), v3(ro.x, ro.y, ro.z),
                                       v3(rdv.x, rdv.y, rdv.z), ro.w, rdv.w, key, st);
        if (so.emits) {
          double4 L = pb.radiance[path];
          V3 add = vmul(beta, so.emitted);
          pb.radiance[path] = make_double4(L.x + add.x, L.y + add.y, L.z + add.z, 0.0);
        }
        // depth budget: the scattered ray of the last allowed hit returns black (main.rs:121-122)
        survives = so.scattered && bounce + 1u < max_depth;
        if (survives) {
          nb = vmul(beta, so.attenuation);
        } else if (BIN == kAnyBin && !sc.has_lights) {
          // the fused kernel leaves L unwritten until a path ends when nothing emits
          pb.radiance[path] = make_double4(0.0, 0.0, 0.0, 0.0);
        }
      }
    }
    uint32_t to = at;  // where the survivor's state goes: in place, or a fresh position of the other set
    if (stream) to = appends.take(survives);
    if (survives) {
      // everything but the direction is known before the random vector
      out_beta[to] = make_double4(nb.x, nb.y, nb.z, tag);  // (.w: the path's pixel and sample, see trace_kernel)
      out_o[to] = make_double4(so.p.x, so.p.y, so.p.z, time);
      if (stream) next_live[to] = path;
    }
    if (kVector) {  // whole waves: the lanes look for their random vectors together
      V3 inside = wave_rand_in_unit_sphere(survives && so.needs_vector, key, xch, 64u, st);
      shade_finish(so, inside);
    }
    if (survives)
      out_d[to] = make_double4(so.direction.x, so.direction.y, so.direction.z, 0.0);
    if (!stream && BIN != 0 && BIN != 1 + RT_MAT_DIFFUSE_LIGHT) {  // misses and lights end their paths
      uint32_t slot = stage_append(&s_next_count, survives);
      if (survives) s_next[slot] = path;
      __syncthreads();
      uint32_t n = s_next_count;
      __syncthreads();  // (no wave appends again before all have read n)
      if (n >= kFlushAt) stage_flush(s_next, &s_next_count, &s_word, next_count, next_live, n);
    }
  }
  if (!stream && BIN != 0 && BIN != 1 + RT_MAT_DIFFUSE_LIGHT) {
    uint32_t n = s_next_count;
    if (n) stage_flush(s_next, &s_next_count, &s_word, next_count, next_live, n);
  }
  if (stream) appends.finish();
  if (STATS) flush_stats(st, g_stats);
}

// ------------------------------------------------------- camera-ray tile lists
// The camera rays of one 8x8 tile of pixels, over every jitter, lens and time sample, stay inside
// a thin beam; which leaf slots that beam can reach does not depend on the sample.  One thread per
// tile walks the hierarchy once per render call with a conservative beam-against-box test and writes
// the (few) leaf slots down; the first bounce of every one of the call's samples then tests its
// tile's slots, wave-uniform, instead of walking the hierarchy per ray.  Closest hit over a
// superset of what a ray can meet, by (t, rank): the same answer as the walk's.
//
// The beam.  Camera::get_ray (camera.rs:22-36) gives X(t) = O' + t (F - O') with O' = O + e,
// e = u rd.x R + v rd.y R, |rd| < 1, and F = lower_left + s horizontal + t' vertical on the tile's
// (s, t') rectangle (main.rs:91-93 with the jitter in [0, 1)).  With g = F - O:
//   X - O = (1 - t) e + t g.
// a = unit(u x v) is perpendicular to every e, so depth a.(X - O) = t (a.g); g ranges over the
// tile's parallelogram, a.g between its corner values.  A side plane k of the pyramid from O
// through the parallelogram (unit inward normal n_k, n_k.g >= 0 on the tile) sees
//   n_k.(X - O) >= -|1 - t| |e| >= -|1 - t| R',   R' = R sqrt(largest eigenvalue of [[u.u u.v][u.v v.v]]).
// A box is out of reach if it lies behind the lens plane, or if for some k all of it is further
// outside plane k than R' max |1 - t| over the t its depth range allows.  Every quantity carries a
// relative pad of 1e-9 against rounding; anything degenerate sends the tile to the walk.
struct TileBeam {
  double o[3], a[3], n[4][3];
  double g_min, g_max, r_lens;
  bool ok;
};
RT_DEV TileBeam tile_beam(const rt_camera& c, const PassInfo& ps, uint32_t tile) {
  TileBeam b;
  b.ok = false;
  const uint32_t per_row = ps.ww >> 3;
  const uint32_t ty = tile / per_row, tx = tile - ty * per_row;
  const double x_lo = (double)(ps.x0 + 8u * tx), y_lo = (double)(ps.y0 + 8u * ty);
  const double pad = 1e-9;
  const double s0 = x_lo / ((double)ps.width - 1.0) - pad, s1 = (x_lo + 8.0) / ((double)ps.width - 1.0) + pad;
  const double t0 = ((double)ps.height - (y_lo + 7.0)) / ((double)ps.height - 1.0) - pad;
  const double t1 = ((double)ps.height - y_lo + 1.0) / ((double)ps.height - 1.0) + pad;
  V3 o = v3(c.origin), ll = v3(c.lower_left), hz = v3(c.horizontal), vt = v3(c.vertical), u = v3(c.u), v = v3(c.v);
  V3 a = v3(u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x);
  const double la = vlen(a);
  if (!(la > 0.0)) return b;
  a = vscale(a, 1.0 / la);
  // corners of the tile on the focal plane, relative to O, in order around the parallelogram
  const double ss[4] = {s0, s1, s1, s0}, tt[4] = {t0, t0, t1, t1};
  V3 g[4];
  double gmin = 0.0, gmax = 0.0;
  for (int k = 0; k < 4; ++k) {
    g[k] = vsub(vadd(vadd(ll, vscale(hz, ss[k])), vscale(vt, tt[k])), o);
    const double d = vdot(a, g[k]);
    gmin = k == 0 ? d : fmin(gmin, d);
    gmax = k == 0 ? d : fmax(gmax, d);
  }
  if (gmax < 0.0) {  // (u x v points backwards: turn it round)
    a = vneg(a);
    const double lo = -gmax, hi = -gmin;
    gmin = lo;
    gmax = hi;
  }
  if (!(gmin > 0.0) || !(gmax < 1e300)) return b;
  const V3 centre = vadd(vadd(g[0], g[1]), vadd(g[2], g[3]));
  for (int k = 0; k < 4; ++k) {
    const V3 p = g[k], q = g[(k + 1) & 3];
    V3 n = v3(p.y * q.z - p.z * q.y, p.z * q.x - p.x * q.z, p.x * q.y - p.y * q.x);
    const double ln = vlen(n);
    if (!(ln > 0.0)) return b;
    n = vscale(n, 1.0 / ln);
    if (vdot(n, centre) < 0.0) n = vneg(n);
    // every corner on the inner side (two of them on the plane, up to rounding)
    for (int j = 0; j < 4; ++j)
      if (!(vdot(n, g[j]) >= -1e-9 * vlen(g[j]))) return b;
    b.n[k][0] = n.x;
    b.n[k][1] = n.y;
    b.n[k][2] = n.z;
  }
  const double uu = vdot(u, u), vv = vdot(v, v), uv = vdot(u, v);
  const double half = 0.5 * (uu - vv);
  const double eig = 0.5 * (uu + vv) + sqrt(half * half + uv * uv);
  b.r_lens = fabs(c.lens_radius) * sqrt(eig) * (1.0 + 1e-9);
  if (!(b.r_lens < 1e300)) return b;
  b.o[0] = o.x; b.o[1] = o.y; b.o[2] = o.z;
  b.a[0] = a.x; b.a[1] = a.y; b.a[2] = a.z;
  b.g_min = gmin * (1.0 - 1e-9);
  b.g_max = gmax * (1.0 + 1e-9);
  b.ok = true;
  return b;
}
// Can no camera ray of the beam reach the box?  (NaN anywhere, e.g. from an infinite box: false.)
RT_DEV bool beam_misses_box(const TileBeam& b, const float* lo, const float* hi) {
  double dmin = 0.0, dmax = 0.0;
  for (int k = 0; k < 3; ++k) {
    const double p = b.a[k] * ((double)lo[k] - b.o[k]), q = b.a[k] * ((double)hi[k] - b.o[k]);
    dmin += fmin(p, q);
    dmax += fmax(p, q);
  }
  const double slack = 1e-9 * (fabs(dmin) + fabs(dmax));
  if (dmax + slack < 0.0) return true;  // behind the lens plane: t > 0 means positive depth
  const double t_lo = fmax(dmin - slack, 0.0) / b.g_max, t_hi = (dmax + slack) / b.g_min;
  const double margin = b.r_lens * fmax(fabs(1.0 - t_lo), fabs(1.0 - t_hi));
  for (int k = 0; k < 4; ++k) {
    double reach = 0.0, size = 0.0;
    for (int j = 0; j < 3; ++j) {
      const double p = b.n[k][j] * ((double)lo[j] - b.o[j]), q = b.n[k][j] * ((double)hi[j] - b.o[j]);
      reach += fmax(p, q);
      size += fmax(fabs(p), fabs(q));
    }
    if (reach < -(margin * (1.0 + 1e-9) + 1e-9 * size)) return true;
  }
  return false;
}
template <bool SPHERES>
__global__ void __launch_bounds__(kBlock)
tile_list_kernel(DeviceScene sc, PassInfo ps, uint32_t* lists, uint32_t n_tiles) {
  const uint32_t tile = blockIdx.x * kBlock + threadIdx.x;
  if (tile >= n_tiles) return;
  constexpr uint32_t kMax = SPHERES ? kTileSphereMax : kTileListWords - 1u;
  uint32_t* out = lists + (size_t)tile * (SPHERES ? kTileSphereWords : kTileListWords);
  const TileBeam beam = tile_beam(sc.cam, ps, tile);
  if (!beam.ok) {
    out[0] = kTileWalk;
    return;
  }
  uint32_t count = 0;
  if (sc.n_nodes != 0) {
    const rt_bvh_node* nodes = (const rt_bvh_node*)sc.nodes;
    uint32_t stack[64];
    uint32_t sp = 0, cur = sc.root_code;  // (the root's own box is not tested, as in the walk)
    for (;;) {
      if (cur & 1u) {
        const uint32_t first = cur >> 4, n = ((cur >> 1) & 7u) + 1u;
        for (uint32_t k = 0; k < n; ++k) {
          if (count == kMax) {
            out[0] = kTileWalk;
            return;
          }
          if (SPHERES) {  // centre and radius beside the slot
            const double* p = sc.prims[first + k].p;
            double* data = (double*)(out + kTileListWords) + 4u * count;
            data[0] = p[0];
            data[1] = p[1];
            data[2] = p[2];
            data[3] = p[3];
          }
          out[1u + count++] = first + k;
        }
      } else {
        const uint32_t first = cur >> (kWideBits + 1u), n = ((cur >> 1) & (kWide - 1u)) + 1u;
        for (uint32_t k = 0; k < n; ++k) {
          const rt_bvh_node rec = nodes[first + k];
          if (beam_misses_box(beam, rec.lo, rec.hi)) continue;
          if (sp == 64u) {
            out[0] = kTileWalk;
            return;
          }
          stack[sp++] = rec.a;  // (the device copy holds the entry code here)
        }
      }
      if (sp == 0u) break;
      cur = stack[--sp];
    }
  }
  out[0] = count;
}

// world.hit(ray, tmin..inf) over the leaf slots of a tile's list (wave-uniform addresses):
// HittableList::hit's first minimum by (t, rank), as closest_hit's.  `list` points at the tile's
// first slot word.
// SPHERES: centre and radius of candidate k sit in the tile's own block, 32 bytes each after the
// 16-word header; the next candidate's are fetched while this one is tested, and no load depends
// on another.
// MEDIA: constant media and records tied to the reference's box chain are settled in place
// (closest_hit sets them aside only because its walk's state is alive around them).
template <bool SPHERES, bool MEDIA>
RT_DEV Closest closest_of_list(const DeviceScene& sc, const uint32_t* list, uint32_t count, V3 o, V3 d, double time,
                               double tmin, const DrawKey& key) {
  const double inf = __longlong_as_double(0x7FF0000000000000ll);
  Closest best;
  best.t = inf;
  best.item = kNoPrim;
  best.kind = 0;
  const double dd = vlen2(d);
  Stats unused;
  if (SPHERES) {
    union Q {
      f32x4 q[2];
      double p[4];
    };
    const f32x4* data = (const f32x4*)(list - 1 + kTileListWords);
    Q next;
    next.q[0] = data[0];
    next.q[1] = data[1];
    uint32_t best_k = kNoPrim;
    for (uint32_t k = 0; k < count; ++k) {
      const Q cur = next;
      if (k + 1u < count) {
        next.q[0] = data[2u * k + 2u];
        next.q[1] = data[2u * k + 3u];
      }
      double t;
      if (!sphere_t<false>(v3(cur.p[0], cur.p[1], cur.p[2]), cur.p[3], o, d, dd, tmin, inf, t, unused)) continue;
      if (t < best.t || (t == best.t && best_k != kNoPrim && sc.ranks[list[k]] < sc.ranks[list[best_k]])) {
        best.t = t;
        best_k = k;
      }
    }
    if (best_k != kNoPrim) best.item = list[best_k];
    return best;
  }
  for (uint32_t k = 0; k < count; ++k) {
    const uint32_t slot = list[k];
    const DPrim whole = sc.prims[slot];
    double t;
    bool hit;
    if ((whole.kind & 15u) == DP_MEDIUM) {
      hit = false;  // (a scene's media are skipped by the kernels that do not carry them)
      if (MEDIA) {
        hit = medium_t<false>(sc, whole, o, d, time, tmin, inf, key, t, unused);
        if (hit && (whole.kind & kChainBit)) hit = reference_chain_passes(sc, (uint32_t)whole.p[3], o, d, tmin);
      }
    } else {
      bool overhang = true;
      hit = base_prim_t<false>(sc, whole, o, d, dd, time, tmin, inf, t, unused, overhang);
      if (MEDIA && hit && (whole.kind & kChainBit) && overhang)
        hit = reference_chain_passes(sc, (uint32_t)whole.p5, o, d, tmin);
    }
    if (hit && (t < best.t || (t == best.t && best.item != kNoPrim && sc.ranks[slot] < sc.ranks[best.item]))) {
      best.t = t;
      best.item = slot;
      best.kind = whole.kind;
    }
  }
  return best;
}

// ------------------------------------------------------------- trace (fused)
// Extend and shade in one kernel, for scenes whose shading is cheap (only
// SolidColor textures): the ray stays in registers between the traversal and the
// scatter, so the hit record, the bin queues and a second pass over the ray
// buffers are never written or read.  Material kinds diverge inside a wave here;
// that costs less than the queue traffic it replaces when shading is a few
// hundred instructions.  Scenes with procedural or image textures take the
// binned extend/shade pair above.
// Per-wave staging of queue appends: a wave owns kWaveStage words of LDS, hands
// out slots with ballot + prefix count, and moves the entries to the global
// queue when kWaveFlushAt have gathered, with one global atomic.  No workgroup
// barrier is involved, so the waves of a workgroup never wait for one another.
#ifndef RT_WAVE_FLUSH_AT
#define RT_WAVE_FLUSH_AT 128
#endif
const uint32_t kWaveFlushAt = RT_WAVE_FLUSH_AT;
const uint32_t kWaveStage = kWaveFlushAt + 64;
#ifndef RT_DEFER_FLUSH_AT
#define RT_DEFER_FLUSH_AT 32
#endif
const uint32_t kDeferFlushAt = RT_DEFER_FLUSH_AT;  // (deferred hits are few)
const uint32_t kDeferStage = kDeferFlushAt + 64;

RT_DEV void wave_flush(const uint32_t* stage, uint32_t n, uint32_t* g_counter, uint32_t* g_queue) {
  wave_lds_sync();
  uint32_t base = 0;
  if (lane_id() == 0) base = atomicAdd(g_counter, n);
  base = __builtin_amdgcn_readfirstlane(base);
  for (uint32_t i = lane_id(); i < n; i += 64u) g_queue[base + i] = stage[i];
  wave_lds_sync();
}

// MEDIA marks the general variant: constant media and records tied to the reference's box
// chain.  A hit on a record with kDeferBit is not shaded here: a Perlin walk needs twice the
// registers of everything else together and would set the occupancy of every ray, and a
// material kind that few lanes of a wave hit would still cost the whole wave its (long)
// scatter code.  The ray, its t and the leaf slot are written out and the path goes to the
// deferred queue, which shade_kernel<kAnyBin> works off after this launch at full lane
// use, appending its survivors to the same next-bounce queue.
// TAIL: once few paths are left, a launch costs more than its rays (a wave needs tens of
// microseconds for one bounce however empty the chip is).  The tail variant takes the queue
// of bounce `bounce` and carries each wave's 64 paths through all remaining bounces in
// place, dead lanes idle: one launch instead of one per bounce.
#ifndef RT_TRACE_MEDIA_WAVES
#define RT_TRACE_MEDIA_WAVES 3
#endif
// MODE: kTraceBounce (one bounce >= 1 of the live queue), kTraceTail, kTraceFirst (bounce 0: the
// camera rays of every path of the pass; a variant of its own so that the other bounces do not
// carry the camera block in scalar registers: that pressure showed as scalar spills all
// through the per-trip code).
// kTraceTailTextured: the tail of a scene with procedural or image textures.  Those hits are
// deferred to a kernel of their own in the other modes (a Perlin walk needs twice the registers
// of everything else); the tail's few paths are shaded in place under a register budget of their
// own (two waves per SIMD), so that the last dozens of bounces are one launch here as well.
enum { kTraceBounce = 0, kTraceTail = 1, kTraceFirst = 2, kTraceTailTextured = 3 };
// Waves per SIMD the register allocation is made for.  The instrumented (STATS) variants carry
// two dozen counters per lane and are not timed: they get registers enough not to spill at all.
constexpr int trace_waves_per_simd(int block, bool stats, bool media, int mode = kTraceBounce) {
  return stats ? (block >= 512 ? 2 : 1) : mode == kTraceTailTextured ? 2 : media ? RT_TRACE_MEDIA_WAVES : 4;
}
// SPHERES: every leaf slot of the scene is a plain sphere without wrappers and nothing is deferred
// (the reference's book_cover.ron and its scale-ups): the leaf test, the hit record and the
// workgroup's copy of the leaf slots (32 bytes each instead of 48) carry that one case only.
// The pixel and sample a path belongs to (the address of its draws) ride in the spare word of
// its attenuation, written by the first bounce: the variants without media read them back after
// the walk instead of deriving them from the path index again on every bounce.
template <int BLOCK, bool STATS, bool MEDIA, bool OVERFLOW, int MODE, bool SPHERES = false>
__global__ void __launch_bounds__(BLOCK, trace_waves_per_simd(BLOCK, STATS, MEDIA, MODE))
trace_kernel(DeviceScene sc, LdsLayout lay, PathBuffers pb, PassInfo ps, uint32_t first_bounce_arg,
             uint32_t max_depth, unsigned long long* g_stats) {
  static_assert(!(MEDIA && SPHERES), "the sphere-only variant carries no media");
  constexpr bool TEXTURED = MODE == kTraceTailTextured;
  static_assert(!(TEXTURED && SPHERES), "the sphere-only variant defers nothing");
  constexpr bool TAIL = MODE == kTraceTail || TEXTURED, FIRST = MODE == kTraceFirst;
  constexpr bool KEY_FROM_PATH = FIRST || MEDIA;  // else from the path's attenuation record
  const uint32_t first_bounce = FIRST ? 0u : first_bounce_arg;
  extern __shared__ __align__(16) char smem[];
  __shared__ uint32_t s_next[(BLOCK / 64) * kWaveStage];
  constexpr bool DEFER = !TAIL;  // (the tail shades everything in place; the host runs none for textured scenes)
  __shared__ uint32_t s_defer[DEFER ? (BLOCK / 64) * kDeferStage : 1];
  const LiveTrips live = live_trips(pb, first_bounce, ps.n_paths);  // (bounce 0: no queue, path = index)
  uint32_t* next_live = next_live_queue(pb, first_bounce, pb.stream ? 0u : blockIdx.x);
  uint32_t* next_count = next_live_counter(pb, first_bounce, pb.stream ? 0u : blockIdx.x);
  uint32_t* counters = pb.counters + (size_t)first_bounce * kCounterStride;
  const uint32_t n_live = live.end;  // (of the trip index space: see LiveTrips)
  if (blockIdx.x * (uint32_t)BLOCK >= n_live) return;
  // Streamed state: where this bounce reads and writes rays and attenuations (the tail works in place).
  const bool stream = pb.stream != 0u;
  const bool in_alt = stream && (first_bounce & 1u), out_alt = stream && (TAIL ? in_alt : !in_alt);
  double4* const here_o = in_alt ? pb.alt_o : pb.ray_o;
  double4* const here_d = in_alt ? pb.alt_d : pb.ray_d;
  double4* const here_beta = in_alt ? pb.alt_beta : pb.beta;
  const double4* const in_o = here_o;
  const double4* const in_d = here_d;
  const double4* const in_beta = here_beta;
  double4* const out_o = out_alt ? pb.alt_o : pb.ray_o;
  double4* const out_d = out_alt ? pb.alt_d : pb.ray_d;
  double4* const out_beta = out_alt ? pb.alt_beta : pb.beta;
  // A wave reserves output positions a chunk at a time (one atomic per chunk) and hands them to its
  // survivors trip by trip; what is left of its last chunk is marked empty when it ends.  Chunks
  // shrink with the queue, so that the empty positions stay a small share of it:
  // waves x chunk <= max(waves x kStreamMinChunk, n_live / 8).
  StreamOut appends, appends_down;
  {
    uint32_t chunk = kStreamMinChunk;
    const uint32_t per_wave = n_live / (gridDim.x * (uint32_t)(BLOCK / 64) * 8u * kStreamClasses);
    while (chunk < kStreamMaxChunk && chunk * 2u <= per_wave) chunk *= 2u;
    appends.init(chunk, next_count, next_live);
    appends_down.init(chunk, next_count, next_live);
  }
  const LdsScene lds = lds_stage<SPHERES ? 2u : 3u>(sc, lay, smem, threadIdx.x, BLOCK);  // the only workgroup barrier
  sc.ops_lds_offset = lds.ops_offset;
  sc.n_ops_lds = lay.n_ops;
  uint32_t* stage = s_next + (threadIdx.x / 64u) * kWaveStage;
  uint32_t* defer_stage = s_defer + (DEFER ? (threadIdx.x / 64u) * kDeferStage : 0u);
  uint32_t n_staged = 0, n_deferred = 0;  // wave-uniform
  Stats st;
  if (STATS) st.clear();
  // A wave's 64 rays per trip.  Which trips a wave runs is settled as it goes: the waves of a
  // launch draw runs of kTripsPerGrab trips from one counter, so that all of them stay busy
  // until the queue is empty whatever their rays cost (with a fixed stride the chip spent the
  // end of every launch waiting for its slowest workgroups).  The tail variant keeps the fixed
  // stride: its paths are few.  The queue entry of the next trip is fetched a trip ahead, so a
  // trip starts with one round trip to memory (the ray); a run is drawn where the last one ends
  // (drawn a trip ahead, the returned value is one more register alive across the whole trip,
  // which measured slower).
  constexpr uint32_t kTripsPerGrab = RT_TRIPS_PER_GRAB;
  // One counter takes about 90 returning atomics per microsecond, and the cheap bounces of a
  // sphere scene draw 70-85 runs of four trips in that time: the trips are split into kWorkSlices
  // stretches with a counter each (in cache lines of their own, the spare words of the bounce's
  // row).  A wave starts in the stretch its workgroup's index selects and moves on to the next
  // one when a stretch has been handed out; after a full round nothing is left.
  constexpr uint32_t kWorkSlices = RT_WORK_SLICES;
  const uint32_t n_trips = n_live >> 6;  // (n_live is a whole number of trips)
  auto slice_start = [&](uint32_t s) -> uint32_t { return ((n_trips * s) / kWorkSlices) << 6; };
  const uint32_t stride = gridDim.x * (uint32_t)BLOCK;
  uint32_t slice = blockIdx.x % kWorkSlices, run_end = 0;  // wave-uniform
  auto draw_run = [&]() -> uint32_t {  // index of the first ray of the run's first trip, or n_live
    for (uint32_t tries = 0; tries < kWorkSlices; ++tries) {
      uint32_t* const word = counters + (slice == 0u ? kWorkWord : 32u * (1u + slice) + kWorkWord);
      uint32_t v = 0;
      if (lane_id() == 0) v = atomicAdd(word, 1u);
      const uint32_t r = (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
      const uint32_t lo = slice_start(slice), hi = slice_start(slice + 1u);
      const uint64_t first = (uint64_t)lo + (uint64_t)r * (kTripsPerGrab * 64u);
      if (first < hi) {
        run_end = (uint32_t)(first + kTripsPerGrab * 64u < hi ? first + kTripsPerGrab * 64u : hi);
        return (uint32_t)first;
      }
      slice = slice + 1u == kWorkSlices ? 0u : slice + 1u;
    }
    return n_live;
  };
  uint32_t base = TAIL ? (uint32_t)__builtin_amdgcn_readfirstlane(
                             (int)(blockIdx.x * (uint32_t)BLOCK + (threadIdx.x & ~63u)))
                       : draw_run();
  uint32_t path_ahead = live.template fetch<FIRST>(base, threadIdx.x & 63u);
  while (base < n_live) {
    uint32_t path = path_ahead;
    bool valid = path != kNoPath;
    // where the next trip starts, when that is known already
    uint32_t next_base = TAIL ? base + stride : base + 64u;
    const bool draws = !TAIL && next_base >= run_end;
    if (!draws && next_base > base) path_ahead = live.template fetch<FIRST>(next_base, threadIdx.x & 63u);
    // the first bounce: the leaf slots this trip's tile of pixels can meet, if the call made lists
    // (a trip is 64 consecutive paths of one sample = one 8x8 tile: wave-uniform)
    uint32_t tile_count = kTileWalk;
    const uint32_t* tile_list = nullptr;
    if (FIRST && !STATS && ps.tile_lists != nullptr) {
      const uint32_t ubase = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
      const uint32_t tile = (ubase - fastdiv(ubase, ps.by_pixels) * ps.n_pixels) >> 6;
      tile_list = ps.tile_lists + (size_t)tile * ps.tile_words + 1u;
      tile_count = tile_list[-1];
    }
    uint32_t bounce = first_bounce;
    bool survives, deferred = false;
    // where the path's ray and attenuation are: its queue position (streamed) or its index
    const uint32_t slot = stream ? base + (threadIdx.x & 63u) : path;
    do {
#ifdef RT_LANE_PROFILE
      unsigned long long tick = __builtin_readcyclecounter();
#endif
      survives = false;
      DrawKey key{ps.seed, 0u, 0u, bounce};
      ShadeOut so;
      so.needs_vector = false;
      so.needs_u = false;
      V3 nb;             // a survivor's new attenuation, its (pixel, sample) word and its time:
      double tag, time;  // set on the way to `survives` (no initial values to keep alive until then)
      if (valid) {
        uint32_t x = 0, y = 0;
        if (KEY_FROM_PATH) {
          uint32_t s_local = fastdiv(path, ps.by_pixels), local = path - s_local * ps.n_pixels;
          pixel_of(ps, local, x, y);
          key.pixel = y * ps.width + x;
          key.sample = ps.sample_base + s_local;
        }
        V3 o, d;
        // the first bounce makes its own camera ray
        if (FIRST) {
          camera_ray(sc, ps, x, y, key, o, d, time);
        } else {
          double4 ro = in_o[slot];
          double4 rdv = in_d[slot];
          o = v3(ro.x, ro.y, ro.z);
          d = v3(rdv.x, rdv.y, rdv.z);
          time = ro.w;
        }
        stat<STATS>(st, RT_STAT_RAYS);
        RT_LP(st, LP_RAY);
        RT_TICK(st, LP_T_SETUP, tick);
        Closest h;
        if (FIRST && !STATS && tile_count != kTileWalk)
          h = closest_of_list<SPHERES, MEDIA>(sc, tile_list, tile_count, o, d, time, 0.001, key);
        else
          h = closest_hit<STATS, MEDIA, MEDIA, OVERFLOW, SPHERES>(sc, lds, o, d, time, 0.001, key, st);
        RT_TICK(st, LP_T_TRAVERSE, tick);
        // ... and starts from beta = 1, L = 0
        V3 beta = v3(1.0, 1.0, 1.0);
        // (tag: pixel and sample as the spare word of the attenuation record keeps them)
        if (!FIRST) {
          double4 bt = in_beta[slot];
          beta = v3(bt.x, bt.y, bt.z);
          tag = bt.w;
          if (!KEY_FROM_PATH) {
            key.pixel = (uint32_t)__double2loint(tag);
            key.sample = (uint32_t)__double2hiint(tag);
          }
        } else {
          tag = __hiloint2double((int)key.sample, (int)key.pixel);
        }
        V3 add = v3(0.0, 0.0, 0.0);
        bool adds = false;
        if (h.item == kNoPrim) {
          // miss: the recursion returns `background` (main.rs:137-139)
          stat<STATS>(st, RT_STAT_SHADE_MISS);
          add = vmul(beta, v3(sc.bg));
          adds = true;
        } else if (!SPHERES && DEFER && (h.kind & kDeferBit)) {
          stat<STATS>(st, RT_STAT_HITS);
          deferred = true;
          // shade_kernel finds ray and attenuation where this bounce read them (the first bounce:
          // where it would have), t in the spare word of the direction, the leaf slot beside them
          if (FIRST) {
            here_o[slot] = make_double4(o.x, o.y, o.z, time);
            here_d[slot] = make_double4(d.x, d.y, d.z, h.t);
            here_beta[slot] = make_double4(1.0, 1.0, 1.0, tag);
          } else {
            reinterpret_cast<double*>(here_d + slot)[3] = h.t;
          }
          pb.hit_item[slot] = h.item;
        } else {
          stat<STATS>(st, RT_STAT_HITS);
          RT_LP(st, LP_SHADE);
          DPrim pr;
          if (SPHERES) {
            pr = sc.prims[h.item];  // (the workgroup's copy holds centre and radius only)
            h.kind = pr.kind;
          } else {
            // (the cold quarter of the record only where something reads it)
            const bool whole = (h.kind & kMaterialElsewhereBit) != 0u;
            pr = whole ? load_prim<true>(sc, lds, h.item) : load_prim<false>(sc, lds, h.item);
          }
          so = shade_begin<STATS, !TEXTURED, SPHERES, true>(sc, (h.kind >> 8) & 7u, pr, material_of(pr), o, d, time,
                                                            h.t, key, st);
          if (so.emits) {
            add = vmul(beta, so.emitted);
            adds = true;
          }
          // depth budget: the scattered ray of the last allowed hit returns black (main.rs:121-122)
          survives = so.scattered && bounce + 1u < max_depth;
          if (survives) nb = vmul(beta, so.attenuation);
        }
        if (sc.has_lights) {
          // L += add; the first bounce also starts every path's L
          if (FIRST) {
            pb.radiance[path] = make_double4(0.0 + add.x, 0.0 + add.y, 0.0 + add.z, 0.0);
          } else if (adds) {
            double4 L = pb.radiance[path];
            pb.radiance[path] = make_double4(L.x + add.x, L.y + add.y, L.z + add.z, 0.0);
          }
        } else if (!survives && !deferred) {
          // nothing emits: L is 0 until the path ends, so it is written once, where it ends
          // (a miss adds the background; absorbed or out of depth adds nothing)
          pb.radiance[path] = make_double4(0.0 + add.x, 0.0 + add.y, 0.0 + add.z, 0.0);
        }
      }
      // Where the survivor's state goes: in place (its index, or its position in the tail), or to the
      // next positions of the wave's chunk of the other set, a fresh chunk when this one is used up.
      uint32_t out_slot = slot;
      if (!TAIL && stream) {
        if (kStreamClasses == 2) {
          const bool up = so.direction.y > 0.0;
          const uint32_t a = appends.take(survives && up), b = appends_down.take(survives && !up);
          out_slot = up ? a : b;
        } else {
          out_slot = appends.take(survives);
        }
      }
      if (survives) {
        // everything but the direction is known before the random vector
        out_beta[out_slot] = make_double4(nb.x, nb.y, nb.z, tag);
        out_o[out_slot] = make_double4(so.p.x, so.p.y, so.p.z, time);
        if (!TAIL && stream) next_live[out_slot] = path;
      }
      RT_TICK(st, LP_T_SHADE, tick);
      {  // whole waves: the lanes look for their random vectors (and uniforms) together
        // (the wave's part of the first two stack rows is idle between traversals)
        double u = 0.0;
        V3 inside = wave_rand_in_unit_sphere(survives && so.needs_vector, survives && so.needs_u, key,
                                             lds.stack - (threadIdx.x & 63u), BLOCK, st, u);
        RT_TICK(st, LP_T_VECTOR, tick);
        shade_finish(so, inside, u);
      }
      if (survives)
        out_d[out_slot] = make_double4(so.direction.x, so.direction.y, so.direction.z, 0.0);
      RT_TICK(st, LP_T_STORE, tick);
      if (TAIL) {
        // the same lane reads its own stores back: program order, no fence needed
        valid = survives;
        ++bounce;
      }
    } while (TAIL && __ballot(valid) != 0ull);
    if (!TAIL && !stream) {
      const unsigned long long mask = __ballot(survives);
      if (survives)
        stage[n_staged + (uint32_t)__popcll(mask & ((1ull << lane_id()) - 1ull))] = path;
      n_staged += (uint32_t)__popcll(mask);
      if (n_staged >= kWaveFlushAt) {
        wave_flush(stage, n_staged, next_count, next_live);
        n_staged = 0;
      }
    }
    if (DEFER) {
      const unsigned long long mask = __ballot(deferred);
      if (mask != 0ull) {
        if (deferred)  // (where the hit lies: the path's index or its queue position)
          defer_stage[n_deferred + (uint32_t)__popcll(mask & ((1ull << lane_id()) - 1ull))] = slot;
        n_deferred += (uint32_t)__popcll(mask);
        if (n_deferred >= kDeferFlushAt) {
          wave_flush(defer_stage, n_deferred, counters + kAnyBin, pb.bins);
          n_deferred = 0;
        }
      }
    }
    if (draws) {
      next_base = draw_run();
      path_ahead = live.template fetch<FIRST>(next_base, threadIdx.x & 63u);
    }
    if (TAIL && next_base <= base) break;  // (32-bit wrap at the very end of the index range)
    base = next_base;
  }
  if (!TAIL && n_staged) wave_flush(stage, n_staged, next_count, next_live);
  if (!TAIL && stream) {
    appends.finish();
    if (kStreamClasses == 2) appends_down.finish();
  }
  if (DEFER && n_deferred) wave_flush(defer_stage, n_deferred, counters + kAnyBin, pb.bins);
  if (STATS) flush_stats(st, g_stats);
}

template <bool STATS, bool SOLID>
void launch_shade(uint32_t bins_present, uint32_t grid, hipStream_t stream, const DeviceScene& ds,
                  const PathBuffers& pb, const PassInfo& ps, uint32_t b, uint32_t max_depth,
                  unsigned long long* gs) {
  // bin 0 (misses) always exists; the others only if the scene has that material kind
  shade_kernel<0, STATS, SOLID><<<grid, kBlock, 0, stream>>>(ds, pb, ps, b, max_depth, gs);
  if (bins_present & 2u) shade_kernel<1, STATS, SOLID><<<grid, kBlock, 0, stream>>>(ds, pb, ps, b, max_depth, gs);
  if (bins_present & 4u) shade_kernel<2, STATS, SOLID><<<grid, kBlock, 0, stream>>>(ds, pb, ps, b, max_depth, gs);
  if (bins_present & 8u) shade_kernel<3, STATS, SOLID><<<grid, kBlock, 0, stream>>>(ds, pb, ps, b, max_depth, gs);
  if (bins_present & 16u) shade_kernel<4, STATS, SOLID><<<grid, kBlock, 0, stream>>>(ds, pb, ps, b, max_depth, gs);
  if (bins_present & 32u) shade_kernel<5, STATS, SOLID><<<grid, kBlock, 0, stream>>>(ds, pb, ps, b, max_depth, gs);
}

// -------------------------------------------------------------- resolve
__global__ void __launch_bounds__(kBlock)
resolve_kernel(PathBuffers pb, PassInfo ps, double* accum) {
  uint32_t pixel = blockIdx.x * kBlock + threadIdx.x;
  if (pixel >= ps.n_pixels) return;
  uint32_t x, y;
  pixel_of(ps, pixel, x, y);
  const uint32_t out = (y - ps.y0) * ps.ww + (x - ps.x0);  // the sums are row-major
  double r, g, b;
  uint32_t s = 0;
  if (ps.first_pass) {  // Iterator::sum = reduce(a + b): starts from the first sample
    double4 L = pb.radiance[pixel];
    r = L.x;
    g = L.y;
    b = L.z;
    s = 1;
  } else {
    r = accum[3 * (size_t)out + 0];
    g = accum[3 * (size_t)out + 1];
    b = accum[3 * (size_t)out + 2];
  }
  for (; s < ps.samples; ++s) {
    double4 L = pb.radiance[(size_t)s * ps.n_pixels + pixel];
    r = r + L.x;
    g = g + L.y;
    b = b + L.z;
  }
  accum[3 * (size_t)out + 0] = r;
  accum[3 * (size_t)out + 1] = g;
  accum[3 * (size_t)out + 2] = b;
}

__global__ void __launch_bounds__(kBlock)
to_float_kernel(const double* in, float* out, uint64_t n) {
  uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < n) out[i] = (float)in[i];
}

// Records as uploaded -> records as the traversal reads them: the entry code in place of `a`,
// `b` cleared, and three never-hit records after the last.
__global__ void __launch_bounds__(kBlock)
device_records_kernel(rt_bvh_node* records, uint32_t n) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i < n) {
    records[i].a = make_entry(records[i].a, records[i].b);
    records[i].b = 0;
  } else if (i < n + kWide - 1u) {
    rt_bvh_node pad;
    for (int a = 0; a < 3; ++a) {
      pad.lo[a] = __int_as_float(0x7F800000);
      pad.hi[a] = __int_as_float((int)0xFF800000u);
    }
    pad.a = 0xFFFFFFFFu;
    pad.b = 0;
    records[i] = pad;
  }
}

// ------------------------------------------------------------ first hit
template <bool CHAINS, bool OVERFLOW>
RT_DEV void first_hit_pixel(const DeviceScene& sc, const LdsScene& lds, const PassInfo& ps,
                            uint32_t local, uint32_t* id, double* t_out, double* normal);

template <bool CHAINS, bool OVERFLOW>
__global__ void __launch_bounds__(kBlock)
first_hit_kernel(DeviceScene sc, LdsLayout lay, PassInfo ps, uint32_t* id, double* t_out,
                 double* normal) {
  extern __shared__ __align__(16) char smem[];
  const LdsScene lds = lds_stage(sc, lay, smem, threadIdx.x, kBlock);
  sc.ops_lds_offset = lds.ops_offset;
  sc.n_ops_lds = lay.n_ops;
  for (uint32_t local = blockIdx.x * kBlock + threadIdx.x; local < ps.n_pixels;
       local += gridDim.x * kBlock)
    first_hit_pixel<CHAINS, OVERFLOW>(sc, lds, ps, local, id, t_out, normal);
}

template <bool CHAINS, bool OVERFLOW>
RT_DEV void first_hit_pixel(const DeviceScene& sc, const LdsScene& lds, const PassInfo& ps,
                            uint32_t local, uint32_t* id, double* t_out, double* normal) {
  uint32_t x, y;
  pixel_of(ps, local, x, y);
  const size_t out = (size_t)(y - ps.y0) * ps.ww + (x - ps.x0);  // the outputs are row-major
  double u = ((double)x + 0.5) / ((double)ps.width - 1.0);
  double v = ((double)(ps.height - y) + 0.5) / ((double)ps.height - 1.0);
  const rt_camera& c = sc.cam;
  V3 offset = v3(0.0, 0.0, 0.0);
  V3 origin = vadd(v3(c.origin), offset);
  V3 dir = vsub(vsub(vadd(vadd(v3(c.lower_left), vscale(v3(c.horizontal), u)),
                          vscale(v3(c.vertical), v)),
                     v3(c.origin)),
                offset);
  DrawKey key{0, 0, 0, 0};
  Stats st;
  Closest h = closest_hit<false, false, CHAINS, OVERFLOW>(sc, lds, origin, dir, c.time0, 0.001, key, st);
  if (h.item == kNoPrim) {
    id[out] = 0xFFFFFFFFu;
    t_out[out] = 0.0;
    normal[3 * out + 0] = 0.0;
    normal[3 * out + 1] = 0.0;
    normal[3 * out + 2] = 0.0;
    return;
  }
  const DPrim pr = sc.prims[h.item];
  Record rec = make_record(sc, pr, origin, dir, c.time0, h.t, false);
  id[out] = pr.object;
  t_out[out] = h.t;
  normal[3 * out + 0] = rec.normal.x;
  normal[3 * out + 1] = rec.normal.y;
  normal[3 * out + 2] = rec.normal.z;
}

// ----------------------------------------------------------------- host

template <typename T>
struct DeviceArray {
  T* p = nullptr;
  size_t n = 0;
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
};

}  // namespace

struct rt_bvh {
  FlatScene flat;
};

struct rt_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string error;
  uint64_t max_paths = 0;
  uint32_t n_cus = 0;
  bool have_scene = false;
  // host copy of what was uploaded; the contexts of a device group share one
  std::shared_ptr<const FlatScene> flat_ptr = std::make_shared<FlatScene>();
  const FlatScene& flat() const { return *flat_ptr; }
  DeviceScene ds{};
  LdsLayout lds{};
  DeviceArray<float4> nodes;
  DeviceArray<DPrim> prims, boundary;
  DeviceArray<uint32_t> ranks;
  DeviceArray<DOp> ops;
  DeviceArray<DAux> aux;
  DeviceArray<DMaterial> materials;
  DeviceArray<DTexture> textures;
  DeviceArray<rt_perlin> perlin;
  DeviceArray<DImage> images;
  DeviceArray<uint8_t> texels;
  DeviceArray<double> chain_boxes;
  DeviceArray<uint32_t> chain_ranges;
  // path buffers, grown on demand
  PathBuffers pb{};
  uint64_t pb_capacity = 0;
  uint64_t pb_state = 0, pb_live = 0;  // entries of each ray/attenuation array and of each live queue
  bool pb_alt = false;                  // the second set of ray/attenuation arrays exists
  bool stream_on = true;                // streamed path state in the fused pipeline (MI355RT_STREAM=0: no)
  // ... for passes of at least this many paths (MI355RT_STREAM_MIN_PATHS).  Smaller passes run
  // faster by index: their launches are short enough for the chunk reservations, all on one
  // counter, to reach the rate that word can take (a 400x225x64 render: 1.6 ms by index, 1.8 ms
  // streamed; even at 2^24 paths, ahead from 2^25).
  uint64_t stream_min_paths = 1ull << 24;
  bool max_paths_auto = false;          // max_paths came from the free device memory, not from the caller
  uint32_t pb_queues = 0;  // bin queues allocated in pb.bins
  bool solid = true;       // every texture is a SolidColor
  bool general = false;    // the scene needs the general trace kernel
  bool textured_tail = true;  // scenes with procedural or image textures run the tail launch too (MI355RT_TEXTURED_TAIL=0: no)
  bool tile_lists_on = true;  // camera-ray tile lists for the plain kernels (MI355RT_TILE_LISTS=0: no)
  DeviceArray<uint32_t> tile_lists;
  bool spheres = false;    // every leaf slot is a plain sphere without wrappers: the sphere-only trace kernel
  uint32_t counters_depth = 0;
  DeviceArray<double> accum;
  DeviceArray<unsigned long long> stats;
  uint32_t* h_probe = nullptr;  // pinned word for queue-size probes
  LdsLayout lds_trace{};        // layout for trace_kernel's workgroup size
  uint32_t trace_block = 256;   // threads per trace_kernel workgroup: 256, 512 or 1024
  uint32_t trace_blocks_per_cu = 16;
  uint32_t blocks_per_cu = 16;  // 256-thread workgroups per CU in a grid (extend, shade, first hit)
  uint32_t tail_paths = 1u << 20;  // fused pipeline: at most this many live paths finish in one launch
  DeviceArray<uint32_t> stack_overflow;
  std::vector<hipEvent_t> events;  // RT_FLAG_TIMING: three per bounce (before extend, between, after shade)
  DeviceArray<unsigned char> staging;     // device side of host-buffer results (rt_render, rt_first_hit)
  unsigned char* scratch_ptr = nullptr;  // this device's copy of `scratch` (rt_render_device_noalloc)
  // Device group (rt_config.n_devices > 1): this context is the root (devices[0]) and owns one
  // context per further device, a float buffer per member for its share of the samples, and
  // the RCCL communicators the shares are reduced over (none: the shares are added through host memory).
  std::vector<rt_ctx*> peers;
  DeviceArray<float> share;
  void* rccl = nullptr;                // dlopen handle of librccl
  std::vector<void*> comms;            // ncclComm_t per member, root first
  bool group = false;                  // render calls go through render_group (peers, or a forced group of one)
  uint32_t reduce_path = RT_REDUCE_NONE;
  double last_render_ms_min = 0.0, last_render_ms_max = 0.0, last_reduce_ms = 0.0;  // rt_group_info
  // Progress of a render call (rt_set_progress): the callback lives on the context the caller
  // holds; the members of a group report to it through `progress_to`, one at a time.
  rt_progress_fn progress_fn = nullptr;
  void* progress_user = nullptr;
  rt_ctx* progress_to = nullptr;  // set on a member while its group renders
  std::mutex progress_mutex;
  uint64_t progress_done = 0, progress_total = 0;
};

namespace {

int fail(rt_ctx* ctx, int code, const std::string& msg) {
  if (ctx) ctx->error = msg;
  return code;
}

#define RT_HIP(ctx, call)                                                                   \
  do {                                                                                      \
    hipError_t e__ = (call);                                                                \
    if (e__ != hipSuccess)                                                                  \
      return fail(ctx, e__ == hipErrorOutOfMemory ? RT_ERR_NOMEM : RT_ERR_DEVICE,           \
                  std::string(#call) + ": " + hipGetErrorString(e__));                      \
  } while (0)

template <typename T, typename S>
int upload(rt_ctx* ctx, DeviceArray<T>& dst, const std::vector<S>& src) {
  static_assert(sizeof(T) % sizeof(S) == 0 || sizeof(S) % sizeof(T) == 0, "layout");
  dst.release();
  size_t bytes = src.size() * sizeof(S);
  // keep every array non-null so kernels may form addresses even for empty scenes
  RT_HIP(ctx, hipMalloc((void**)&dst.p, std::max<size_t>(bytes, 64)));
  dst.n = bytes / sizeof(T);
  if (bytes) RT_HIP(ctx, hipMemcpy(dst.p, src.data(), bytes, hipMemcpyHostToDevice));
  return RT_OK;
}

void free_paths(rt_ctx* ctx) {
  PathBuffers& pb = ctx->pb;
  void* ptrs[] = {pb.ray_o, pb.ray_d, pb.beta, pb.radiance, pb.hit_item,
                  pb.live[0], pb.live[1], pb.bins, pb.counters, pb.alt_o, pb.alt_d, pb.alt_beta};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  pb = PathBuffers{};
  ctx->pb_capacity = 0;
  ctx->pb_state = ctx->pb_live = 0;
  ctx->pb_alt = false;
  ctx->pb_queues = 0;
  ctx->counters_depth = 0;
}

// `queues`: bin queues the pipeline needs (0: fused with nothing deferred, 1: the deferred
// queue, kBins: binned); the hit slot array exists whenever a queue does.
// `stream_entries`: 0, or the positions a streamed render's queues and state arrays must hold
// (stream_positions): then the second set of state arrays exists too and a live queue is one run
// of that many entries instead of kSubQueues regions of `capacity`.
int ensure_paths(rt_ctx* ctx, uint64_t capacity, uint32_t max_depth, uint32_t queues, uint64_t stream_entries = 0) {
  PathBuffers& pb = ctx->pb;
  const bool alt = stream_entries != 0;
  // (held by index, a live queue is kSubQueues regions of pb.capacity entries, the largest capacity so far)
  uint64_t state = alt ? stream_entries : capacity;
  uint64_t live = alt ? stream_entries : std::max(capacity, ctx->pb_capacity) * kSubQueues;
  if (capacity > ctx->pb_capacity || queues > ctx->pb_queues || state > ctx->pb_state || live > ctx->pb_live ||
      (alt && !ctx->pb_alt)) {
    queues = std::max(queues, ctx->pb_queues);
    capacity = std::max(capacity, ctx->pb_capacity);
    state = std::max(state, ctx->pb_state);
    live = std::max(live, ctx->pb_live);
    const bool want_alt = alt || ctx->pb_alt;
    uint32_t* keep_counters = pb.counters;
    uint32_t keep_depth = ctx->counters_depth;
    pb.counters = nullptr;
    free_paths(ctx);
    pb.counters = keep_counters;
    ctx->counters_depth = keep_depth;
    RT_HIP(ctx, hipMalloc((void**)&pb.ray_o, state * sizeof(double4)));
    RT_HIP(ctx, hipMalloc((void**)&pb.ray_d, state * sizeof(double4)));
    RT_HIP(ctx, hipMalloc((void**)&pb.beta, state * sizeof(double4)));
    if (want_alt) {
      RT_HIP(ctx, hipMalloc((void**)&pb.alt_o, state * sizeof(double4)));
      RT_HIP(ctx, hipMalloc((void**)&pb.alt_d, state * sizeof(double4)));
      RT_HIP(ctx, hipMalloc((void**)&pb.alt_beta, state * sizeof(double4)));
    }
    RT_HIP(ctx, hipMalloc((void**)&pb.radiance, capacity * sizeof(double4)));
    RT_HIP(ctx, hipMalloc((void**)&pb.live[0], live * sizeof(uint32_t)));
    RT_HIP(ctx, hipMalloc((void**)&pb.live[1], live * sizeof(uint32_t)));
    if (queues) {  // hits are recorded and paths queued only for a separate shade kernel
      RT_HIP(ctx, hipMalloc((void**)&pb.hit_item, state * sizeof(uint32_t)));  // (by path index or by position)
      RT_HIP(ctx, hipMalloc((void**)&pb.bins, capacity * queues * sizeof(uint32_t)));
    }
    pb.capacity = capacity;
    ctx->pb_capacity = capacity;
    ctx->pb_state = state;
    ctx->pb_live = live;
    ctx->pb_alt = want_alt;
    ctx->pb_queues = queues;
  }
  if (max_depth > ctx->counters_depth) {
    if (pb.counters) (void)hipFree(pb.counters);
    pb.counters = nullptr;
    RT_HIP(ctx, hipMalloc((void**)&pb.counters,
                          (size_t)max_depth * kCounterStride * sizeof(uint32_t)));
    ctx->counters_depth = max_depth;
  }
  return RT_OK;
}

// The same buffers as ensure_paths makes, laid out in this device's copy of `scratch`
// (256-byte aligned pieces), plus the window's f64 sums.  Picks the largest number of samples
// per pass that fits; fails when not even one sample of every window pixel does.
int carve_scratch(rt_ctx* ctx, uint64_t n_pix, uint32_t n_samples, uint32_t max_depth, uint32_t queues,
                  PathBuffers& pb, double*& accum, uint32_t& per_pass) {
  auto up = [](uint64_t b) { return (b + 255u) & ~(uint64_t)255u; };
  const uint64_t fixed = up(n_pix * 3 * sizeof(double)) +
                         up((uint64_t)max_depth * kCounterStride * sizeof(uint32_t));
  auto need = [&](uint64_t cap) {
    return fixed + 4 * up(cap * sizeof(double4)) + 2 * up(cap * kSubQueues * sizeof(uint32_t)) +
           (queues ? up(cap * sizeof(uint32_t)) + up(cap * queues * sizeof(uint32_t)) : 0);
  };
  uint64_t s = std::max<uint32_t>(n_samples, 1u);
  while (s > 1 && (n_pix * s > ctx->max_paths || need(n_pix * s) > RT_NOALLOC_SCRATCH_BYTES)) s = (s + 1) / 2;
  if (need(n_pix * s) > RT_NOALLOC_SCRATCH_BYTES)
    return fail(ctx, RT_ERR_LIMIT,
                "rt_render_device_noalloc: a window of " + std::to_string(n_pix) + " pixels at depth " +
                    std::to_string(max_depth) + " needs " + std::to_string(need(n_pix)) +
                    " bytes of scratch, the limit is " + std::to_string(RT_NOALLOC_SCRATCH_BYTES));
  const uint64_t cap = n_pix * s;
  unsigned char* at = ctx->scratch_ptr;
  auto take = [&](uint64_t bytes) {
    unsigned char* r = at;
    at += up(bytes);
    return r;
  };
  pb = PathBuffers{};
  accum = (double*)take(n_pix * 3 * sizeof(double));
  pb.counters = (uint32_t*)take((uint64_t)max_depth * kCounterStride * sizeof(uint32_t));
  pb.ray_o = (double4*)take(cap * sizeof(double4));
  pb.ray_d = (double4*)take(cap * sizeof(double4));
  pb.beta = (double4*)take(cap * sizeof(double4));
  pb.radiance = (double4*)take(cap * sizeof(double4));
  pb.live[0] = (uint32_t*)take(cap * kSubQueues * sizeof(uint32_t));
  pb.live[1] = (uint32_t*)take(cap * kSubQueues * sizeof(uint32_t));
  if (queues) {
    pb.hit_item = (uint32_t*)take(cap * sizeof(uint32_t));
    pb.bins = (uint32_t*)take(cap * queues * sizeof(uint32_t));
  }
  pb.capacity = cap;
  per_pass = (uint32_t)s;
  return RT_OK;
}

int check_window(rt_ctx* ctx, uint32_t width, uint32_t height, const rt_window* win, PassInfo& ps) {
  if (width < 2 || height < 2) return fail(ctx, RT_ERR_INVALID, "image must be at least 2x2");
  if ((uint64_t)width * height > 0xFFFFFFFFull)
    return fail(ctx, RT_ERR_LIMIT, "image has more than 2^32-1 pixels");
  ps.width = width;
  ps.height = height;
  if (!win || win->w == 0 || win->h == 0) {
    ps.x0 = ps.y0 = 0;
    ps.ww = width;
    ps.wh = height;
  } else {
    if ((uint64_t)win->x0 + win->w > width || (uint64_t)win->y0 + win->h > height)
      return fail(ctx, RT_ERR_INVALID, "window exceeds the image");
    ps.x0 = win->x0;
    ps.y0 = win->y0;
    ps.ww = win->w;
    ps.wh = win->h;
  }
  ps.n_pixels = ps.ww * ps.wh;
  ps.by_pixels = make_fastdiv(ps.n_pixels);
  const char* tiles_env = getenv("MI355RT_TILES");  // 0: row-major numbering (A/B timing)
  const bool tiles_off = tiles_env && atoi(tiles_env) == 0;
  ps.tiled = (!tiles_off && ps.ww % 8u == 0 && ps.wh % 8u == 0) ? 1u : 0u;
  ps.by_ww = make_fastdiv(ps.tiled ? ps.ww / 8u : ps.ww);
  // the two divisors are checked on the values where a wrong multiplier would show first
  for (const FastDiv* f : {&ps.by_pixels, &ps.by_ww})
    for (uint64_t k : {(uint64_t)1, (uint64_t)2, (uint64_t)3, (uint64_t)(0xFFFFFFFFu / f->d)})
      for (int64_t e = -1; e <= 1; ++e) {
        uint64_t n = k * f->d + (uint64_t)e;
        if (n <= 0xFFFFFFFFull && fastdiv((uint32_t)n, *f) != (uint32_t)(n / f->d))
          return fail(ctx, RT_ERR_INVALID, "internal: invariant division check failed");
      }
  return RT_OK;
}

uint32_t ops_count_host(uint32_t w) { return w >> 24; }
uint32_t blocks_for(uint64_t n) { return (uint32_t)((n + kBlock - 1) / kBlock); }

// LDS plan for workgroups of `block` threads within `budget` bytes: the traversal
// stack first (it is not optional), then as many top-of-tree records as fit, then
// leaf slots with what is left.
LdsLayout plan_lds(const FlatScene& f, uint32_t block, uint32_t budget, uint32_t stack_cap,
                   uint32_t slot_bytes = kPrimHotBytes) {
  LdsLayout l{};
  // one spare entry: a visit stores before it knows whether it keeps; two rows at least: they
  // double as the exchange area of wave_rand_in_unit_sphere.  The worst-case need can be several
  // times what a ray ever uses; entries beyond stack_cap go to the global overflow area.
  const uint32_t need = std::max(f.stack_entries + 1u, 2u);
  l.stack_depth = std::min(need, std::max(stack_cap, 2u));
  l.entry_bits = entry_code_bits((uint32_t)f.nodes.size(), (uint32_t)f.items.size());
  // rows for the leaf slots a traversal sets aside (constant media, records tied to the
  // reference's box chain): as many as a ray can meet, eight at most (a full list is settled
  // mid-traversal), none for scenes without such records
  uint32_t flagged = 0;
  for (const DPrim& pr : f.prims) flagged += ((pr.kind & 15u) == DP_MEDIUM || (pr.kind & kChainBit)) ? 1u : 0u;
  l.cand_rows = std::min(flagged, 8u);
  // scenes with a few dozen wrapper ops keep them in LDS (cornell: four)
  l.n_ops = f.ops.size() <= 64 ? (uint32_t)f.ops.size() : 0u;
  uint32_t used = (l.stack_depth + l.cand_rows) * block * (uint32_t)sizeof(uint32_t) + l.n_ops * 32u;
  budget = std::max(budget, used + 4096u);  // leave room for a few records
  uint32_t room = budget - used;
  // (+3: the never-hit records that end the device copy, so the last group stays in LDS)
  l.n_nodes = std::min<uint32_t>((uint32_t)f.nodes.size() + kWide - 1u, room / 32u);
  room -= l.n_nodes * 32u;
  l.n_prims = std::min<uint32_t>((uint32_t)f.prims.size(), room / slot_bytes);
  l.bytes = used + l.n_nodes * 32u + l.n_prims * slot_bytes;
  return l;
}

// A run-time trace mode as a compile-time one.
template <typename F>
auto with_mode3(int mode, F&& f) {  // (the modes of a scene without textures)
  if (mode == kTraceTail) return f(std::integral_constant<int, kTraceTail>{});
  if (mode == kTraceFirst) return f(std::integral_constant<int, kTraceFirst>{});
  return f(std::integral_constant<int, kTraceBounce>{});
}
template <typename F>
auto with_mode(int mode, F&& f) {
  if (mode == kTraceTailTextured) return f(std::integral_constant<int, kTraceTailTextured>{});
  if (mode == kTraceTail) return f(std::integral_constant<int, kTraceTail>{});
  if (mode == kTraceFirst) return f(std::integral_constant<int, kTraceFirst>{});
  return f(std::integral_constant<int, kTraceBounce>{});
}
// A run-time bool as a compile-time one: f(std::true_type) or f(std::false_type).
template <typename F>
auto with_bool(bool v, F&& f) {
  return v ? f(std::true_type{}) : f(std::false_type{});
}

// The sphere-only variant: 512-thread workgroups, never together with media.
void launch_trace_spheres(rt_ctx* ctx, bool stats, bool tail, uint32_t grid, const PathBuffers& pb,
                          const PassInfo& ps, uint32_t b, uint32_t max_depth, unsigned long long* gs) {
  const LdsLayout& l = ctx->lds_trace;
  with_bool(stats, [&](auto S) {
    return with_bool(l.overflow != nullptr, [&](auto O) {
      return with_mode3(tail ? kTraceTail : b == 0 ? kTraceFirst : kTraceBounce, [&](auto T) {
        trace_kernel<512, decltype(S)::value, false, decltype(O)::value, decltype(T)::value, true>
            <<<grid, 512, l.bytes, ctx->stream>>>(ctx->ds, l, pb, ps, b, max_depth, gs);
        return 0;
      });
    });
  });
}
template <int BLOCK>
void launch_trace_block(rt_ctx* ctx, bool stats, bool media, bool tail, uint32_t grid,
                        const PathBuffers& pb, const PassInfo& ps, uint32_t b, uint32_t max_depth,
                        unsigned long long* gs) {
  const LdsLayout& l = ctx->lds_trace;
  with_bool(stats, [&](auto S) {
    return with_bool(media, [&](auto M) {
      return with_bool(l.overflow != nullptr, [&](auto O) {
        return with_mode(tail ? (ctx->solid ? kTraceTail : kTraceTailTextured) : b == 0 ? kTraceFirst : kTraceBounce,
                         [&](auto T) {
          trace_kernel<BLOCK, decltype(S)::value, decltype(M)::value, decltype(O)::value,
                       decltype(T)::value>
              <<<grid, BLOCK, l.bytes, ctx->stream>>>(ctx->ds, l, pb, ps, b, max_depth, gs);
          return 0;
        });
      });
    });
  });
}
void launch_trace(rt_ctx* ctx, bool stats, bool media, bool tail, uint32_t grid, const PathBuffers& pb,
                  const PassInfo& ps, uint32_t b, uint32_t max_depth, unsigned long long* gs) {
  if (ctx->spheres) launch_trace_spheres(ctx, stats, tail, grid, pb, ps, b, max_depth, gs);
  else if (ctx->trace_block == 512) launch_trace_block<512>(ctx, stats, media, tail, grid, pb, ps, b, max_depth, gs);
  else launch_trace_block<256>(ctx, stats, media, tail, grid, pb, ps, b, max_depth, gs);
}
void launch_extend(rt_ctx* ctx, bool stats, bool media, uint32_t grid, const PathBuffers& pb,
                   const PassInfo& ps, uint32_t b, unsigned long long* gs) {
  const LdsLayout& l = ctx->lds;
  with_bool(stats, [&](auto S) {
    return with_bool(media, [&](auto M) {
      return with_bool(l.overflow != nullptr, [&](auto O) {
        extend_kernel<decltype(S)::value, decltype(M)::value, decltype(O)::value>
            <<<grid, kBlock, l.bytes, ctx->stream>>>(ctx->ds, l, pb, ps, b, gs);
        return 0;
      });
    });
  });
}
// Kernels that may be launched with more than the default dynamic LDS limit.
template <int BLOCK>
hipError_t allow_trace_lds(int bytes) {
  const hipFuncAttribute attr = hipFuncAttributeMaxDynamicSharedMemorySize;
  hipError_t e = hipSuccess;
  for (int v = 0; v < 32 && e == hipSuccess; ++v)
    e = with_bool(v & 1, [&](auto S) {
      return with_bool(v & 2, [&](auto M) {
        return with_bool(v & 4, [&](auto O) {
          return with_mode(v >> 3, [&](auto T) {
            return hipFuncSetAttribute(
                (const void*)trace_kernel<BLOCK, decltype(S)::value, decltype(M)::value,
                                          decltype(O)::value, decltype(T)::value>,
                attr, bytes);
          });
        });
      });
    });
  return e;
}
hipError_t allow_trace_spheres_lds(int bytes) {
  const hipFuncAttribute attr = hipFuncAttributeMaxDynamicSharedMemorySize;
  hipError_t e = hipSuccess;
  for (int v = 0; v < 12 && e == hipSuccess; ++v)
    e = with_bool(v & 1, [&](auto S) {
      return with_bool(v & 2, [&](auto O) {
        return with_mode3(v >> 2, [&](auto T) {
          return hipFuncSetAttribute(
              (const void*)trace_kernel<512, decltype(S)::value, false, decltype(O)::value, decltype(T)::value, true>,
              attr, bytes);
        });
      });
    });
  return e;
}
hipError_t allow_extend_lds(int bytes) {
  const hipFuncAttribute attr = hipFuncAttributeMaxDynamicSharedMemorySize;
  hipError_t e = hipSuccess;
  for (int v = 0; v < 8 && e == hipSuccess; ++v)
    e = with_bool(v & 1, [&](auto S) {
      return with_bool(v & 2, [&](auto M) {
        return with_bool(v & 4, [&](auto O) {
          return hipFuncSetAttribute(
              (const void*)extend_kernel<decltype(S)::value, decltype(M)::value, decltype(O)::value>, attr,
              bytes);
        });
      });
    });
  for (int v = 0; v < 4 && e == hipSuccess; ++v)
    e = with_bool(v & 1, [&](auto C) {
      return with_bool(v & 2, [&](auto O) {
        return hipFuncSetAttribute((const void*)first_hit_kernel<decltype(C)::value, decltype(O)::value>,
                                   attr, bytes);
      });
    });
  return e;
}

// Render the sample range into f64 sums (window-sized) on the device; *accum_out names them.
// in_scratch: every buffer comes out of this device's `scratch` array and nothing is allocated
// (no statistics, no launch timing: their events would have to be created).
int render_into_accum(rt_ctx* ctx, const rt_render_params* p, uint64_t* stats, double** accum_out,
                      bool in_scratch = false) {
  if (!ctx) return RT_ERR_INVALID;
  if (!p) return fail(ctx, RT_ERR_INVALID, "null params");
  if (!ctx->have_scene) return fail(ctx, RT_ERR_INVALID, "no scene uploaded");
  if (p->sample_end < p->sample_begin) return fail(ctx, RT_ERR_INVALID, "empty sample range");
  RT_HIP(ctx, hipSetDevice(ctx->device));
  PassInfo ps{};
  int rc = check_window(ctx, p->width, p->height, &p->window, ps);
  if (rc != RT_OK) return rc;
  ps.seed = p->seed;
  const bool want_stats = (p->flags & RT_FLAG_STATS) && stats && !in_scratch;
  const bool want_timing = (p->flags & RT_FLAG_TIMING) && stats && !in_scratch;
  size_t n_events = 0;
  double extend_ms = 0.0, shade_ms = 0.0;
  uint64_t timed_launches = 0;
  auto stamp = [&]() -> hipError_t {
    if (n_events == ctx->events.size()) {
      hipEvent_t e;
      hipError_t rc2 = hipEventCreate(&e);
      if (rc2 != hipSuccess) return rc2;
      ctx->events.push_back(e);
    }
    return hipEventRecord(ctx->events[n_events++], ctx->stream);
  };
  // elapsed times of the recorded triples; the stream must be idle
  auto drain = [&]() -> hipError_t {
    for (size_t i = 0; i + 2 < n_events; i += 3) {
      float a = 0.f, b = 0.f;
      hipError_t rc2 = hipEventElapsedTime(&a, ctx->events[i], ctx->events[i + 1]);
      if (rc2 == hipSuccess) rc2 = hipEventElapsedTime(&b, ctx->events[i + 1], ctx->events[i + 2]);
      if (rc2 != hipSuccess) return rc2;
      extend_ms += a;
      shade_ms += b;
      ++timed_launches;
    }
    n_events = 0;
    return hipSuccess;
  };
  const uint32_t n_samples = p->sample_end - p->sample_begin;
  const uint64_t n_pix = ps.n_pixels;
  if (n_pix > ctx->max_paths)
    return fail(ctx, RT_ERR_LIMIT, "window has more pixels than max_paths_in_flight");

  // the MEDIA kernels carry the rare cases: constant media and records tied to the reference's box chain
  const bool media = ctx->flat().has_media || !ctx->flat().chain_ranges.empty();
  const bool solid = ctx->solid;
  // traversal and scatter run fused, with the hits on procedural or image textures deferred
  // to a shade kernel of their own; RT_FLAG_BINNED forces the extend / shade-per-material pipeline
  const bool fused = !(p->flags & RT_FLAG_BINNED);
  const bool defer = fused && ctx->flat().defers;
  const bool general = ctx->general;
  const uint32_t queues = fused ? (defer ? 1u : 0u) : kBins;
  uint32_t bins_present = 1u;
  for (const DMaterial& m : ctx->flat().materials) bins_present |= 2u << m.kind;

  // Streamed path state (PathBuffers): the fused kernels, outside the no-allocation entry point
  // (whose scratch array is laid out for state by path index).
  bool stream = fused && !in_scratch && ctx->stream_on;
  const uint32_t tmax_grid = ctx->n_cus * std::max(1u, ctx->trace_blocks_per_cu);
  // Positions a streamed pass of n paths can reach: its survivors plus what its waves leave unused
  // of their last chunks (trace_kernel: waves x chunk <= max(waves x kStreamMinChunk, positions / 8)).
  auto stream_positions = [&](uint64_t n) {
    const uint64_t waves = (uint64_t)tmax_grid * (std::max(ctx->trace_block, 512u) / 64u);
    // (the deferred shade kernel's waves append with the smallest chunk)
    const uint64_t shade_waves = defer ? (uint64_t)ctx->n_cus * ctx->blocks_per_cu * (kBlock / 64u) : 0u;
    return std::max(n + n / 7u, n + waves * kStreamClasses * kStreamMinChunk) + 2u * shade_waves * kStreamMinChunk + 1024u;
  };
  PathBuffers scratch_pb{};
  double* accum = nullptr;
  uint32_t per_pass = 0;
  if (in_scratch) {
    rc = carve_scratch(ctx, n_pix, n_samples, std::max(p->max_depth, 1u), queues, scratch_pb, accum, per_pass);
    if (rc != RT_OK) return rc;
  } else {
    if (ctx->accum.n < n_pix * 3) {
      ctx->accum.release();
      RT_HIP(ctx, hipMalloc((void**)&ctx->accum.p, n_pix * 3 * sizeof(double)));
      ctx->accum.n = n_pix * 3;
    }
    accum = ctx->accum.p;
    // (a streamed path takes about 270 bytes where one held by index takes 200: see create_single)
    const uint64_t max_paths = stream && ctx->max_paths_auto ? std::max<uint64_t>(ctx->max_paths / 27u * 20u, n_pix) : ctx->max_paths;
    per_pass = (uint32_t)std::min<uint64_t>(max_paths / n_pix, n_samples);
    if (per_pass == 0) per_pass = 1;
    // passes of equal size (1024 samples with room for 486 at a time: 342 + 341 + 341, not 486 + 486 + 52)
    if (n_samples) {
      const uint32_t n_passes = (n_samples + per_pass - 1) / per_pass;
      per_pass = (n_samples + n_passes - 1) / n_passes;
    }
  }
  stream = stream && n_pix * per_pass >= ctx->stream_min_paths;
  *accum_out = accum;
  RT_HIP(ctx, hipMemsetAsync(accum, 0, n_pix * 3 * sizeof(double), ctx->stream));
  if (want_stats)
    RT_HIP(ctx, hipMemsetAsync(ctx->stats.p, 0, RT_STAT_COUNT * sizeof(unsigned long long),
                               ctx->stream));
  if (n_samples == 0 || p->max_depth == 0) {
    // depth 0: ray_color returns black for every sample (main.rs:121-122)
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (want_stats) std::memset(stats, 0, RT_STAT_COUNT * sizeof(uint64_t));
    return RT_OK;
  }

  if (!in_scratch) {
    // (device memory that turns out not to be there after all: smaller passes)
    while ((rc = ensure_paths(ctx, n_pix * per_pass, p->max_depth, queues,
                              stream ? stream_positions(n_pix * per_pass) : 0u)) == RT_ERR_NOMEM && per_pass > 1) {
      (void)hipGetLastError();
      free_paths(ctx);
      per_pass = (per_pass + 1) / 2;
    }
    if (rc != RT_OK) return rc;
  }
  PathBuffers pb = in_scratch ? scratch_pb : ctx->pb;
  pb.stream = stream ? 1u : 0u;
  const uint32_t max_grid = ctx->n_cus * ctx->blocks_per_cu;
  // What each 8x8 tile's camera rays can meet, once per call (see tile_list_kernel): the fused
  // kernels' first bounce tests those few leaf slots instead of walking the hierarchy per ray.
  // Not for the instrumented kernels (their counts are the walk's), nor where nothing may be allocated.
  ps.tile_lists = nullptr;
  if (fused && ps.tiled && ctx->tile_lists_on && !want_stats && !in_scratch) {
    const uint32_t n_tiles = (ps.ww >> 3) * (ps.wh >> 3);
    ps.tile_words = ctx->spheres ? kTileSphereWords : kTileListWords;
    const size_t words = (size_t)n_tiles * ps.tile_words;
    if (ctx->tile_lists.n < words) {
      ctx->tile_lists.release();
      RT_HIP(ctx, hipMalloc((void**)&ctx->tile_lists.p, words * sizeof(uint32_t)));
      ctx->tile_lists.n = words;
    }
    if (ctx->spheres)
      tile_list_kernel<true><<<blocks_for(n_tiles), kBlock, 0, ctx->stream>>>(ctx->ds, ps, ctx->tile_lists.p, n_tiles);
    else
      tile_list_kernel<false><<<blocks_for(n_tiles), kBlock, 0, ctx->stream>>>(ctx->ds, ps, ctx->tile_lists.p, n_tiles);
    ps.tile_lists = ctx->tile_lists.p;
  }
  rt_ctx* const report = ctx->progress_to ? ctx->progress_to : ctx;
  if (!ctx->progress_to) {  // (a group sets its totals itself)
    ctx->progress_done = 0;
    ctx->progress_total = n_pix * n_samples;
  }

  for (uint32_t done = 0; done < n_samples; done += per_pass) {
    ps.samples = std::min(per_pass, n_samples - done);
    ps.sample_base = p->sample_begin + done;
    ps.n_paths = (uint32_t)(n_pix * ps.samples);
    ps.first_pass = done == 0 ? 1u : 0u;
    RT_HIP(ctx, hipMemsetAsync(pb.counters, 0,
                               (size_t)p->max_depth * kCounterStride * sizeof(uint32_t),
                               ctx->stream));
    if (!fused)  // the fused kernel makes the camera rays in its first bounce
      raygen_kernel<<<blocks_for(ps.n_paths), kBlock, 0, ctx->stream>>>(ctx->ds, pb, ps);
    uint32_t grid = std::min(blocks_for(ps.n_paths), max_grid);
    auto trace_blocks_for = [&](uint64_t n) {
      return (uint32_t)std::min<uint64_t>((n + ctx->trace_block - 1) / ctx->trace_block, tmax_grid);
    };
    uint32_t tgrid = std::max(1u, trace_blocks_for(ps.n_paths));
    for (uint32_t b = 0; b < p->max_depth; ++b) {
      unsigned long long* gs = ctx->stats.p;
      if (want_timing) RT_HIP(ctx, stamp());
      if (fused) {
        // cheap shading: traversal and scatter in one kernel per bounce
        launch_trace(ctx, want_stats, general, false, tgrid, pb, ps, b, p->max_depth, gs);
        if (want_timing) RT_HIP(ctx, stamp());
        if (defer)
          with_bool(want_stats, [&](auto S) {
            return with_bool(solid, [&](auto T) {
              shade_kernel<kAnyBin, decltype(S)::value, decltype(T)::value>
                  <<<grid, kBlock, 0, ctx->stream>>>(ctx->ds, pb, ps, b, p->max_depth, gs);
              return 0;
            });
          });
      } else {
        launch_extend(ctx, want_stats, media, grid, pb, ps, b, gs);
        if (want_timing) RT_HIP(ctx, stamp());
        with_bool(want_stats, [&](auto S) {
          return with_bool(solid, [&](auto T) {
            launch_shade<decltype(S)::value, decltype(T)::value>(bins_present, grid, ctx->stream, ctx->ds, pb,
                                                                 ps, b, p->max_depth, gs);
            return 0;
          });
        });
      }
      if (want_timing) RT_HIP(ctx, stamp());
      // every 8 bounces look at the live count: stop when no path is left and
      // shrink the grid as the queues drain
      if ((b & 7u) == 7u && b + 1 < p->max_depth) {
        RT_HIP(ctx, hipMemcpyAsync(ctx->h_probe, pb.counters + (size_t)b * kCounterStride,
                                   kCounterStride * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        uint32_t live = 0;
        for (uint32_t q = 0; q < kSubQueues; ++q) live += ctx->h_probe[32u * (1u + q)];
        if (want_timing) RT_HIP(ctx, drain());
        if (live == 0) break;
        grid = std::max(1u, std::min(blocks_for(live), max_grid));
        tgrid = std::max(1u, trace_blocks_for(live));
        if (fused && (solid || ctx->textured_tail) && live <= ctx->tail_paths) {
          // few paths left: all remaining bounces in one launch
          if (want_timing) RT_HIP(ctx, stamp());
          launch_trace(ctx, want_stats, general, true, tgrid, pb, ps, b + 1, p->max_depth, gs);
          if (want_timing) {
            RT_HIP(ctx, stamp());
            RT_HIP(ctx, stamp());
          }
          break;
        }
      }
    }
    resolve_kernel<<<blocks_for(n_pix), kBlock, 0, ctx->stream>>>(pb, ps, accum);
    if (getenv("MI355RT_DEBUG_LIVE")) {  // live paths after every bounce of this pass (a tail launch leaves its rows at zero)
      std::vector<uint32_t> rows((size_t)p->max_depth * kCounterStride);
      RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
      RT_HIP(ctx, hipMemcpy(rows.data(), pb.counters, rows.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
      fprintf(stderr, "mi355rt: pass of %u samples, %u paths; live after bounce b:", ps.samples, ps.n_paths);
      for (uint32_t b = 0; b < p->max_depth; ++b) {
        uint64_t live = 0;
        for (uint32_t q = 0; q < kSubQueues; ++q) live += rows[(size_t)b * kCounterStride + 32u * (1u + q)];
        fprintf(stderr, " %llu", (unsigned long long)live);
        if (live == 0) break;
      }
      fprintf(stderr, "\n");
    }
    if (report->progress_fn) {
      // a pass counts once it has finished: wait for it (only callers that asked pay for this)
      RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
      std::lock_guard<std::mutex> lock(report->progress_mutex);
      report->progress_done += n_pix * ps.samples;
      report->progress_fn(report->progress_user, report->progress_done, report->progress_total);
    }
  }
  RT_HIP(ctx, hipGetLastError());
  RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (want_stats)
    RT_HIP(ctx, hipMemcpy(stats, ctx->stats.p, RT_STAT_COUNT * sizeof(uint64_t),
                          hipMemcpyDeviceToHost));
  else if (want_timing)
    std::memset(stats, 0, RT_STAT_COUNT * sizeof(uint64_t));
  if (want_timing) {
    RT_HIP(ctx, drain());
    stats[RT_STAT_EXTEND_NS] = (uint64_t)(extend_ms * 1e6);
    stats[RT_STAT_SHADE_NS] = (uint64_t)(shade_ms * 1e6);
    stats[RT_STAT_EXTEND_LAUNCHES] = timed_launches;
    stats[RT_STAT_SHADE_LAUNCHES] = timed_launches;
  }
  return RT_OK;
}

}  // namespace

extern "C" {

static int create_single(int device, uint64_t max_paths_in_flight, rt_ctx** out) {
  if (!out) return RT_ERR_INVALID;
  *out = nullptr;
  rt_ctx* ctx = new (std::nothrow) rt_ctx();
  if (!ctx) return RT_ERR_NOMEM;
  ctx->device = device;
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0 || ctx->device < 0 || ctx->device >= count) {
    delete ctx;
    return RT_ERR_DEVICE;  // no usable HIP device: there is no other way to render
  }
  hipDeviceProp_t prop;
  // the address of this device's copy of the no-allocation scratch array, with the device current
  if (hipSetDevice(ctx->device) != hipSuccess ||
      hipGetDeviceProperties(&prop, ctx->device) != hipSuccess ||
      hipStreamCreate(&ctx->stream) != hipSuccess ||
      hipMalloc((void**)&ctx->stats.p, RT_STAT_COUNT * sizeof(unsigned long long)) != hipSuccess ||
      hipHostMalloc((void**)&ctx->h_probe, kCounterStride * sizeof(uint32_t)) != hipSuccess ||
      hipGetSymbolAddress((void**)&ctx->scratch_ptr, HIP_SYMBOL(scratch)) != hipSuccess ||
      ctx->scratch_ptr == nullptr) {
    (void)hipGetLastError();
    rt_destroy(ctx);  // frees whatever was created before the failure
    return RT_ERR_DEVICE;
  }
  ctx->n_cus = (uint32_t)prop.multiProcessorCount;
  if (max_paths_in_flight) {
    ctx->max_paths = max_paths_in_flight;
  } else if (const char* e = getenv("MI355RT_MAX_PATHS")) {
    ctx->max_paths = std::max<uint64_t>(strtoull(e, nullptr, 10), 1u << 16);
  } else {
    // default: up to 70 % of the free device memory at ~200 bytes of state and queues per path,
    // between 2^22 and 2^30 paths.  Fewer, larger passes spend less time in launch tails and in
    // nearly empty late bounces: a 1024-spp 1080p frame as 2 passes of 512 samples instead of 16 of
    // 64 renders 11 % faster, cornell at depth 500 16 % (the buffers are only allocated as large as
    // a render's window and sample count need).
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = 0;
    uint64_t fit = (uint64_t)((double)free_b * 0.7) / 200u;
    ctx->max_paths = std::min<uint64_t>(std::max<uint64_t>(fit, 1ull << 22), 1ull << 30);
    ctx->max_paths_auto = true;
  }
  // path indices and the grid-stride loops over them are 32-bit
  if (ctx->max_paths > (1ull << 31)) {
    rt_destroy(ctx);
    return RT_ERR_LIMIT;
  }
  *out = ctx;
  return RT_OK;
}


// RCCL is looked up at run time, and only for device groups: a single-GPU caller never loads it.
namespace {
struct Rccl {
  int (*CommInitAll)(void**, int, const int*) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*Reduce)(const void*, void*, size_t, int, int, int, void*, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
const int kNcclFloat = 7, kNcclSum = 0;  // ncclFloat32, ncclSum (rccl.h)
bool load_rccl(void* lib, Rccl& r) {
  r.CommInitAll = (decltype(r.CommInitAll))dlsym(lib, "ncclCommInitAll");
  r.CommDestroy = (decltype(r.CommDestroy))dlsym(lib, "ncclCommDestroy");
  r.Reduce = (decltype(r.Reduce))dlsym(lib, "ncclReduce");
  r.GroupStart = (decltype(r.GroupStart))dlsym(lib, "ncclGroupStart");
  r.GroupEnd = (decltype(r.GroupEnd))dlsym(lib, "ncclGroupEnd");
  r.GetErrorString = (decltype(r.GetErrorString))dlsym(lib, "ncclGetErrorString");
  return r.CommInitAll && r.CommDestroy && r.Reduce && r.GroupStart && r.GroupEnd && r.GetErrorString;
}
}  // namespace

int rt_create(const rt_config* cfg, rt_ctx** out) {
  if (!out) return RT_ERR_INVALID;
  *out = nullptr;
  const uint32_t n = cfg ? cfg->n_devices : 0u;
  if (n > 1 && !cfg->devices) return RT_ERR_INVALID;
  const uint64_t max_paths = cfg ? cfg->max_paths_in_flight : 0;
  rt_ctx* root = nullptr;
  int rc = create_single(n > 1 ? cfg->devices[0] : n == 1 && cfg->devices ? cfg->devices[0] : cfg ? cfg->device : 0,
                         max_paths, &root);
  // a group of one member with a one-rank communicator: the RCCL path on a machine with one GPU
  const char* force = getenv("MI355RT_FORCE_RCCL");
  const bool group_of_one = n == 1 && cfg->devices && force && atoi(force);
  if (rc != RT_OK || (n <= 1 && !group_of_one)) {
    *out = root;
    return rc;
  }
  bool distinct = true;
  for (uint32_t k = 1; k < n; ++k) {
    rt_ctx* peer = nullptr;
    rc = create_single(cfg->devices[k], max_paths, &peer);
    if (rc != RT_OK) {
      rt_destroy(root);
      return rc;
    }
    root->peers.push_back(peer);
    for (uint32_t j = 0; j < k; ++j) distinct = distinct && cfg->devices[j] != cfg->devices[k];
  }
  root->group = true;
  // One communicator per member, made in this one process (ncclCommInitAll).  A group that names
  // a device twice (two shares on one GPU: a rehearsal on a single-GPU machine) cannot have
  // one and adds its shares through host memory.  A group of distinct devices without a usable
  // librccl is an error, not a quiet detour through the host: MI355RT_STAGED_REDUCE=1 asks for
  // that detour in so many words.
  const char* staged = getenv("MI355RT_STAGED_REDUCE");
  root->reduce_path = RT_REDUCE_HOST_STAGED;
  if (distinct && !(staged && atoi(staged))) {
    auto give_up = [&](const std::string& why) {
      fprintf(stderr, "mi355rt: %s (set MI355RT_STAGED_REDUCE=1 to add the shares through host memory)\n",
              why.c_str());
      rt_destroy(root);
      return RT_ERR_DEVICE;
    };
    root->rccl = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!root->rccl) root->rccl = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!root->rccl) {
      const char* why = dlerror();
      return give_up(std::string("a device group needs librccl: ") + (why ? why : "not found"));
    }
    Rccl r;
    if (!load_rccl(root->rccl, r)) return give_up("librccl lacks one of the six entry points the group reduce uses");
    root->comms.assign(n, nullptr);
    int nrc = r.CommInitAll(root->comms.data(), (int)n, cfg->devices);
    if (nrc != 0) {
      root->comms.clear();
      return give_up(std::string("ncclCommInitAll: ") + r.GetErrorString(nrc));
    }
    root->reduce_path = RT_REDUCE_RCCL;
  }
  if (getenv("MI355RT_DEBUG"))
    fprintf(stderr, "mi355rt: device group of %u, shares added %s\n", n,
            root->reduce_path == RT_REDUCE_RCCL ? "by ncclReduce over the group's communicators" : "through host memory");
  *out = root;
  return RT_OK;
}

int rt_group_info(const rt_ctx* ctx, rt_group_report* out) {
  if (!ctx || !out) return RT_ERR_INVALID;
  out->n_members = (uint32_t)ctx->peers.size() + 1u;
  out->reduce_path = ctx->reduce_path;
  out->render_ms_min = ctx->last_render_ms_min;
  out->render_ms_max = ctx->last_render_ms_max;
  out->reduce_ms = ctx->last_reduce_ms;
  return RT_OK;
}

static int upload_flat(rt_ctx* ctx);

int rt_upload_scene(rt_ctx* ctx, const rt_scene* scene) {
  if (!ctx) return RT_ERR_INVALID;
  if (!scene) return fail(ctx, RT_ERR_INVALID, "null scene");
  RT_HIP(ctx, hipSetDevice(ctx->device));
  ctx->have_scene = false;
  int code = RT_OK;
  std::string err;
  // Large scenes get their hierarchy from the device (bvh_device.hip): the host's SAH build of a
  // million spheres takes longer than rendering them.  MI355RT_DEVICE_BUILD=0 / 1: never / always.
  bool device_build = scene->n_prims >= (1u << 17);
  if (const char* e = getenv("MI355RT_DEVICE_BUILD")) device_build = atoi(e) != 0;
  if (kWide != 4u) device_build = false;  // (the device's builder makes four-wide records)
  // Either device step may fail (out of memory, most likely); flatten_scene then does that step on
  // the host.  The runtime's error state is cleared here, or the next hipGetLastError() of the
  // upload would report the failure the fallback has just dealt with.
  // MI355RT_FAIL_DEVICE_BUILD (bit 0: the hierarchy, bit 1: the order) makes them fail, for tests.
  const char* fail_env = getenv("MI355RT_FAIL_DEVICE_BUILD");
  const int fail_steps = fail_env ? atoi(fail_env) : 0;
  const WideBuild wide = [&](const std::vector<rt_bvh_node>& leaf_boxes, std::vector<rt_bvh_node>& records,
                             std::vector<uint32_t>& leaf_order, uint32_t& depth) {
    std::string e = (fail_steps & 1) ? std::string("device build: failure requested")
                                     : device_build_wide(ctx->stream, leaf_boxes, records, leaf_order, depth);
    if (!e.empty()) (void)hipGetLastError();
    return e;
  };
  const RefOrderFn ref_order = [&](const std::vector<double>& centres, uint32_t n, std::vector<uint32_t>& order) {
    std::string e = (fail_steps & 2) ? std::string("device order: failure requested")
                                     : device_reference_order(ctx->stream, centres, n, order);
    if (!e.empty()) (void)hipGetLastError();
    return e;
  };
  try {
    auto flat = std::make_shared<FlatScene>();
    err = flatten_scene(*scene, *flat, code, device_build ? &wide : nullptr, device_build ? &ref_order : nullptr);
    if (err.empty()) ctx->flat_ptr = flat;
  } catch (const std::exception& e) {
    return fail(ctx, RT_ERR_NOMEM, e.what());
  }
  if (!err.empty()) return fail(ctx, code, err);
  // the members of a device group get the same flattened scene: built once, copied to all of
  // them side by side (one host thread per member, as in a render)
  std::vector<int> codes(ctx->peers.size(), RT_OK);
  std::vector<std::thread> threads;
  for (size_t k = 0; k < ctx->peers.size(); ++k) {
    ctx->peers[k]->flat_ptr = ctx->flat_ptr;
    threads.emplace_back([&, k]() { codes[k] = upload_flat(ctx->peers[k]); });
  }
  int rc = upload_flat(ctx);
  for (std::thread& t : threads) t.join();
  for (size_t k = 0; k < ctx->peers.size() && rc == RT_OK; ++k)
    if (codes[k] != RT_OK) {
      rc = codes[k];
      ctx->error = "device " + std::to_string(ctx->peers[k]->device) + ": " + ctx->peers[k]->error;
    }
  return rc;
}

// The device side of rt_upload_scene for one context: ctx->flat_ptr to device memory, LDS plans.
static int upload_flat(rt_ctx* ctx) {
  RT_HIP(ctx, hipSetDevice(ctx->device));
  ctx->have_scene = false;
  const FlatScene& f = ctx->flat();
  int rc;
  const auto upload_t0 = std::chrono::steady_clock::now();
  {
    // device copy of the records: the traversal entry code in place of `a`, and three
    // never-hit records at the end so that a visit may always read four children
    ctx->nodes.release();
    const size_t n_rec = f.nodes.size();
    RT_HIP(ctx, hipMalloc((void**)&ctx->nodes.p, (n_rec + kWide - 1u) * sizeof(rt_bvh_node)));
    ctx->nodes.n = 2 * (n_rec + kWide - 1u);
    if (n_rec)
      RT_HIP(ctx, hipMemcpyAsync(ctx->nodes.p, f.nodes.data(), n_rec * sizeof(rt_bvh_node), hipMemcpyHostToDevice,
                                 ctx->stream));
    device_records_kernel<<<blocks_for(n_rec + kWide - 1u), kBlock, 0, ctx->stream>>>((rt_bvh_node*)ctx->nodes.p,
                                                                            (uint32_t)n_rec);
    RT_HIP(ctx, hipGetLastError());
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  if ((rc = upload(ctx, ctx->prims, f.prims)) != RT_OK) return rc;
  if ((rc = upload(ctx, ctx->ranks, f.ranks)) != RT_OK) return rc;
  if ((rc = upload(ctx, ctx->boundary, f.boundary)) != RT_OK) return rc;
  if ((rc = upload(ctx, ctx->ops, f.ops)) != RT_OK) return rc;
  if ((rc = upload(ctx, ctx->aux, f.aux)) != RT_OK) return rc;
  if ((rc = upload(ctx, ctx->materials, f.materials)) != RT_OK) return rc;
  if ((rc = upload(ctx, ctx->textures, f.textures)) != RT_OK) return rc;
  if ((rc = upload(ctx, ctx->perlin, f.perlin)) != RT_OK) return rc;
  if ((rc = upload(ctx, ctx->images, f.images)) != RT_OK) return rc;
  if ((rc = upload(ctx, ctx->texels, f.texels)) != RT_OK) return rc;
  if ((rc = upload(ctx, ctx->chain_boxes, f.chain_boxes)) != RT_OK) return rc;
  if ((rc = upload(ctx, ctx->chain_ranges, f.chain_ranges)) != RT_OK) return rc;
  DeviceScene& ds = ctx->ds;
  ds.nodes = ctx->nodes.p;
  ds.prims = ctx->prims.p;
  ds.ranks = ctx->ranks.p;
  ds.boundary = ctx->boundary.p;
  ds.ops = ctx->ops.p;
  ds.aux = ctx->aux.p;
  ds.materials = ctx->materials.p;
  ds.textures = ctx->textures.p;
  ds.perlin = ctx->perlin.p;
  ds.images = ctx->images.p;
  ds.texels = ctx->texels.p;
  ds.chain_boxes = ctx->chain_boxes.p;
  ds.chain_ranges = ctx->chain_ranges.p;
  ds.n_nodes = (uint32_t)f.nodes.size();
  ds.has_media = f.has_media ? 1u : 0u;
  ds.has_lights = 0u;
  for (const DMaterial& m : f.materials)
    if (m.kind == RT_MAT_DIFFUSE_LIGHT) ds.has_lights = 1u;
  ds.root_code = f.nodes.empty() ? 1u : make_entry(f.nodes[0].a, f.nodes[0].b);
  ds.ops_lds_offset = 0u;
  ds.n_ops_lds = 0u;
  ds.cam = f.camera;
  std::memcpy(ds.bg, f.background, sizeof ds.bg);
  // LDS plans.  extend/first_hit: 256-thread workgroups, four to a CU next to their
  // 12.3 KB of staged queues.  trace: one plan per workgroup size it can run with.
  {
    uint32_t budget = 27u * 1024u;
    if (const char* e = getenv("MI355RT_LDS_BYTES")) budget = (uint32_t)strtoul(e, nullptr, 10);
    // LDS stack entries per lane; what a deeper hierarchy can need beyond them lives in global
    // memory.  33 keeps a 512-thread workgroup's stack within its half of the CU's LDS; a lower
    // cap buys LDS for more records but measured slower (book2: 41.6 ms at 16 against 39.5 ms).
    uint32_t stack_cap = 33u;
    if (const char* e = getenv("MI355RT_STACK_CAP")) stack_cap = (uint32_t)strtoul(e, nullptr, 10);
    // (every knob starts from its default at each upload: a variable that has been unset no longer applies)
    ctx->tail_paths = 1u << 20;
    ctx->textured_tail = ctx->tile_lists_on = ctx->stream_on = true;
    ctx->stream_min_paths = 1ull << 24;
    if (const char* e = getenv("MI355RT_TAIL_PATHS")) ctx->tail_paths = (uint32_t)strtoul(e, nullptr, 10);
    if (const char* e = getenv("MI355RT_TEXTURED_TAIL")) ctx->textured_tail = atoi(e) != 0;
    if (const char* e = getenv("MI355RT_TILE_LISTS")) ctx->tile_lists_on = atoi(e) != 0;
    if (const char* e = getenv("MI355RT_STREAM")) ctx->stream_on = atoi(e) != 0;
    if (const char* e = getenv("MI355RT_STREAM_MIN_PATHS")) ctx->stream_min_paths = strtoull(e, nullptr, 10);
    ctx->blocks_per_cu = 16u;
    if (const char* e = getenv("MI355RT_BLOCKS_PER_CU"))
      ctx->blocks_per_cu = std::max(1u, (uint32_t)strtoul(e, nullptr, 10));
    ctx->lds = plan_lds(f, kBlock, std::min(budget, 160u * 1024u - 13u * 1024u), stack_cap);
    if (const char* e = getenv("MI355RT_LDS_PRIMS")) {
      LdsLayout& l = ctx->lds;
      uint32_t n = std::min<uint32_t>(l.n_prims, (uint32_t)strtoul(e, nullptr, 10));
      l.bytes -= (l.n_prims - n) * kPrimHotBytes;
      l.n_prims = n;
    }
    // The general trace kernel (media, box chains) needs ~165 registers:
    // 256-thread workgroups, three to a CU (3 waves per SIMD); the plain one fits 128 registers
    // and runs 512-thread workgroups, two to a CU (4 waves per SIMD).
    bool solid = true;
    for (const DTexture& t : f.textures) solid = solid && t.kind == RT_TEX_SOLID;
    ctx->solid = solid;
    ctx->general = f.has_media || !f.chain_ranges.empty();
    uint32_t tb = ctx->general ? 256u : 512u;
    if (const char* e = getenv("MI355RT_TRACE_BLOCK")) tb = (uint32_t)strtoul(e, nullptr, 10);
    if (tb != 256u && tb != 512u)
      return fail(ctx, RT_ERR_INVALID, "MI355RT_TRACE_BLOCK must be 256 or 512");
    ctx->trace_block = tb;
    // plain spheres only, nothing deferred (book_cover.ron and its scale-ups): the sphere-only kernel
    bool spheres = !ctx->general && !f.defers && tb == 512u && !f.prims.empty();
    for (const DPrim& pr : f.prims)
      spheres = spheres && (pr.kind & 15u) == DP_SPHERE && ops_count_host(pr.ops) == 0u && !(pr.kind & kDeferBit);
    if (const char* e = getenv("MI355RT_SPHERES_KERNEL")) spheres = spheres && atoi(e) != 0;  // (A/B timing)
    ctx->spheres = spheres;
    const uint32_t per_cu = std::max(1u, (ctx->general ? 256u * RT_TRACE_MEDIA_WAVES : 1024u) / tb);  // resident
    // launched: as many workgroups as are resident together; the waves draw their work from a
    // counter, so nothing is gained by more
    ctx->trace_blocks_per_cu = per_cu;
    if (const char* e = getenv("MI355RT_TRACE_BLOCKS_PER_CU"))
      ctx->trace_blocks_per_cu = std::max(1u, (uint32_t)strtoul(e, nullptr, 10));
    // LDS of a CU split between the resident workgroups, minus the per-wave queue stages
    const uint32_t stage_bytes = (tb / 64u) * (kWaveStage + kDeferStage) * 4u;
    uint32_t tbudget = (160u * 1024u) / per_cu - stage_bytes - 1024u;
    if (const char* e = getenv("MI355RT_TRACE_LDS_BYTES")) tbudget = (uint32_t)strtoul(e, nullptr, 10);
    tbudget = std::min(tbudget, 160u * 1024u - stage_bytes - 512u);
    ctx->lds_trace = plan_lds(f, tb, tbudget, stack_cap, ctx->spheres ? 32u : kPrimHotBytes);
    // overflow area for the stack entries LDS does not hold: one column per thread of the
    // largest grid any traversal kernel is launched with
    ctx->stack_overflow.release();
    const uint32_t need = std::max(f.stack_entries + 1u, 2u);
    if (need > ctx->lds.stack_depth) {
      const uint32_t threads = ctx->n_cus * std::max(ctx->blocks_per_cu * (uint32_t)kBlock,
                                                     ctx->trace_blocks_per_cu * tb);
      const size_t words = (size_t)(need - ctx->lds.stack_depth) * threads;
      RT_HIP(ctx, hipMalloc((void**)&ctx->stack_overflow.p, words * sizeof(uint32_t)));
      ctx->stack_overflow.n = words;
      ctx->lds.overflow = ctx->lds_trace.overflow = ctx->stack_overflow.p;
      ctx->lds.overflow_stride = ctx->lds_trace.overflow_stride = threads;
    }
  }
  if (getenv("MI355RT_DEBUG"))
    fprintf(stderr, "mi355rt: copy to the device and LDS plans %.1f ms\n",
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - upload_t0).count());
  if (getenv("MI355RT_DEBUG"))
    fprintf(stderr, "mi355rt: %zu records, %zu leaf slots, stack %u; extend LDS %u B (%u records, %u slots); "
            "trace block %u LDS %u B (%u records, %u slots)\n", f.nodes.size(), f.prims.size(),
            f.stack_entries, ctx->lds.bytes, ctx->lds.n_nodes, ctx->lds.n_prims, ctx->trace_block,
            ctx->lds_trace.bytes, ctx->lds_trace.n_nodes, ctx->lds_trace.n_prims);
  if (ctx->lds.bytes > 32u * 1024u) RT_HIP(ctx, allow_extend_lds((int)ctx->lds.bytes));
  if (ctx->lds_trace.bytes > 32u * 1024u) {
    int bytes = (int)ctx->lds_trace.bytes;
    RT_HIP(ctx, ctx->spheres ? allow_trace_spheres_lds(bytes)
                : ctx->trace_block == 512 ? allow_trace_lds<512>(bytes) : allow_trace_lds<256>(bytes));
  }
  ctx->have_scene = true;
  return RT_OK;
}

int rt_render_f64(rt_ctx* ctx, const rt_render_params* p, double* rgb_sum, uint64_t* stats) {
  if (!ctx) return RT_ERR_INVALID;
  if (!rgb_sum) return fail(ctx, RT_ERR_INVALID, "null output");
  if (ctx->group)
    return fail(ctx, RT_ERR_UNSUPPORTED, "rt_render_f64: a device group adds float shares (use rt_render)");
  double* accum = nullptr;
  int rc = render_into_accum(ctx, p, stats, &accum);
  if (rc != RT_OK) return rc;
  PassInfo ps{};
  check_window(ctx, p->width, p->height, &p->window, ps);
  RT_HIP(ctx, hipMemcpy(rgb_sum, accum, (size_t)ps.n_pixels * 3 * sizeof(double),
                        hipMemcpyDeviceToHost));
  return RT_OK;
}

static int output_on_device(rt_ctx* ctx, const void* d_out) {
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, d_out) != hipSuccess ||
      attr.type != hipMemoryTypeDevice || attr.device != ctx->device) {
    (void)hipGetLastError();
    return fail(ctx, RT_ERR_INVALID, "output is not device memory of the context's device");
  }
  return RT_OK;
}


// A device group's render: member k of N renders samples [begin + n k / N, begin + n (k + 1) / N)
// of every pixel on its own device and stream, from its own host thread; the float shares are
// then added onto the root's device: one ncclReduce(sum) to rank 0 over the group's communicators
// (RCCL; between the GPUs of one node that is xGMI), or through host memory for a group
// without communicators.  d_out: window-sized float sums on the root's device.  Replaces the
// per-pixel `.sum()` over samples (src/main.rs:97) across devices: the only coupling there is.
static int render_group(rt_ctx* root, const rt_render_params* p, float* d_out, uint64_t* stats) {
  if (!p) return fail(root, RT_ERR_INVALID, "null params");
  if (p->sample_end < p->sample_begin) return fail(root, RT_ERR_INVALID, "empty sample range");
  std::vector<rt_ctx*> members{root};
  members.insert(members.end(), root->peers.begin(), root->peers.end());
  const uint32_t N = (uint32_t)members.size();
  PassInfo ps{};
  int rc = check_window(root, p->width, p->height, &p->window, ps);
  if (rc != RT_OK) return rc;
  const uint64_t n = (uint64_t)ps.n_pixels * 3;
  const uint64_t total = p->sample_end - p->sample_begin;
  std::vector<int> codes(N, RT_OK);
  std::vector<std::vector<uint64_t>> member_stats(N, std::vector<uint64_t>(stats ? RT_STAT_COUNT : 0));
  root->progress_done = 0;
  root->progress_total = (uint64_t)ps.n_pixels * total;
  struct ReportToRoot {  // every member, the root included, reports to the root while the group renders
    std::vector<rt_ctx*>& m;
    ReportToRoot(std::vector<rt_ctx*>& members, rt_ctx* to) : m(members) {
      for (rt_ctx* c : m) c->progress_to = to;
    }
    ~ReportToRoot() {
      for (rt_ctx* c : m) c->progress_to = nullptr;
    }
  } report_to_root(members, root);
  std::vector<double> member_ms(N, 0.0);
  auto work = [&](uint32_t k) {
    rt_ctx* m = members[k];
    auto run = [&]() -> int {
      const auto t0 = std::chrono::steady_clock::now();
      RT_HIP(m, hipSetDevice(m->device));
      if (m->share.n < n) {
        m->share.release();
        RT_HIP(m, hipMalloc((void**)&m->share.p, n * sizeof(float)));
        m->share.n = n;
      }
      rt_render_params pk = *p;
      pk.sample_begin = p->sample_begin + (uint32_t)(total * k / N);
      pk.sample_end = p->sample_begin + (uint32_t)(total * (k + 1) / N);
      double* accum = nullptr;
      int r = render_into_accum(m, &pk, stats ? member_stats[k].data() : nullptr, &accum);
      if (r != RT_OK) return r;
      to_float_kernel<<<blocks_for(n), kBlock, 0, m->stream>>>(accum, m->share.p, n);
      RT_HIP(m, hipGetLastError());
      RT_HIP(m, hipStreamSynchronize(m->stream));
      member_ms[k] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      return RT_OK;
    };
    codes[k] = run();
  };
  {
    std::vector<std::thread> threads;
    for (uint32_t k = 1; k < N; ++k) threads.emplace_back(work, k);
    work(0);
    for (std::thread& t : threads) t.join();
  }
  for (uint32_t k = 0; k < N; ++k)
    if (codes[k] != RT_OK)
      return fail(root, codes[k], "device " + std::to_string(members[k]->device) + ": " + members[k]->error);
  if (stats) {
    for (int i = 0; i < RT_STAT_COUNT; ++i) {
      stats[i] = 0;
      for (uint32_t k = 0; k < N; ++k) stats[i] += member_stats[k][i];
    }
  }
  root->last_render_ms_min = *std::min_element(member_ms.begin(), member_ms.end());
  root->last_render_ms_max = *std::max_element(member_ms.begin(), member_ms.end());
  const auto reduce_t0 = std::chrono::steady_clock::now();
  if (root->reduce_path == RT_REDUCE_RCCL) {
    Rccl r;
    if (root->comms.size() != N || !load_rccl(root->rccl, r))
      return fail(root, RT_ERR_DEVICE, "the group's RCCL communicators are gone");
    // One ncclReduce per member between ncclGroupStart and ncclGroupEnd (one thread drives every
    // device), each on its member's stream, i.e. behind that member's share.  The bracket is
    // closed on every path: an early return between the two would leave the group open.
    std::string what;
    int nrc = r.GroupStart();
    if (nrc != 0) return fail(root, RT_ERR_DEVICE, std::string("ncclGroupStart: ") + r.GetErrorString(nrc));
    for (uint32_t k = 0; k < N && what.empty(); ++k) {
      hipError_t e = hipSetDevice(members[k]->device);
      if (e != hipSuccess) {
        what = std::string("hipSetDevice: ") + hipGetErrorString(e);
        break;
      }
      nrc = r.Reduce(members[k]->share.p, k == 0 ? (void*)d_out : (void*)members[k]->share.p, n, kNcclFloat,
                     kNcclSum, 0, root->comms[k], members[k]->stream);
      if (nrc != 0) what = std::string("ncclReduce: ") + r.GetErrorString(nrc);
    }
    const int erc = r.GroupEnd();
    if (what.empty() && erc != 0) what = std::string("ncclGroupEnd: ") + r.GetErrorString(erc);
    for (uint32_t k = 0; k < N && what.empty(); ++k) {
      hipError_t e = hipSetDevice(members[k]->device);
      if (e == hipSuccess) e = hipStreamSynchronize(members[k]->stream);
      if (e != hipSuccess) what = std::string("after ncclReduce: ") + hipGetErrorString(e);
    }
    (void)hipSetDevice(root->device);
    if (!what.empty()) return fail(root, RT_ERR_DEVICE, what);
  } else {
    std::vector<float> sum(n), part(n);
    for (uint32_t k = 0; k < N; ++k) {
      RT_HIP(root, hipSetDevice(members[k]->device));
      RT_HIP(root, hipMemcpy(k == 0 ? sum.data() : part.data(), members[k]->share.p, n * sizeof(float),
                             hipMemcpyDeviceToHost));
      if (k)
        for (uint64_t i = 0; i < n; ++i) sum[i] += part[i];
    }
    RT_HIP(root, hipSetDevice(root->device));
    RT_HIP(root, hipMemcpy(d_out, sum.data(), n * sizeof(float), hipMemcpyHostToDevice));
  }
  root->last_reduce_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - reduce_t0).count();
  return RT_OK;
}

int rt_render_device(rt_ctx* ctx, const rt_render_params* p, float* d_rgb_sum, uint64_t* stats) {
  if (!ctx) return RT_ERR_INVALID;
  if (!d_rgb_sum) return fail(ctx, RT_ERR_INVALID, "null output");
  int rc = output_on_device(ctx, d_rgb_sum);
  if (rc != RT_OK) return rc;
  if (ctx->group) return render_group(ctx, p, d_rgb_sum, stats);
  double* accum = nullptr;
  rc = render_into_accum(ctx, p, stats, &accum);
  if (rc != RT_OK) return rc;
  PassInfo ps{};
  check_window(ctx, p->width, p->height, &p->window, ps);
  uint64_t n = (uint64_t)ps.n_pixels * 3;
  to_float_kernel<<<blocks_for(n), kBlock, 0, ctx->stream>>>(accum, d_rgb_sum, n);
  RT_HIP(ctx, hipGetLastError());
  RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return RT_OK;
}

int rt_set_progress(rt_ctx* ctx, rt_progress_fn fn, void* user) {
  if (!ctx) return RT_ERR_INVALID;
  std::lock_guard<std::mutex> lock(ctx->progress_mutex);
  ctx->progress_fn = fn;
  ctx->progress_user = fn ? user : nullptr;
  return RT_OK;
}

int rt_render_device_noalloc(rt_ctx* ctx, const rt_render_params* p, float* d_rgb_sum) {
  if (!ctx) return RT_ERR_INVALID;
  if (!d_rgb_sum) return fail(ctx, RT_ERR_INVALID, "null output");
  if (ctx->group) return fail(ctx, RT_ERR_UNSUPPORTED, "rt_render_device_noalloc: not for device groups");
  int rc = output_on_device(ctx, d_rgb_sum);
  if (rc != RT_OK) return rc;
  double* accum = nullptr;
  rc = render_into_accum(ctx, p, nullptr, &accum, true);
  if (rc != RT_OK) return rc;
  PassInfo ps{};
  check_window(ctx, p->width, p->height, &p->window, ps);
  uint64_t n = (uint64_t)ps.n_pixels * 3;
  to_float_kernel<<<blocks_for(n), kBlock, 0, ctx->stream>>>(accum, d_rgb_sum, n);
  RT_HIP(ctx, hipGetLastError());
  RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return RT_OK;
}

// Device-side staging for the entry points that hand results over in host buffers; kept by
// the context and grown on demand, so repeated calls do not allocate.
static int ensure_staging(rt_ctx* ctx, size_t bytes) {
  if (ctx->staging.n >= bytes) return RT_OK;
  ctx->staging.release();
  RT_HIP(ctx, hipMalloc((void**)&ctx->staging.p, bytes));
  ctx->staging.n = bytes;
  return RT_OK;
}

int rt_render(rt_ctx* ctx, const rt_render_params* p, float* rgb_sum, uint64_t* stats) {
  if (!ctx) return RT_ERR_INVALID;
  if (!rgb_sum) return fail(ctx, RT_ERR_INVALID, "null output");
  if (ctx->group) {
    if (!p) return fail(ctx, RT_ERR_INVALID, "null params");
    PassInfo gps{};
    int grc = check_window(ctx, p->width, p->height, &p->window, gps);
    if (grc != RT_OK) return grc;
    const uint64_t gn = (uint64_t)gps.n_pixels * 3;
    RT_HIP(ctx, hipSetDevice(ctx->device));
    if ((grc = ensure_staging(ctx, gn * sizeof(float))) != RT_OK) return grc;
    if ((grc = render_group(ctx, p, (float*)ctx->staging.p, stats)) != RT_OK) return grc;
    RT_HIP(ctx, hipMemcpy(rgb_sum, ctx->staging.p, gn * sizeof(float), hipMemcpyDeviceToHost));
    return RT_OK;
  }
  double* accum = nullptr;
  int rc = render_into_accum(ctx, p, stats, &accum);
  if (rc != RT_OK) return rc;
  PassInfo ps{};
  check_window(ctx, p->width, p->height, &p->window, ps);
  uint64_t n = (uint64_t)ps.n_pixels * 3;
  if ((rc = ensure_staging(ctx, n * sizeof(float))) != RT_OK) return rc;
  float* d_out = (float*)ctx->staging.p;
  to_float_kernel<<<blocks_for(n), kBlock, 0, ctx->stream>>>(accum, d_out, n);
  RT_HIP(ctx, hipMemcpyAsync(rgb_sum, d_out, n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return RT_OK;
}

int rt_first_hit(rt_ctx* ctx, uint32_t width, uint32_t height, const rt_window* window,
                 uint32_t* id, double* t, double* normal) {
  if (!ctx) return RT_ERR_INVALID;
  if (!id || !t || !normal) return fail(ctx, RT_ERR_INVALID, "null output");
  if (!ctx->have_scene) return fail(ctx, RT_ERR_INVALID, "no scene uploaded");
  RT_HIP(ctx, hipSetDevice(ctx->device));
  PassInfo ps{};
  int rc = check_window(ctx, width, height, window, ps);
  if (rc != RT_OK) return rc;
  size_t n = ps.n_pixels;
  // staging: normals (3 f64), t (f64), id (u32) per pixel, in that order (alignment)
  if ((rc = ensure_staging(ctx, n * (4 * sizeof(double) + sizeof(uint32_t)))) != RT_OK) return rc;
  double* d_n = (double*)ctx->staging.p;
  double* d_t = d_n + 3 * n;
  uint32_t* d_id = (uint32_t*)(d_t + n);
  const uint32_t fgrid = std::min(blocks_for(n), ctx->n_cus * ctx->blocks_per_cu);
  with_bool(!ctx->flat().chain_ranges.empty(), [&](auto C) {
    return with_bool(ctx->lds.overflow != nullptr, [&](auto O) {
      first_hit_kernel<decltype(C)::value, decltype(O)::value>
          <<<fgrid, kBlock, ctx->lds.bytes, ctx->stream>>>(ctx->ds, ctx->lds, ps, d_id, d_t, d_n);
      return 0;
    });
  });
  RT_HIP(ctx, hipGetLastError());
  // the three results follow the kernel on its stream; one wait for all of them
  RT_HIP(ctx, hipMemcpyAsync(id, d_id, n * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  RT_HIP(ctx, hipMemcpyAsync(t, d_t, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  RT_HIP(ctx, hipMemcpyAsync(normal, d_n, n * 3 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return RT_OK;
}

int rt_bvh_build(const rt_scene* scene, rt_bvh** out) {
  if (!scene || !out) return RT_ERR_INVALID;
  *out = nullptr;
  rt_bvh* b = new (std::nothrow) rt_bvh();
  if (!b) return RT_ERR_NOMEM;
  int code = RT_OK;
  std::string err;
  try {
    err = flatten_scene(*scene, b->flat, code);
  } catch (const std::exception&) {
    delete b;
    return RT_ERR_NOMEM;
  }
  if (!err.empty()) {
    delete b;
    return code;
  }
  *out = b;
  return RT_OK;
}

int rt_bvh_info(const rt_bvh* bvh, uint32_t* n_nodes, uint32_t* n_items, uint32_t* max_depth) {
  if (!bvh) return RT_ERR_INVALID;
  if (n_nodes) *n_nodes = (uint32_t)bvh->flat.nodes.size();
  if (n_items) *n_items = (uint32_t)bvh->flat.items.size();
  if (max_depth) *max_depth = bvh->flat.bvh_depth;
  return RT_OK;
}

int rt_bvh_copy(const rt_bvh* bvh, rt_bvh_node* nodes, rt_bvh_item* items) {
  if (!bvh || !nodes || !items) return RT_ERR_INVALID;
  if (!bvh->flat.nodes.empty())
    std::memcpy(nodes, bvh->flat.nodes.data(), bvh->flat.nodes.size() * sizeof(rt_bvh_node));
  if (!bvh->flat.items.empty())
    std::memcpy(items, bvh->flat.items.data(), bvh->flat.items.size() * sizeof(rt_bvh_item));
  return RT_OK;
}

void rt_bvh_free(rt_bvh* bvh) { delete bvh; }

const char* rt_last_error(const rt_ctx* ctx) { return ctx ? ctx->error.c_str() : "null context"; }

void rt_destroy(rt_ctx* ctx) {
  if (!ctx) return;
  if (!ctx->comms.empty()) {
    Rccl r;
    if (load_rccl(ctx->rccl, r))
      for (void* c : ctx->comms)
        if (c) (void)r.CommDestroy(c);
  }
  if (ctx->rccl) dlclose(ctx->rccl);
  for (rt_ctx* peer : ctx->peers) rt_destroy(peer);
  (void)hipSetDevice(ctx->device);
  ctx->share.release();
  free_paths(ctx);
  ctx->nodes.release();
  ctx->prims.release();
  ctx->boundary.release();
  ctx->ranks.release();
  ctx->ops.release();
  ctx->aux.release();
  ctx->materials.release();
  ctx->textures.release();
  ctx->perlin.release();
  ctx->images.release();
  ctx->texels.release();
  ctx->chain_boxes.release();
  ctx->stack_overflow.release();
  ctx->tile_lists.release();
  ctx->chain_ranges.release();
  ctx->accum.release();
  ctx->staging.release();
  ctx->stats.release();
  if (ctx->h_probe) (void)hipHostFree(ctx->h_probe);
  for (hipEvent_t e : ctx->events) (void)hipEventDestroy(e);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

}  // extern "C"

#ifdef RT_LANE_PROFILE
// development build only: read and clear the loop-trip counters
extern "C" int rt_debug_lane_trips(unsigned long long* out, int n) {
  unsigned long long h[2 * LP_COUNT];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_lane_trips), sizeof(h)) != hipSuccess) return -1;
  for (int i = 0; i < n && i < 2 * LP_COUNT; ++i) out[i] = h[i];
  std::memset(h, 0, sizeof(h));
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_lane_trips), h, sizeof(h)) != hipSuccess) return -1;
  return 2 * LP_COUNT;
}
#endif
