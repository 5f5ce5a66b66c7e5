// Records the device kernels read, and the host-side flattening of an rt_scene
// into them.  Layout rules: every record is a multiple of 16 bytes and is read
// with 16-byte vector loads; f64 parameters are stored exactly as the reference
// would compute them on the host, so the device repeats only per-ray arithmetic.
#pragma once
#include <cstdint>
#include <functional>
#include <string>
#include <vector>

#include "../../include/mi355rt.h"

namespace rt {

enum DPrimKind : uint32_t {
  DP_SPHERE = 0,
  DP_MOVING_SPHERE = 1,
  DP_RECT = 2,
  DP_MEDIUM = 3
};

// 64-byte primitive record: 48 bytes the traversal reads for every test (staged in
// LDS), then 16 bytes only shading and the rarer kinds need (read from global memory).
//   kind word: bits 0-3 DPrimKind, bits 4-5 rect axis, bits 8-10 material kind
//              (rt_material_kind) of `material`, bit 11 "texture needs (u,v)",
//              bit 12 "a hit counts only if the ray passes the reference's box chain"
//              (a cuboid side that sticks out of Cuboid::bounding_box, see FlatScene::chain_boxes),
//              bit 13 "shade hits on this record in the deferred kernel" (a texture that is not a
//              SolidColor: see FlatScene::defers), bits 15-31 the material index when it fits
//              (bit 14 set: it does not, read `material`): what a hit is shaded with is then
//              known from the 48 bytes the traversal has read anyway.
//   ops word : first op (low 24 bits) | op count (high 8 bits), outermost first.
//   sphere        p = {cx, cy, cz, radius, -}
//   moving sphere p = {c0x, c0y, c0z, radius, time0}, p5 = aux index; aux = {c1x, c1y, c1z, time1}
//   rect          p = {center[axis], min[d1], max[d1], min[d2], max[d2]}, p5 = chain (bit 12)
//                 (d1 = axis.next(), d2 = d1.next(); min/max as rect.rs:20-34 computes them)
//   medium        p = {neg_inv_density, first boundary prim, boundary prim count, chain (bit 12), -}
struct DPrim {
  double p[5];
  uint32_t kind;
  uint32_t ops;
  // --- the first 48 bytes end here ---
  double p5;
  uint32_t material;
  uint32_t object;
};
const uint32_t kPrimHotBytes = 48;
static_assert(sizeof(DPrim) == 64, "DPrim must be 64 bytes");

struct DOp {  // 32 bytes
  uint32_t kind;  // rt_op_kind
  uint32_t pad;
  double a[3];    // translate: offset; rotate_y: sin, cos
};
static_assert(sizeof(DOp) == 32, "DOp must be 32 bytes");

struct DAux {  // 32 bytes
  double v[4];
};

struct DMaterial {  // 48 bytes
  uint32_t kind;
  uint32_t texture;
  double albedo[3];  // textured kinds: the colour of `texture` when that is a SolidColor
  double param;
  uint32_t needs_uv;
  uint32_t textured;  // reads a texture that is not a SolidColor
};
static_assert(sizeof(DMaterial) == 48, "DMaterial must be 48 bytes");

struct DTexture {  // 48 bytes
  uint32_t kind;
  uint32_t a, b;
  uint32_t pad;
  double color[3];
  double scale;
};
static_assert(sizeof(DTexture) == 48, "DTexture must be 48 bytes");

struct DImage {  // 16 bytes
  uint32_t width, height;
  uint64_t offset;
};

// Children per interior record of the wide hierarchy: 4 as shipped; a build with -DRT_WIDE=8
// (every file of the library) walks eight-wide records instead, the host's builder only.
#ifndef RT_WIDE
#define RT_WIDE 4
#endif
const uint32_t kWide = RT_WIDE;
const uint32_t kWideBits = RT_WIDE == 8 ? 3u : 2u;
static_assert(RT_WIDE == 4 || RT_WIDE == 8, "RT_WIDE must be 4 or 8");
const uint32_t kMaxBvhDepth = 32;      // levels of the binary tree the wide one is collapsed from
const uint32_t kMaxLeafItems = 8;     // upper bound; the builder's default is lower
const uint32_t kNoPrim = 0xFFFFFFFFu;
const uint32_t kChainBit = 1u << 12;
const uint32_t kDeferBit = 1u << 13;
const uint32_t kMaterialElsewhereBit = 1u << 14;
const uint32_t kMaterialShift = 15;

// Everything the device needs, still in host memory.
struct FlatScene {
  std::vector<rt_bvh_node> nodes;
  std::vector<rt_bvh_item> items;   // one per leaf slot, same order as prims
  std::vector<DPrim> prims;         // leaf slots
  std::vector<uint32_t> ranks;      // tie-break rank per leaf slot
  std::vector<DPrim> boundary;      // boundary primitives of constant media
  std::vector<DOp> ops;
  std::vector<DAux> aux;
  std::vector<DMaterial> materials;
  std::vector<DTexture> textures;
  std::vector<rt_perlin> perlin;
  std::vector<DImage> images;
  std::vector<uint8_t> texels;
  // The reference's own hierarchy (bvh.rs:33-77) culls with Aabb::intersect at every node from
  // the root to the leaf list an object sits in.  For geometry inside its object's
  // bounding_box() that never removes a hit, so the device ignores it; the Y sides of a
  // Cuboid with size.x != size.z reach outside Cuboid::bounding_box (cuboid.rs:21-30 and
  // :45-47), and there the reference's result depends on those boxes.  Only when a scene has
  // such sides: chain_boxes holds {min xyz, max xyz} per box, chain_ranges {first, count}
  // per chain (root first), and the flagged records name their chain.
  std::vector<double> chain_boxes;
  std::vector<uint32_t> chain_ranges;
  rt_camera camera{};
  double background[3] = {0, 0, 0};
  uint32_t bvh_depth = 0;      // levels of the wide hierarchy
  uint32_t stack_entries = 1;  // traversal stack entries a ray can need
  bool has_media = false;
  // The fused kernel shades a wave's hits in place, every material kind present in the wave
  // one after the other.  Hits on materials with procedural or image textures are marked
  // (kDeferBit) and shaded by a kernel of their own over a compacted queue.
  bool defers = false;  // some record carries kDeferBit
};

// A builder of the wide hierarchy other than the host's: from one float box per leaf slot
// (rounded outwards; a, b unused) to the records (breadth-first, children consecutive, a leaf
// record naming one position of leaf_order), the leaf slots in leaf order and the number of
// levels below the root.  Returns "" or an error message.
typedef std::function<std::string(const std::vector<rt_bvh_node>& leaf_boxes, std::vector<rt_bvh_node>& records,
                                  std::vector<uint32_t>& leaf_order, uint32_t& depth)>
    WideBuild;

// The reference's resolution order of the objects by other means than the host's sorts: from
// the centres of the objects' reference boxes (x of every object, then y, then z) to the
// object ids in that order.  Returns "" or an error message.
typedef std::function<std::string(const std::vector<double>& centres, uint32_t n, std::vector<uint32_t>& order)>
    RefOrderFn;

// Flatten + build.  Returns "" on success, else an error message; `code`
// receives the rt_status to report.  `wide`: builds the hierarchy instead of the host's SAH
// build (null: the host's).
std::string flatten_scene(const rt_scene& scene, FlatScene& out, int& code, const WideBuild* wide = nullptr,
                          const RefOrderFn* ref_order = nullptr);

}  // namespace rt
