// Bounds checks over an rt_scene: every index must land inside its array,
// because the render path follows them without looking.
#pragma once
#include <string>

#include "../../include/mi355rt.h"

namespace rt {

// Returns "" when the scene is well formed, else what is wrong.
inline std::string validate_scene(const rt_scene& s) {
  if (s.n_objects && !s.objects) return "objects pointer";
  if (s.n_prims && !s.prims) return "prims pointer";
  if (s.n_ops && !s.ops) return "ops pointer";
  if (s.n_materials && !s.materials) return "materials pointer";
  if (s.n_textures && !s.textures) return "textures pointer";
  if (s.n_perlin && !s.perlin) return "perlin pointer";
  if (s.n_images && !s.images) return "images pointer";
  if (s.n_texel_bytes && !s.texels) return "texels pointer";
  const size_t np = s.n_prims, no = s.n_ops, nm = s.n_materials, nt = s.n_textures;
  for (uint32_t i = 0; i < s.n_objects; ++i) {
    const rt_object& o = s.objects[i];
    if (o.prim_count == 0 || (size_t)o.first_prim + o.prim_count > np) return "object prim range";
    if ((size_t)o.first_op + o.op_count > no) return "object op range";
    if ((size_t)o.inner_first_op + o.inner_op_count > no) return "object inner op range";
    if (o.op_count > 255 || o.inner_op_count > 255) return "object op count (max 255)";
    if (o.is_medium && o.medium_material >= nm) return "medium material index";
    if (o.is_medium && s.materials[o.medium_material].kind != RT_MAT_ISOTROPIC)
      return "medium material kind";
  }
  for (uint32_t i = 0; i < s.n_prims; ++i) {
    const rt_prim& p = s.prims[i];
    if (p.kind > RT_PRIM_RECT) return "prim kind";
    if (p.axis > 2) return "prim axis";
    if (p.material >= nm) return "prim material index";
  }
  for (uint32_t i = 0; i < s.n_ops; ++i)
    if (s.ops[i].kind > RT_OP_ROTATE_Y) return "op kind";
  for (uint32_t i = 0; i < s.n_materials; ++i) {
    const rt_material& m = s.materials[i];
    if (m.kind >= RT_MAT_KINDS) return "material kind";
    bool textured = m.kind == RT_MAT_LAMBERTIAN || m.kind == RT_MAT_DIFFUSE_LIGHT ||
                    m.kind == RT_MAT_ISOTROPIC;
    if (textured && m.texture >= nt) return "material texture index";
  }
  for (uint32_t i = 0; i < s.n_textures; ++i) {
    const rt_texture& t = s.textures[i];
    if (t.kind > RT_TEX_IMAGE) return "texture kind";
    // children precede their parent, so a walk down a checkerboard always ends
    if (t.kind == RT_TEX_CHECKER && (t.a >= i || t.b >= i)) return "checker child index";
    if (t.kind == RT_TEX_PERLIN && t.a >= s.n_perlin) return "perlin index";
    if (t.kind == RT_TEX_IMAGE && t.a >= s.n_images) return "image index";
  }
  for (uint32_t i = 0; i < s.n_images; ++i) {
    const rt_image& im = s.images[i];
    if (im.width == 0 || im.height == 0) return "image size";
    if (im.offset + (uint64_t)im.width * im.height * 3 > s.n_texel_bytes) return "image texel range";
  }
  for (uint32_t i = 0; i < s.n_perlin; ++i)
    for (int k = 0; k < 256; ++k)
      if (s.perlin[i].perm_x[k] > 255 || s.perlin[i].perm_y[k] > 255 || s.perlin[i].perm_z[k] > 255)
        return "perlin permutation";
  return "";
}

}  // namespace rt
