// Owning storage behind an rt_scene view, shared by the loader, the frozen-file
// reader/writer and the procedural generator.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/mi355rt.h"

struct rt_host_scene {
  rt_camera_desc camera_desc{};
  rt_scene view{};
  std::vector<rt_object> objects;
  std::vector<rt_prim> prims;
  std::vector<rt_op> ops;
  std::vector<rt_material> materials;
  std::vector<rt_texture> textures;
  std::vector<rt_perlin> perlin;
  std::vector<rt_image> images;
  std::vector<uint8_t> texels;

  // Point the view's arrays at the vectors (call after any mutation).
  void refresh_view();
};

namespace rthost {

// Camera::build()...done() (src/camera.rs:52-86) with aspect = width / height
// (src/scene/loader.rs:66).  Throws std::runtime_error with the reference's
// `ensure!` messages.
rt_camera build_camera(const rt_camera_desc& d, uint32_t width, uint32_t height);

// Seeded stand-in for thread_rng() at load time.
struct LoadRng {
  uint64_t seed;
  uint64_t counter = 0;
  explicit LoadRng(uint64_t s) : seed(s) {}
  double next_f64();                      // Standard f64: [0,1)
  double range(double lo, double hi);     // gen_range(lo..hi), throws on an empty range
  uint64_t below(uint64_t n);             // gen_range(0..n)
};

// Perlin::new (src/texture/perlin.rs:17-36).
void make_perlin(LoadRng& rng, rt_perlin& out);

struct AssetDirs {
  static std::vector<std::string>& list();
};

// Reads a binary PPM (P6, maxval 255) into RGB8.
bool read_ppm(const std::string& path, uint32_t& w, uint32_t& h, std::vector<uint8_t>& rgb,
              std::string& err);

rt_host_scene* load_ron_text(const std::string& text, const std::string& base_dir,
                             const rt_host_options& opt);

}  // namespace rthost
