// Command-line driver: `raytracer <scene.ron>` renders to ./image.png like the
// reference's main() (src/main.rs:45-118), on the HIP render path.  Options
// override what the scene file fixes (size, spp, depth) and expose the seeds the
// reference leaves to thread_rng().
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mi355rt.h"

static void usage() {
  std::fprintf(stderr,
               "usage: raytracer <scene.ron|scene.rtscene> [--width N] [--height N] [--spp N]\n"
               "       [--max-depth N] [--seed N] [--scene-seed N] [--device N] [--out image.png]\n"
               "       [--assets DIR] [--freeze out.rtscene]\n");
}

int main(int argc, char** argv) {
  if (argc < 2) {
    usage();
    return 2;
  }
  std::string path, out = "image.png", freeze;
  rt_host_options opt{};
  opt.scene_seed = 1;
  uint64_t seed = 1;
  int device = 0;
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    auto next = [&](const char* what) -> const char* {
      if (i + 1 >= argc) {
        std::fprintf(stderr, "error: %s needs a value\n", what);
        std::exit(2);
      }
      return argv[++i];
    };
    if (a == "--width") opt.width = (uint32_t)std::strtoul(next("--width"), nullptr, 10);
    else if (a == "--height") opt.height = (uint32_t)std::strtoul(next("--height"), nullptr, 10);
    else if (a == "--spp") opt.samples_per_pixel = (uint32_t)std::strtoul(next("--spp"), nullptr, 10);
    else if (a == "--max-depth") opt.max_depth = (uint32_t)std::strtoul(next("--max-depth"), nullptr, 10);
    else if (a == "--seed") seed = std::strtoull(next("--seed"), nullptr, 10);
    else if (a == "--scene-seed") opt.scene_seed = std::strtoull(next("--scene-seed"), nullptr, 10);
    else if (a == "--device") device = std::atoi(next("--device"));
    else if (a == "--out") out = next("--out");
    else if (a == "--assets") rt_host_add_asset_dir(next("--assets"));
    else if (a == "--freeze") freeze = next("--freeze");
    else if (a == "-h" || a == "--help") {
      usage();
      return 0;
    } else if (path.empty()) path = a;
    else {
      usage();
      return 2;
    }
  }
  if (path.empty()) {
    usage();
    return 2;
  }
  rt_host_scene* scene = nullptr;
  int rc;
  bool frozen = path.size() > 8 && path.substr(path.size() - 8) == ".rtscene";
  rc = frozen ? rt_host_load_frozen(path.c_str(), &scene) : rt_host_load_ron(path.c_str(), &opt, &scene);
  if (rc != RT_OK) {
    std::fprintf(stderr, "Error: %s\n", rt_host_last_error());
    return 1;
  }
  const rt_scene* view = rt_host_scene_view(scene);
  if (frozen && (opt.width || opt.height)) {
    rc = rt_host_set_image(scene, opt.width ? opt.width : view->width,
                           opt.height ? opt.height : view->height);
    if (rc != RT_OK) {
      std::fprintf(stderr, "Error: %s\n", rt_host_last_error());
      return 1;
    }
  }
  if (!freeze.empty() && rt_host_save_frozen(scene, freeze.c_str()) != RT_OK) {
    std::fprintf(stderr, "Error: %s\n", rt_host_last_error());
    return 1;
  }
  const uint32_t w = view->width, h = view->height;
  const uint32_t spp = frozen && opt.samples_per_pixel ? opt.samples_per_pixel : view->samples_per_pixel;
  const uint32_t depth = frozen && opt.max_depth ? opt.max_depth : view->max_depth;

  rt_ctx* ctx = nullptr;
  rt_config cfg{};
  cfg.device = device;
  if (rt_create(&cfg, &ctx) != RT_OK) {
    std::fprintf(stderr, "Error: no usable HIP device %d\n", device);
    return 1;
  }
  if (rt_upload_scene(ctx, view) != RT_OK) {
    std::fprintf(stderr, "Error: %s\n", rt_last_error(ctx));
    return 1;
  }
  auto start = std::chrono::steady_clock::now();  // the reference's clock starts after loading
  std::vector<double> sum((size_t)w * h * 3);
  rt_render_params p{};
  p.width = w;
  p.height = h;
  p.sample_begin = 0;
  p.sample_end = spp;
  p.max_depth = depth;
  p.seed = seed;
  if (rt_render_f64(ctx, &p, sum.data(), nullptr) != RT_OK) {
    std::fprintf(stderr, "Error: %s\n", rt_last_error(ctx));
    return 1;
  }
  std::vector<uint8_t> rgb((size_t)w * h * 3);
  rt_host_resolve_srgb8(sum.data(), (uint64_t)w * h, spp, rgb.data());
  if (rt_host_write_png(out.c_str(), w, h, rgb.data()) != RT_OK) {
    std::fprintf(stderr, "Error: %s\n", rt_host_last_error());
    return 1;
  }
  auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(
                std::chrono::steady_clock::now() - start).count();
  char buf[64];
  rt_host_human_duration((uint64_t)ms, buf, sizeof buf);
  std::printf("Done in %s\n", buf);
  rt_destroy(ctx);
  rt_host_free(scene);
  return 0;
}
