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

#include <unistd.h>

#include "../../include/mi355rt.h"

// The reference's progress line (src/main.rs:58-61: "{bar} {human_pos}/{human_len} ({percent}%)
// {elapsed_precise}" over the pixel count), redrawn after every finished pass of the render.
struct Bar {
  uint64_t pixels;
  std::chrono::steady_clock::time_point start;
};
static std::string with_commas(uint64_t v) {
  std::string digits = std::to_string(v), out;
  for (size_t i = 0; i < digits.size(); ++i) {
    if (i && (digits.size() - i) % 3 == 0) out += ',';
    out += digits[i];
  }
  return out;
}
static void draw_bar(void* user, uint64_t done, uint64_t total) {
  const Bar* bar = (const Bar*)user;
  const double part = total ? (double)done / (double)total : 1.0;
  const int cells = 20, full = (int)(part * cells);
  std::string line;
  for (int i = 0; i < cells; ++i) line += i < full ? "\xe2\x96\x88" : "\xe2\x96\x91";
  const uint64_t secs = (uint64_t)std::chrono::duration_cast<std::chrono::seconds>(
                            std::chrono::steady_clock::now() - bar->start).count();
  std::fprintf(stderr, "\r%s %s/%s (%d%%) %02llu:%02llu:%02llu", line.c_str(),
               with_commas((uint64_t)(part * (double)bar->pixels)).c_str(), with_commas(bar->pixels).c_str(),
               (int)(part * 100.0), (unsigned long long)(secs / 3600), (unsigned long long)(secs / 60 % 60),
               (unsigned long long)(secs % 60));
  std::fflush(stderr);
}

static void usage() {
  std::fprintf(stderr,
               "usage: raytracer <scene.ron|scene.rtscene> [--width N] [--height N] [--spp N]\n"
               "       [--max-depth N] [--seed N] [--scene-seed N] [--device N] [--out image.png]\n"
               "       [--assets DIR] [--freeze out.rtscene] [--gpus N | --devices a,b,...]\n"
               "       [--dump-pfm linear.pfm] [--first-hit hits.rtfh] [--progress | --no-progress]\n"
               "  --gpus N         render on devices 0..N-1 (sample ranges, one reduce onto device 0)\n"
               "  --progress       draw the progress line on stderr after every pass (default: when stderr\n"
               "                   is a terminal)\n"
               "  --dump-pfm F     also write the linear per-pixel mean radiance (before the clamp and the\n"
               "                   8-bit resolve) as a little-endian RGB PFM\n"
               "  --first-hit F    also write the pinhole pixel-centre first-hit pass: 'RTFH', width, height\n"
               "                   (u32), then ids (u32), t (f64) and normals (3 f64) per pixel, row 0 on top\n");
}

int main(int argc, char** argv) {
  if (argc < 2) {
    usage();
    return 2;
  }
  std::string path, out = "image.png", freeze, pfm, first_hit;
  std::vector<int32_t> devices;
  rt_host_options opt{};
  opt.scene_seed = 1;
  uint64_t seed = 1;
  int device = 0;
  bool progress = isatty(fileno(stderr)) != 0;
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
    else if (a == "--dump-pfm") pfm = next("--dump-pfm");
    else if (a == "--first-hit") first_hit = next("--first-hit");
    else if (a == "--progress") progress = true;
    else if (a == "--no-progress") progress = false;
    else if (a == "--gpus") {
      char* end = nullptr;
      const char* v = next("--gpus");
      long n = std::strtol(v, &end, 10);
      if (end == v || *end || n < 1 || n > 64) {
        std::fprintf(stderr, "error: --gpus wants a count from 1 to 64, not '%s'\n", v);
        return 2;
      }
      devices.clear();
      for (long k = 0; k < n; ++k) devices.push_back((int32_t)k);
    } else if (a == "--devices") {
      devices.clear();
      const char* v = next("--devices");
      for (const char* c = v; *c;) {
        char* end = nullptr;
        long d = std::strtol(c, &end, 10);
        if (end == c || d < 0 || (*end && *end != ',') || (*end == ',' && !end[1])) {
          std::fprintf(stderr, "error: --devices wants ordinals separated by commas, not '%s'\n", v);
          return 2;
        }
        devices.push_back((int32_t)d);
        c = *end == ',' ? end + 1 : end;
      }
      if (devices.empty()) {
        std::fprintf(stderr, "error: --devices wants at least one ordinal\n");
        return 2;
      }
    }
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
  cfg.device = devices.empty() ? device : devices[0];
  cfg.n_devices = (uint32_t)devices.size();
  cfg.devices = devices.empty() ? nullptr : devices.data();
  if (rt_create(&cfg, &ctx) != RT_OK) {
    std::fprintf(stderr, "Error: no usable HIP device %d (or device group)\n", cfg.device);
    return 1;
  }
  if (rt_upload_scene(ctx, view) != RT_OK) {
    std::fprintf(stderr, "Error: %s\n", rt_last_error(ctx));
    return 1;
  }
  rt_group_report group{};
  rt_group_info(ctx, &group);
  if (group.reduce_path != RT_REDUCE_NONE)
    std::fprintf(stderr, "%u devices, shares added %s\n", group.n_members,
                 group.reduce_path == RT_REDUCE_RCCL ? "by one RCCL reduce" : "through host memory");
  auto start = std::chrono::steady_clock::now();  // the reference's clock starts after loading
  std::vector<double> sum((size_t)w * h * 3);
  rt_render_params p{};
  p.width = w;
  p.height = h;
  p.sample_begin = 0;
  p.sample_end = spp;
  p.max_depth = depth;
  p.seed = seed;
  Bar bar{(uint64_t)w * h, start};
  if (progress) {
    rt_set_progress(ctx, draw_bar, &bar);
    draw_bar(&bar, 0, 1);
  }
  if (group.reduce_path != RT_REDUCE_NONE) {  // a device group adds float shares
    std::vector<float> fsum(sum.size());
    if (rt_render(ctx, &p, fsum.data(), nullptr) != RT_OK) {
      std::fprintf(stderr, "Error: %s\n", rt_last_error(ctx));
      return 1;
    }
    for (size_t i = 0; i < sum.size(); ++i) sum[i] = (double)fsum[i];
  } else if (rt_render_f64(ctx, &p, sum.data(), nullptr) != RT_OK) {
    std::fprintf(stderr, "Error: %s\n", rt_last_error(ctx));
    return 1;
  }
  if (progress) {
    rt_set_progress(ctx, nullptr, nullptr);
    std::fputc('\n', stderr);
  }
  if (group.reduce_path != RT_REDUCE_NONE && rt_group_info(ctx, &group) == RT_OK)
    std::fprintf(stderr, "members rendered %.1f to %.1f ms, reduce %.1f ms\n", group.render_ms_min,
                 group.render_ms_max, group.reduce_ms);
  if (!pfm.empty()) {  // PFM rows run bottom to top; a negative scale means little-endian
    FILE* f = std::fopen(pfm.c_str(), "wb");
    if (!f) {
      std::fprintf(stderr, "Error: cannot write %s\n", pfm.c_str());
      return 1;
    }
    std::fprintf(f, "PF\n%u %u\n-1.0\n", w, h);
    std::vector<float> row((size_t)w * 3);
    for (uint32_t y = h; y-- > 0;) {
      for (size_t i = 0; i < row.size(); ++i) row[i] = (float)(sum[(size_t)y * w * 3 + i] / (double)spp);
      std::fwrite(row.data(), sizeof(float), row.size(), f);
    }
    std::fclose(f);
  }
  if (!first_hit.empty()) {
    std::vector<uint32_t> id((size_t)w * h);
    std::vector<double> t((size_t)w * h), normal((size_t)w * h * 3);
    if (rt_first_hit(ctx, w, h, nullptr, id.data(), t.data(), normal.data()) != RT_OK) {
      std::fprintf(stderr, "Error: %s\n", rt_last_error(ctx));
      return 1;
    }
    FILE* f = std::fopen(first_hit.c_str(), "wb");
    if (!f) {
      std::fprintf(stderr, "Error: cannot write %s\n", first_hit.c_str());
      return 1;
    }
    std::fwrite("RTFH", 1, 4, f);
    std::fwrite(&w, 4, 1, f);
    std::fwrite(&h, 4, 1, f);
    std::fwrite(id.data(), 4, id.size(), f);
    std::fwrite(t.data(), 8, t.size(), f);
    std::fwrite(normal.data(), 8, normal.size(), f);
    std::fclose(f);
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
