// Host-side code under AddressSanitizer / UBSan: every scene file of a directory through the RON
// parser and loader, the frozen-scene round trip, the JPEG decoder on every .jpg there (whole and
// truncated at many lengths), the flatten + BVH build, resolve and the PNG writer.
// build: see tools/host_sanitize.sh.  usage: host_sanitize <scene dir> <tmp dir>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dirent.h>
#include <string>
#include <vector>

#include "../include/mi355rt.h"
#include "../raytracer_amd/csrc/device_scene.h"

static std::vector<uint8_t> slurp(const std::string& p) {
  std::vector<uint8_t> v;
  FILE* f = fopen(p.c_str(), "rb");
  if (!f) return v;
  fseek(f, 0, SEEK_END);
  long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  v.resize((size_t)n);
  if (fread(v.data(), 1, v.size(), f) != v.size()) v.clear();
  fclose(f);
  return v;
}

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  const std::string dir = argv[1], tmp = argv[2];
  rt_host_add_asset_dir(dir.c_str());
  DIR* d = opendir(dir.c_str());
  if (!d) return 2;
  int scenes = 0, jpegs = 0, failures = 0;
  while (dirent* e = readdir(d)) {
    const std::string name = e->d_name, path = dir + "/" + name;
    if (name.size() > 4 && name.substr(name.size() - 4) == ".ron") {
      rt_host_options opt{};
      opt.scene_seed = 1;
      rt_host_scene* s = nullptr;
      if (rt_host_load_ron(path.c_str(), &opt, &s) != RT_OK) {
        printf("%s: %s\n", name.c_str(), rt_host_last_error());
        ++failures;
        continue;
      }
      ++scenes;
      const std::string frozen = tmp + "/" + name + ".rtscene";
      rt_host_scene* again = nullptr;
      if (rt_host_save_frozen(s, frozen.c_str()) != RT_OK || rt_host_load_frozen(frozen.c_str(), &again) != RT_OK) {
        printf("%s: frozen round trip: %s\n", name.c_str(), rt_host_last_error());
        ++failures;
      }
      rt_host_set_image(s, 64, 36);
      rt::FlatScene flat;
      int code = RT_OK;
      const std::string err = rt::flatten_scene(*rt_host_scene_view(s), flat, code);
      if (!err.empty()) {
        printf("%s: flatten: %s\n", name.c_str(), err.c_str());
        ++failures;
      }
      // truncated copies of the scene text must fail cleanly
      std::vector<uint8_t> text = slurp(path);
      for (size_t cut = 0; cut < text.size(); cut += 1 + text.size() / 40) {
        std::string part((const char*)text.data(), cut);
        rt_host_scene* t = nullptr;
        if (rt_host_load_ron_text(part.c_str(), dir.c_str(), &opt, &t) == RT_OK) rt_host_free(t);
      }
      if (again) rt_host_free(again);
      rt_host_free(s);
    } else if (name.size() > 4 && name.substr(name.size() - 4) == ".jpg") {
      std::vector<uint8_t> data = slurp(path);
      uint32_t w = 0, h = 0;
      std::vector<uint8_t> rgb(16u << 20);
      if (rt_host_decode_jpeg(data.data(), data.size(), &w, &h, rgb.data(), rgb.size()) != RT_OK) {
        printf("%s: %s\n", name.c_str(), rt_host_last_error());
        ++failures;
      } else {
        ++jpegs;
        const std::string png = tmp + "/" + name + ".png";
        if (rt_host_write_png(png.c_str(), w, h, rgb.data()) != RT_OK) ++failures;
      }
      for (size_t cut = 0; cut < data.size(); cut += 1 + data.size() / 200)  // truncated files
        rt_host_decode_jpeg(data.data(), cut, &w, &h, rgb.data(), rgb.size());
      std::vector<uint8_t> bad = data;  // corrupted bytes
      for (size_t k = 0; k < 200; ++k) {
        bad = data;
        bad[(k * 7919u) % bad.size()] ^= (uint8_t)(1u << (k % 8));
        bad[(k * 104729u) % bad.size()] = (uint8_t)k;
        rt_host_decode_jpeg(bad.data(), bad.size(), &w, &h, rgb.data(), rgb.size());
      }
    }
  }
  closedir(d);
  std::vector<double> sums(64 * 36 * 3, 12.5);
  std::vector<uint8_t> out(64 * 36 * 3);
  rt_host_resolve_srgb8(sums.data(), 64 * 36, 16, out.data());
  char buf[64];
  rt_host_human_duration(3723004, buf, sizeof buf);
  printf("scenes %d, jpegs %d, failures %d, duration text '%s'\n", scenes, jpegs, failures, buf);
  return failures ? 1 : 0;
}
