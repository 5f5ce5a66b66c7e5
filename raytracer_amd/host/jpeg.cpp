#include "jpeg.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace rt {
namespace {

const uint8_t kZigZag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,
                             12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
                             35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
                             58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Huffman {
  bool present = false;
  // canonical code: for each length 1..16 the first code, the last code and the index of its
  // first symbol
  int32_t min_code[17], max_code[17], first_symbol[17];
  uint8_t symbols[256];
};

struct Component {
  uint8_t id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
  int32_t dc_pred = 0;
  uint32_t blocks_w = 0, blocks_h = 0;  // in 8x8 blocks, padded to whole MCUs
  std::vector<uint8_t> plane;           // blocks_w*8 x blocks_h*8 samples
};

struct BitReader {
  const uint8_t* p;
  const uint8_t* end;
  uint32_t acc = 0;
  int bits = 0;
  bool hit_marker = false;

  void fill() {
    while (bits <= 24) {
      uint32_t byte = 0;
      if (!hit_marker && p < end) {
        if (p[0] == 0xFF) {
          if (p + 1 < end && p[1] == 0x00) {
            byte = 0xFF;
            p += 2;
          } else {
            hit_marker = true;  // a marker: feed zeros, the caller deals with it
          }
        } else {
          byte = *p++;
        }
      }
      acc |= byte << (24 - bits);
      bits += 8;
    }
  }
  int get(int n) {  // n <= 16
    if (n == 0) return 0;
    if (bits < n) fill();
    int v = (int)(acc >> (32 - n));
    acc <<= n;
    bits -= n;
    return v;
  }
  void reset() {
    acc = 0;
    bits = 0;
    hit_marker = false;
  }
};

int decode_symbol(BitReader& br, const Huffman& h, bool& ok) {
  int32_t code = 0;
  for (int len = 1; len <= 16; ++len) {
    code = (code << 1) | br.get(1);
    if (h.max_code[len] >= 0 && code <= h.max_code[len] && code >= h.min_code[len])
      return h.symbols[h.first_symbol[len] + (code - h.min_code[len])];
  }
  ok = false;
  return 0;
}

int extend(int v, int n) { return n && v < (1 << (n - 1)) ? v - (1 << n) + 1 : v; }

// 8x8 inverse DCT in double precision, level shift, clamp.
void idct_block(const int32_t* coef, uint8_t* out, size_t stride) {
  static double basis[8][8];
  static bool ready = false;
  if (!ready) {
    for (int x = 0; x < 8; ++x)
      for (int u = 0; u < 8; ++u)
        basis[x][u] = (u == 0 ? std::sqrt(0.125) : 0.5) * std::cos((2 * x + 1) * u * M_PI / 16.0);
    ready = true;
  }
  double tmp[64];
  for (int y = 0; y < 8; ++y)  // rows: over u
    for (int x = 0; x < 8; ++x) {
      double s = 0.0;
      for (int u = 0; u < 8; ++u) s += basis[x][u] * coef[y * 8 + u];
      tmp[y * 8 + x] = s;
    }
  for (int x = 0; x < 8; ++x)  // columns: over v
    for (int y = 0; y < 8; ++y) {
      double s = 0.0;
      for (int v = 0; v < 8; ++v) s += basis[y][v] * tmp[v * 8 + x];
      long r = std::lround(s) + 128;
      out[y * stride + x] = (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r);
    }
}

uint16_t be16(const uint8_t* p) { return (uint16_t)((p[0] << 8) | p[1]); }

// A component plane at full resolution.  Chroma at half resolution is interpolated with the
// triangle filter libjpeg calls "fancy upsampling" (3/4 nearest + 1/4 next, per axis); other
// ratios are replicated.
std::vector<uint8_t> full_resolution(const Component& c, uint32_t hmax, uint32_t vmax, uint32_t width,
                                     uint32_t height) {
  const size_t stride = (size_t)c.blocks_w * 8;
  const uint32_t cw = (width * c.h + hmax - 1) / hmax, ch = (height * c.v + vmax - 1) / vmax;
  std::vector<uint8_t> out((size_t)width * height);
  auto in = [&](uint32_t x, uint32_t y) -> int { return c.plane[(size_t)y * stride + x]; };
  const bool h2 = hmax == 2u * c.h, v2 = vmax == 2u * c.v, h1 = hmax == c.h, v1 = vmax == c.v;
  if (h1 && v1) {
    for (uint32_t y = 0; y < height; ++y)
      std::memcpy(out.data() + (size_t)y * width, c.plane.data() + (size_t)y * stride, width);
  } else if (h2 && (v1 || v2)) {
    std::vector<int> col(cw);
    for (uint32_t y = 0; y < height; ++y) {
      // vertical part: 3 * near row + far row (v2), or 4 * row (v1)
      uint32_t near = v2 ? y / 2 : y;
      uint32_t far = near;
      if (v2) far = (y & 1) ? std::min(near + 1, ch - 1) : (near ? near - 1 : 0);
      for (uint32_t x = 0; x < cw; ++x) col[x] = v2 ? 3 * in(x, near) + in(x, far) : 4 * in(x, near);
      uint8_t* row = out.data() + (size_t)y * width;
      for (uint32_t x = 0; x < width; ++x) {
        uint32_t i = x / 2;
        int v;
        if (x & 1)
          v = i + 1 < cw ? (3 * col[i] + col[i + 1] + 7) >> 4 : (4 * col[i] + 7) >> 4;
        else
          v = i ? (3 * col[i] + col[i - 1] + 8) >> 4 : (4 * col[i] + 8) >> 4;
        row[x] = (uint8_t)v;
      }
    }
  } else {
    for (uint32_t y = 0; y < height; ++y)
      for (uint32_t x = 0; x < width; ++x)
        out[(size_t)y * width + x] = (uint8_t)in((uint32_t)((size_t)x * c.h / hmax), (uint32_t)((size_t)y * c.v / vmax));
  }
  return out;
}

}  // namespace

bool decode_jpeg(const uint8_t* data, size_t size, uint32_t& width, uint32_t& height,
                 std::vector<uint8_t>& rgb, std::string& err) {
  auto fail = [&](const char* what) {
    err = std::string("JPEG: ") + what;
    return false;
  };
  if (size < 4 || data[0] != 0xFF || data[1] != 0xD8) return fail("no SOI marker");
  uint16_t quant[4][64];
  bool have_quant[4] = {false, false, false, false};
  Huffman dc[4], ac[4];
  std::vector<Component> comps;
  uint32_t restart_interval = 0;
  bool have_frame = false, adobe_rgb = false;
  uint32_t hmax = 1, vmax = 1;
  size_t pos = 2;
  for (;;) {
    while (pos < size && data[pos] != 0xFF) ++pos;  // tolerate padding
    while (pos < size && data[pos] == 0xFF) ++pos;
    if (pos >= size) return fail("no scan found");
    uint8_t marker = data[pos++];
    if (marker == 0xD9) return fail("no scan found");
    if (marker == 0x01 || (marker >= 0xD0 && marker <= 0xD7)) continue;
    if (pos + 2 > size) return fail("truncated segment");
    size_t len = be16(data + pos);
    if (len < 2 || pos + len > size) return fail("truncated segment");
    const uint8_t* seg = data + pos + 2;
    size_t n = len - 2;
    pos += len;
    if (marker == 0xDB) {  // DQT
      size_t i = 0;
      while (i < n) {
        int pq = seg[i] >> 4, tq = seg[i] & 15;
        ++i;
        if (tq > 3 || pq > 1 || i + (pq ? 128 : 64) > n) return fail("bad quantisation table");
        for (int k = 0; k < 64; ++k) {
          quant[tq][kZigZag[k]] = pq ? be16(seg + i + 2 * k) : seg[i + k];
        }
        i += pq ? 128 : 64;
        have_quant[tq] = true;
      }
    } else if (marker == 0xC4) {  // DHT
      size_t i = 0;
      while (i < n) {
        if (i + 17 > n) return fail("bad Huffman table");
        int tc = seg[i] >> 4, th = seg[i] & 15;
        if (tc > 1 || th > 3) return fail("bad Huffman table");
        Huffman& h = tc ? ac[th] : dc[th];
        const uint8_t* counts = seg + i + 1;
        size_t total = 0;
        for (int l = 0; l < 16; ++l) total += counts[l];
        if (total > 256 || i + 17 + total > n) return fail("bad Huffman table");
        std::memcpy(h.symbols, seg + i + 17, total);
        int32_t code = 0;
        int32_t k = 0;
        for (int l = 1; l <= 16; ++l) {
          h.first_symbol[l] = k;
          h.min_code[l] = code;
          code += counts[l - 1];
          k += counts[l - 1];
          h.max_code[l] = counts[l - 1] ? code - 1 : -1;
          code <<= 1;
        }
        h.present = true;
        i += 17 + total;
      }
    } else if (marker == 0xC0 || marker == 0xC1) {  // SOF0 / SOF1 (Huffman, sequential)
      if (n < 6) return fail("bad frame header");
      if (seg[0] != 8) return fail("only 8-bit samples are supported");
      height = be16(seg + 1);
      width = be16(seg + 3);
      int nc = seg[5];
      if (width == 0 || height == 0) return fail("empty image");
      if ((nc != 1 && nc != 3) || n < 6 + 3 * (size_t)nc) return fail("only grey and 3-component images are supported");
      comps.resize(nc);
      for (int c = 0; c < nc; ++c) {
        comps[c].id = seg[6 + 3 * c];
        comps[c].h = seg[7 + 3 * c] >> 4;
        comps[c].v = seg[7 + 3 * c] & 15;
        comps[c].tq = seg[8 + 3 * c];
        if (comps[c].h < 1 || comps[c].h > 4 || comps[c].v < 1 || comps[c].v > 4 || comps[c].tq > 3)
          return fail("bad component");
        hmax = std::max<uint32_t>(hmax, comps[c].h);
        vmax = std::max<uint32_t>(vmax, comps[c].v);
      }
      have_frame = true;
    } else if (marker == 0xC2 || (marker >= 0xC5 && marker <= 0xCF && marker != 0xC8 && marker != 0xCC)) {
      return fail("progressive, lossless and arithmetic-coded files are not supported");
    } else if (marker == 0xDD) {
      if (n < 2) return fail("bad restart interval");
      restart_interval = be16(seg);
    } else if (marker == 0xEE) {  // Adobe: transform 0 with three components means RGB
      if (n >= 12 && std::memcmp(seg, "Adobe", 5) == 0 && seg[11] == 0) adobe_rgb = true;
    } else if (marker == 0xDA) {  // SOS: the one scan of a baseline file
      if (!have_frame) return fail("scan before frame header");
      if (n < 1 || seg[0] != comps.size() || n < 1 + 2 * (size_t)seg[0] + 3)
        return fail("only single-scan (interleaved) files are supported");
      for (size_t c = 0; c < comps.size(); ++c) {
        uint8_t id = seg[1 + 2 * c];
        if (id != comps[c].id) return fail("scan components out of order");
        comps[c].td = seg[2 + 2 * c] >> 4;
        comps[c].ta = seg[2 + 2 * c] & 15;
        if (comps[c].td > 3 || comps[c].ta > 3 || !dc[comps[c].td].present || !ac[comps[c].ta].present ||
            !have_quant[comps[c].tq])
          return fail("scan refers to a missing table");
      }
      break;
    }
    // every other segment (APPn, COM, ...) is skipped
  }

  const uint32_t mcu_w = 8 * hmax, mcu_h = 8 * vmax;
  const uint32_t mcus_x = (width + mcu_w - 1) / mcu_w, mcus_y = (height + mcu_h - 1) / mcu_h;
  if (comps.size() == 1) {  // a single component is never interleaved: its MCU is one block
    comps[0].h = comps[0].v = 1;
    hmax = vmax = 1;
  }
  const uint32_t g_mcus_x = comps.size() == 1 ? (width + 7) / 8 : mcus_x;
  const uint32_t g_mcus_y = comps.size() == 1 ? (height + 7) / 8 : mcus_y;
  for (Component& c : comps) {
    c.blocks_w = g_mcus_x * c.h;
    c.blocks_h = g_mcus_y * c.v;
    c.plane.assign((size_t)c.blocks_w * 8 * c.blocks_h * 8, 0);
  }
  BitReader br{data + pos, data + size};
  uint32_t until_restart = restart_interval;
  int expected_rst = 0;
  for (uint32_t my = 0; my < g_mcus_y; ++my)
    for (uint32_t mx = 0; mx < g_mcus_x; ++mx) {
      if (restart_interval && until_restart == 0) {
        // byte-align, expect RSTn
        br.reset();
        const uint8_t* p = br.p;
        while (p + 1 < br.end && !(p[0] == 0xFF && p[1] >= 0xD0 && p[1] <= 0xD7)) ++p;
        if (p + 1 >= br.end || p[1] != 0xD0 + expected_rst) return fail("missing restart marker");
        br.p = p + 2;
        expected_rst = (expected_rst + 1) & 7;
        for (Component& c : comps) c.dc_pred = 0;
        until_restart = restart_interval;
      }
      for (Component& c : comps)
        for (uint32_t by = 0; by < c.v; ++by)
          for (uint32_t bx = 0; bx < c.h; ++bx) {
            int32_t coef[64] = {0};
            bool ok = true;
            int t = decode_symbol(br, dc[c.td], ok);
            if (!ok || t > 11) return fail("corrupt entropy-coded data");
            c.dc_pred += extend(br.get(t), t);
            coef[0] = c.dc_pred * quant[c.tq][0];
            for (int k = 1; k < 64;) {
              int rs = decode_symbol(br, ac[c.ta], ok);
              if (!ok) return fail("corrupt entropy-coded data");
              int r = rs >> 4, s = rs & 15;
              if (s == 0) {
                if (r != 15) break;  // end of block
                k += 16;
                continue;
              }
              k += r;
              if (k > 63) return fail("corrupt entropy-coded data");
              coef[kZigZag[k]] = extend(br.get(s), s) * quant[c.tq][kZigZag[k]];
              ++k;
            }
            size_t stride = (size_t)c.blocks_w * 8;
            size_t x0 = ((size_t)mx * c.h + bx) * 8, y0 = ((size_t)my * c.v + by) * 8;
            idct_block(coef, c.plane.data() + y0 * stride + x0, stride);
          }
      if (restart_interval) --until_restart;
    }

  rgb.resize((size_t)width * height * 3);
  std::vector<std::vector<uint8_t>> full;
  for (const Component& c : comps) full.push_back(full_resolution(c, hmax, vmax, width, height));
  for (uint32_t y = 0; y < height; ++y)
    for (uint32_t x = 0; x < width; ++x) {
      int v[3] = {0, 128, 128};
      for (size_t c = 0; c < comps.size(); ++c) v[c] = full[c][(size_t)y * width + x];
      uint8_t* out = rgb.data() + ((size_t)y * width + x) * 3;
      if (comps.size() == 1) {
        out[0] = out[1] = out[2] = (uint8_t)v[0];
      } else if (adobe_rgb) {
        out[0] = (uint8_t)v[0];
        out[1] = (uint8_t)v[1];
        out[2] = (uint8_t)v[2];
      } else {  // JFIF YCbCr -> RGB
        double Y = v[0], cb = v[1] - 128.0, cr = v[2] - 128.0;
        long r = std::lround(Y + 1.402 * cr);
        long g = std::lround(Y - 0.344136286 * cb - 0.714136286 * cr);
        long b = std::lround(Y + 1.772 * cb);
        out[0] = (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r);
        out[1] = (uint8_t)(g < 0 ? 0 : g > 255 ? 255 : g);
        out[2] = (uint8_t)(b < 0 ? 0 : b > 255 ? 255 : b);
      }
    }
  return true;
}

}  // namespace rt
