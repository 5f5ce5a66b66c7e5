// Philox4x32-10 counter-based generator (Salmon et al., SC'11) and the mappings
// from its 128-bit blocks to the uniform variates the render path draws.
//
// The reference draws from rand 0.8.5's thread_rng() (unseeded ChaCha12); only
// the distributions are reproducible, so this build replaces the *sequence* by
// a counter addressed as (pixel, sample, bounce, purpose|seq) under a 64-bit
// key.  A draw is then independent of traversal order and of how samples are
// sharded over devices.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define RT_HD __host__ __device__ __forceinline__
#else
#define RT_HD inline
#endif

namespace rt {

struct u32x4 {
  uint32_t x, y, z, w;
};

RT_HD uint32_t mulhi32(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umulhi(a, b);
#else
  return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32);
#endif
}

RT_HD u32x4 philox4x32_10(u32x4 c, uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
  const uint32_t W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#if defined(__HIP_DEVICE_COMPILE__)
  // The key is the same for every draw of a launch, so the compiler would keep all twenty round
  // keys in scalar registers for the whole kernel (and spill other scalars to make room for them).
  // Hidden behind this barrier they are formed again at every block, by scalar additions that cost
  // the vector pipeline nothing.
  asm volatile("" : "+s"(k0), "+s"(k1));
#endif
#if defined(__HIPCC__)
#pragma unroll
#endif
#ifndef RT_PHILOX_ROUNDS
#define RT_PHILOX_ROUNDS 10
#endif
  for (int r = 0; r < RT_PHILOX_ROUNDS; ++r) {
    // one 32x32->64 product each (v_mad_u64_u32 issues faster than v_mul_hi_u32 + v_mul_lo_u32)
    const uint64_t p0 = (uint64_t)M0 * (uint64_t)c.x, p1 = (uint64_t)M1 * (uint64_t)c.z;
    uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    u32x4 n;
    n.x = hi1 ^ c.y ^ k0;
    n.y = lo1;
    n.z = hi0 ^ c.w ^ k1;
    n.w = lo0;
    c = n;
    k0 += W0;
    k1 += W1;
  }
  return c;
}

// Purposes (high byte of counter word 3).
enum : uint32_t {
  DRAW_JITTER = 0,   // (jx, jy) of src/main.rs:87-90
  DRAW_LENS = 1,     // rand_in_unit_disk, seq = rejection iteration (vec3.rs:49-60)
  DRAW_TIME = 2,     // shutter time (camera.rs:26-27)
  DRAW_SCATTER = 3,  // rand_unit_vector, seq = rejection iteration (vec3.rs:35-47)
  DRAW_FRESNEL = 4,  // dielectric reflect/refract choice (dielectric.rs:27)
  DRAW_MEDIUM = 5    // free-flight distance, seq = object index (constant_medium.rs:54)
};

RT_HD u32x4 draw_block(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t bounce,
                       uint32_t purpose, uint32_t seq) {
  u32x4 c;
  c.x = pixel;
  c.y = sample;
  c.z = bounce;
  c.w = (purpose << 24) | (seq & 0x00FFFFFFu);
  return philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
}

// 53-bit uniform in [0,1): the `Standard` f64 distribution of rand 0.8.5.
RT_HD double u01_53(uint32_t lo, uint32_t hi) {
  uint64_t v = ((uint64_t)hi << 32) | lo;
  return (double)(v >> 11) * (1.0 / 9007199254740992.0);
}

// Three 42-bit uniforms in [-1,1) out of one block: the per-component draw of
// rand_unit_vector / rand_in_unit_disk (Uniform::new_inclusive(-1,1); the
// closed end has measure zero and is always rejected by len^2 < 1).
RT_HD void sym3_42(u32x4 b, double& x, double& y, double& z) {
  uint64_t a = ((uint64_t)b.y << 32) | b.x;
  uint64_t c = ((uint64_t)b.w << 32) | b.z;
  uint64_t ix = a >> 22;
  uint64_t iy = c >> 22;
  uint64_t iz = ((a & 0x1FFFFFull) << 21) | (c & 0x1FFFFFull);
  const double s = 2.0 / 4398046511104.0;  // 2 / 2^42
  x = (double)ix * s - 1.0;
  y = (double)iy * s - 1.0;
  z = (double)iz * s - 1.0;
}

}  // namespace rt
