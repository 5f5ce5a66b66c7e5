// Issue cost of single VALU instructions on gfx950: each kernel runs a long run of
// independent instructions of one kind; cycles come from s_memtime.
// build: hipcc -O2 --offload-arch=gfx950 -o build/valu_rates tools/ubench/valu_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x

template <int KIND>
__global__ void rate_kernel(unsigned long long* out, int iters) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
  double d0 = threadIdx.x, d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3;
  unsigned u0 = threadIdx.x, u1 = u0 * 3 + 1, u2 = u0 * 5 + 2, u3 = u0 * 7 + 3;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
  unsigned long long q0 = u0, q1 = u1, q2 = u2, q3 = u3;
  unsigned um = 0xFFFF0000u ^ threadIdx.x;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    if (KIND == 0) {
      REP16(asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
    } else if (KIND == 1) {
      REP16(asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));)
    } else if (KIND == 2) {
      REP16(asm volatile("v_fma_f64 %0, %0, %0, %0\n v_fma_f64 %1, %1, %1, %1\n v_fma_f64 %2, %2, %2, %2\n v_fma_f64 %3, %3, %3, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));)
    } else if (KIND == 3) {
      REP16(asm volatile("v_mul_lo_u32 %0, %0, %0\n v_mul_lo_u32 %1, %1, %1\n v_mul_lo_u32 %2, %2, %2\n v_mul_lo_u32 %3, %3, %3" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));)
    } else if (KIND == 4) {
      REP16(asm volatile("v_mul_hi_u32 %0, %0, %0\n v_mul_hi_u32 %1, %1, %1\n v_mul_hi_u32 %2, %2, %2\n v_mul_hi_u32 %3, %3, %3" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));)
    } else if (KIND == 5) {
      REP16(asm volatile("v_min_f32 %0, %0, %1\n v_max_f32 %1, %1, %2\n v_min_f32 %2, %2, %3\n v_max_f32 %3, %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
    } else if (KIND == 6) {
      REP16(asm volatile("v_mul_f64 %0, %0, %0\n v_add_f64 %1, %1, %1\n v_mul_f64 %2, %2, %2\n v_add_f64 %3, %3, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));)
    } else if (KIND == 7) {
      REP16(asm volatile("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));)
    } else if (KIND == 8) {
      REP16(asm volatile("v_xor_b32 %0, %0, %1\n v_xor_b32 %1, %1, %2\n v_xor_b32 %2, %2, %3\n v_xor_b32 %3, %3, %0" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));)
    } else if (KIND == 9) {  // one dependent chain of f64 FMAs
      REP16(asm volatile("v_fma_f64 %0, %0, %0, %0\n v_fma_f64 %0, %0, %0, %0\n v_fma_f64 %0, %0, %0, %0\n v_fma_f64 %0, %0, %0, %0" : "+v"(d0));)
    } else if (KIND == 10) {  // one dependent chain of f32 FMAs
      REP16(asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0" : "+v"(a0));)
    } else if (KIND == 11) {
      REP16(asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : : "vcc");)
    } else if (KIND == 12) {
      REP16(asm volatile("v_rsq_f64 %0, %0\n v_rsq_f64 %1, %1\n v_rsq_f64 %2, %2\n v_rsq_f64 %3, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));)
    } else if (KIND == 14) {  // 32x32 -> 64-bit product (+ 64-bit addend) in one instruction
      REP16(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %5, %6, %1\n v_mad_u64_u32 %2, vcc, %6, %7, %2\n v_mad_u64_u32 %3, vcc, %7, %4, %3" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(u0), "v"(u1), "v"(u2), "v"(u3) : "vcc");)
    } else if (KIND == 15) {
      REP16(asm volatile("v_min3_f32 %0, %0, %1, %2\n v_max3_f32 %1, %1, %2, %3\n v_min3_f32 %2, %2, %3, %0\n v_max3_f32 %3, %3, %0, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
    } else if (KIND == 16) {
      REP16(asm volatile("v_sqrt_f64 %0, %0\n v_sqrt_f64 %1, %1\n v_sqrt_f64 %2, %2\n v_sqrt_f64 %3, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));)
    } else if (KIND == 17) {
      REP16(asm volatile("v_lshl_add_u32 %0, %0, 1, %1\n v_lshl_or_b32 %1, %1, 2, %2\n v_add3_u32 %2, %2, %3, %0\n v_bfe_u32 %3, %3, 1, 9" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));)
    } else if (KIND == 18) {  // select by an SGPR pair that is not VCC
      REP16(asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]\n v_cndmask_b32_e64 %1, %1, %2, s[10:11]\n v_cndmask_b32_e64 %2, %2, %3, s[10:11]\n v_cndmask_b32_e64 %3, %3, %0, s[10:11]" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : : "s10", "s11");)
    } else if (KIND == 19) {  // select by a mask held in a vector register
      REP16(asm volatile("v_bfi_b32 %0, %4, %0, %1\n v_bfi_b32 %1, %4, %1, %2\n v_bfi_b32 %2, %4, %2, %3\n v_bfi_b32 %3, %4, %3, %0" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(um));)
    } else if (KIND == 20) {  // compare into VCC, then select by it
      REP16(asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_lt_f32 vcc, %1, %0\n v_cndmask_b32 %3, %3, %2, vcc" : "+v"(a0), "+v"(a1), "+v"(u2), "+v"(u3) : : "vcc");)
    } else if (KIND == 13) {
      REP16(asm volatile("v_cvt_f64_u32 %0, %4\n v_cvt_f64_u32 %1, %5\n v_cvt_f64_u32 %2, %6\n v_cvt_f64_u32 %3, %7" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(u0), "v"(u1), "v"(u2), "v"(u3));)
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float sink = a0 + a1 + a2 + a3 + p0.x + p1.y + p2.x + p3.y + (float)(d0 + d1 + d2 + d3) + (float)(u0 ^ u1 ^ u2 ^ u3) + (float)(q0 ^ q1 ^ q2 ^ q3);
  if (sink == 12345.678f) out[1] = 1;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

template <int KIND>
void run(const char* name, int waves_per_simd) {
  unsigned long long* d;
  hipMalloc(&d, 16);
  const int iters = 2000;
  // one workgroup per CU would need occupancy control; launch 256 * k workgroups of 256 threads
  // (4 waves -> one per SIMD), k workgroups resident per CU give k waves per SIMD
  int blocks = 256 * waves_per_simd;
  rate_kernel<KIND><<<blocks, 256>>>(d, 10);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0);
  rate_kernel<KIND><<<blocks, 256>>>(d, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[2];
  hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
  double n_inst = (double)iters * 64.0;  // per wave
  printf("%-14s waves/SIMD %d: %.2f counter ticks per instr per wave, %.3f ms; SIMD rate %.2f instr/ns/SIMD\n",
         name, waves_per_simd, (double)h[0] / n_inst, ms,
         n_inst * waves_per_simd / (ms * 1e6));
  hipFree(d);
}

int main() {
  for (int w : {1, 2, 4}) {
    run<0>("v_fma_f32", w);
    run<1>("v_pk_fma_f32", w);
    run<2>("v_fma_f64", w);
    run<6>("v_mul/add_f64", w);
    run<3>("v_mul_lo_u32", w);
    run<4>("v_mul_hi_u32", w);
    run<5>("v_min/max_f32", w);
    run<8>("v_xor_b32", w);
    run<11>("v_cndmask", w);
    run<7>("v_rcp_f64", w);
    run<12>("v_rsq_f64", w);
    run<13>("v_cvt_f64_u32", w);
    run<14>("v_mad_u64_u32", w);
    run<15>("v_min3/max3_f32", w);
    run<16>("v_sqrt_f64", w);
    run<17>("3-op int ALU", w);
    run<18>("v_cndmask sgpr", w);
    run<19>("v_bfi_b32", w);
    run<20>("v_cmp+v_cndmask", w);
    run<9>("dep v_fma_f64", w);
    run<10>("dep v_fma_f32", w);
  }
  return 0;
}
