// Would the matrix pipe pay for the constant-operand half of a Montgomery reduction?  (round-1 review, item 9)
//
// The m*q half of a reduction (81 of the 243 multiply-adds of a dual product with 9 x 29-bit limbs) is a contraction
// with the constant q, and the matrix pipe is idle.  A full-width m*q on v_mfma_i32_32x32x16_i8 needs, per wave of 64
// reductions: all of m first (a 45-multiply-add triangular product T_lo * (-q^-1) on VALU, because the digit-serial m_k
// depends on the running column), m as 33 balanced byte digits in the B-operand layout (bit-field extracts +
// v_permlane32_swap: an item's K-slices live in lanes L and L+32), 8 MFMAs, and the 35 high output columns folded back
// into nine 29-bit limbs (shift-adds + carries).  This microbenchmark times the two instruction MIXES on the device —
// not a functional reduction: the arithmetic of variant 1 is meaningless, its instruction stream is what the real thing
// would issue — with two waves per SIMD like the pair kernels:
//   variant 0: 243 v_mad_i64_i32                                   (the product as shipped; its 103 other instructions
//                                                                    are common to both variants and left out)
//   variant 1: 207 v_mad_i64_i32 + 8 v_mfma_i32_32x32x16_i8 + 8 v_permlane32_swap + 40 v_bfe_i32 (byte digits)
//              + 40 v_lshl_add_u32 (column folding) + 27 carry instructions (v_add / v_bfe / v_ashr)
// Build: hipcc -O3 --offload-arch=gfx950 mfma_reduce_cost.hip -o mfma_reduce_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

typedef int v16i __attribute__((ext_vector_type(16)));
constexpr int ITERS = 2000;

template <int VARIANT>
__global__ void __launch_bounds__(256) mix_kernel(uint32_t* out, uint32_t seed) {
  int32_t a[9], b[9];
  int64_t acc[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) { a[i] = (int32_t)(seed * (i + 3) + threadIdx.x * 2654435761u); b[i] = (int32_t)(seed ^ (threadIdx.x * 40503u + i)); acc[i] = i; }
  v16i c0 = {0}, c1 = {0};
  long ma = (long)seed * 0x9E3779B97F4A7C15l + threadIdx.x, mb = (long)seed + threadIdx.x * 7;
  uint32_t x = seed + threadIdx.x, y = seed * 3 + threadIdx.x;
  for (int it = 0; it < ITERS; ++it) {
    constexpr int NMAD = VARIANT == 0 ? 243 : 207;
#pragma unroll
    for (int k = 0; k < NMAD; ++k) {
      asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc[k % 9]) : "v"(a[k % 9]), "v"(b[(k * 5) % 9]) : "vcc");
    }
    if (VARIANT == 1) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {      // 8 MFMAs in two accumulator sets, spaced like the real use (one K-block pair per output block)
        c0 = __builtin_amdgcn_mfma_i32_32x32x16_i8(ma, mb, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_i32_32x32x16_i8(mb, ma, c1, 0, 0, 0);
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));
#pragma unroll
      for (int k = 0; k < 40; ++k) asm volatile("v_bfe_i32 %0, %1, %2, 8" : "=v"(x) : "v"(y), "n"(3));          // byte digits of m
#pragma unroll
      for (int k = 0; k < 40; ++k) asm volatile("v_lshl_add_u32 %0, %1, 8, %0" : "+v"(y) : "v"(x));            // fold columns into limbs
#pragma unroll
      for (int k = 0; k < 9; ++k) asm volatile("v_add_u32 %0, %0, %1\n\tv_bfe_i32 %1, %0, 0, 29\n\tv_ashrrev_i32 %0, 29, %0" : "+v"(x), "+v"(y));
      ma += c0[0]; mb += c1[5];
    }
  }
  uint64_t s = 0;
#pragma unroll
  for (int i = 0; i < 9; ++i) s += (uint64_t)acc[i];
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32) ^ x ^ y ^ (uint32_t)c0[3] ^ (uint32_t)c1[7];
}

template <int VARIANT>
static double run(int n_cu, uint32_t* d_out) {
  const int blocks = n_cu * 2;           // 256-thread workgroups: two per CU = two waves per SIMD
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  mix_kernel<VARIANT><<<blocks, 256>>>(d_out, 12345u);
  CHECK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipEventRecord(e0));
    mix_kernel<VARIANT><<<blocks, 256>>>(d_out, 12345u);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  return best * 1e6 / ITERS;             // ns per iteration (per wave pair on a SIMD)
}

int main() {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  uint32_t* d_out; CHECK(hipMalloc(&d_out, sizeof(uint32_t) * 256 * prop.multiProcessorCount * 2));
  double t0 = run<0>(prop.multiProcessorCount, d_out), t1 = run<1>(prop.multiProcessorCount, d_out);
  printf("{\"arch\": \"%s\", \"waves_per_simd\": 2, \"ns_per_product_mix\": {\"valu_only_243_mads\": %.1f, "
         "\"mfma_variant_207_mads_8_mfma_conversions\": %.1f}, \"mfma_variant_over_valu\": %.3f}\n",
         prop.gcnArchName, t0, t1, t1 / t0);
  CHECK(hipFree(d_out));
  return 0;
}
