// Ceiling experiment: chains of the production Fq product (bn254_field.h) with all operands in
// registers, at 1/2/4 waves per SIMD.   hipcc -O3 --offload-arch=gfx950 -I.. fp_mul_chain.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "../bn254_field.h"
using namespace bn254;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

template <int CHAINS>
__global__ void __launch_bounds__(64) k_chain(const int32_t* in, int32_t* out, int n, int iters) {
  int i = blockIdx.x * 64 + threadIdx.x;
  Fp x[CHAINS], y;
  for (int k = 0; k < BN_LIMBS; ++k) { y.v[k] = in[k * n + i] & BN_MASK; }
  for (int c = 0; c < CHAINS; ++c) for (int k = 0; k < BN_LIMBS; ++k) x[c].v[k] = (in[(BN_LIMBS + k) * n + i] + c) & BN_MASK;
  for (int t = 0; t < iters; ++t) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) x[c] = fp_mul(x[c], y);
  }
  Fp s = x[0];
  for (int c = 1; c < CHAINS; ++c) s = fp_add(s, x[c]);
  for (int k = 0; k < BN_LIMBS; ++k) out[k * n + i] = s.v[k];
}
// same work through the Fq2 layer (Karatsuba + lazy adds + norms), still all in registers
__global__ void __launch_bounds__(64) k_chain_fp2(const int32_t* in, int32_t* out, int n, int iters) {
  int i = blockIdx.x * 64 + threadIdx.x;
  Fp2 x, y;
  for (int k = 0; k < BN_LIMBS; ++k) { y.c0.v[k] = in[k * n + i] & BN_MASK; y.c1.v[k] = (in[k * n + i] >> 3) & BN_MASK; x.c0.v[k] = in[(BN_LIMBS + k) * n + i] & BN_MASK; x.c1.v[k] = (in[(BN_LIMBS + k) * n + i] >> 2) & BN_MASK; }
  for (int t = 0; t < iters; ++t) x = fp2_norm(fp2_mul(x, y));
  for (int k = 0; k < BN_LIMBS; ++k) out[k * n + i] = x.c0.v[k] ^ x.c1.v[k];
}
int main() {
  const int iters = 4000;
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  int simds = prop.multiProcessorCount * 4;
  for (int w : {1, 2, 4, 8}) {
    int n = simds * 64 * w;
    int32_t *d_in, *d_out;
    CHECK(hipMalloc(&d_in, sizeof(int32_t) * 2 * BN_LIMBS * n)); CHECK(hipMalloc(&d_out, sizeof(int32_t) * BN_LIMBS * n));
    CHECK(hipMemset(d_in, 0x5a, sizeof(int32_t) * 2 * BN_LIMBS * n));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto timeit = [&](auto launch, const char* name, double muls_per_lane) {
      launch(); CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(e0)); launch(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      double total = muls_per_lane * n;
      printf("{\"kernel\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.3f, \"fq_products_per_s\": %.4e, \"ns_per_product_per_wave\": %.2f, \"TMAC32_per_s_algorithmic\": %.2f}\n",
             name, w, ms, total / (ms * 1e-3), ms * 1e6 / muls_per_lane, total * 136 / (ms * 1e-3) / 1e12);
    };
    timeit([&] { k_chain<1><<<n / 64, 64>>>(d_in, d_out, n, iters); }, "fp_mul chain x1", iters);
    timeit([&] { k_chain<2><<<n / 64, 64>>>(d_in, d_out, n, iters); }, "fp_mul chain x2 (independent)", 2.0 * iters);
    timeit([&] { k_chain_fp2<<<n / 64, 64>>>(d_in, d_out, n, iters); }, "fp2_mul+norm chain", 3.0 * iters);
    CHECK(hipFree(d_in)); CHECK(hipFree(d_out));
  }
  return 0;
}
