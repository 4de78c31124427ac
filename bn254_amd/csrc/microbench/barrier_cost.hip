// microbenchmark: cost of a workgroup barrier for a 4-wave workgroup with one wave per SIMD (the wave-role kernels,
// bn254_quad.hip, execute several hundred per verify), with all waves in the SAME code and with every wave in its OWN code
// (roles).  usage: ./barrier_cost  -> ns per iteration
#include <hip/hip_runtime.h>
#include <cstdio>
template <int R> __device__ __noinline__ unsigned block_of_work(unsigned acc) {
  // ~1200 straight-line independent-ish VALU instructions, different constants per R so that the four blocks are distinct code
#pragma unroll
  for (int k = 0; k < 600; ++k) acc = (acc ^ (unsigned)(k * 7919 + R * 104729)) + (acc >> ((k + R) & 15));
  return acc;
}
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) k_barriers(int iters, int mode, unsigned* out) {
  extern __shared__ int lds[];
  unsigned acc = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  for (int i = 0; i < iters; ++i) {
    if (mode == 1) acc = block_of_work<0>(acc);                        // every wave the same code
    else if (mode == 2) {                                              // every wave its own code
      if (w == 0) acc = block_of_work<0>(acc); else if (w == 1) acc = block_of_work<1>(acc);
      else if (w == 2) acc = block_of_work<2>(acc); else acc = block_of_work<3>(acc);
    }
    lds[threadIdx.x] = (int)acc;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    acc += (unsigned)lds[(threadIdx.x + 64) & 255];
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
int main() {
  unsigned* out; hipMalloc(&out, 1024 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[3] = {"barrier only", "same code in the four waves", "own code per wave"};
  for (int wgs : {1, 256}) for (int mode : {0, 1, 2}) {
    const int iters = 5000;
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0); k_barriers<<<wgs, 256, 100 * 1024>>>(iters, mode, out); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("{\"workgroups\": %d, \"between_barriers\": \"%s\", \"ns_per_iteration\": %.1f}\n", wgs, names[mode], best * 1e6 / iters);
  }
  return 0;
}
