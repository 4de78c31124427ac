// Does a lone wave issue INDEPENDENT v_mad_u64_u32 faster than a DEPENDENT chain?  And does the carry-out
// destination (vcc vs distinct SGPR pairs) matter?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
constexpr int ITERS = 1 << 17;
template <int P>
__global__ void __launch_bounds__(64) kern(uint32_t* out, uint32_t seed) {
  uint32_t a = seed + threadIdx.x * 2654435761u, b = seed ^ (threadIdx.x * 40503u + 977u);
  uint64_t acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = a + j;
  for (int i = 0; i < ITERS; ++i) {
    if (P == 0) {   // dependent chain, carry-out to vcc
#pragma unroll
      for (int j = 0; j < 8; ++j) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[0]) : "v"(a), "v"(b) : "vcc");
    } else if (P == 1) {   // 8 independent accumulators, carry-out to vcc
#pragma unroll
      for (int j = 0; j < 8; ++j) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b) : "vcc");
    } else if (P == 2) {   // 8 independent accumulators, distinct SGPR carry-outs
      asm volatile("v_mad_u64_u32 %0, s[20:21], %8, %9, %0\n v_mad_u64_u32 %1, s[22:23], %8, %9, %1\n v_mad_u64_u32 %2, s[24:25], %8, %9, %2\n"
                   "v_mad_u64_u32 %3, s[26:27], %8, %9, %3\n v_mad_u64_u32 %4, s[28:29], %8, %9, %4\n v_mad_u64_u32 %5, s[30:31], %8, %9, %5\n"
                   "v_mad_u64_u32 %6, s[32:33], %8, %9, %6\n v_mad_u64_u32 %7, s[34:35], %8, %9, %7"
                   : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]) : "v"(a), "v"(b)
                   : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s32", "s33", "s34", "s35");
    } else if (P == 6) {   // ONE dependent chain, rotating distinct SGPR carry-outs: pure accumulator latency
      asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0\n v_mad_u64_u32 %0, s[22:23], %1, %2, %0\n v_mad_u64_u32 %0, s[24:25], %1, %2, %0\n"
                   "v_mad_u64_u32 %0, s[26:27], %1, %2, %0\n v_mad_u64_u32 %0, s[28:29], %1, %2, %0\n v_mad_u64_u32 %0, s[30:31], %1, %2, %0\n"
                   "v_mad_u64_u32 %0, s[32:33], %1, %2, %0\n v_mad_u64_u32 %0, s[34:35], %1, %2, %0"
                   : "+v"(acc[0]) : "v"(a), "v"(b)
                   : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s32", "s33", "s34", "s35");
    } else if (P == 7) {   // TWO interleaved dependent chains, rotating distinct SGPR carry-outs
      asm volatile("v_mad_u64_u32 %0, s[20:21], %2, %3, %0\n v_mad_u64_u32 %1, s[22:23], %2, %3, %1\n v_mad_u64_u32 %0, s[24:25], %2, %3, %0\n"
                   "v_mad_u64_u32 %1, s[26:27], %2, %3, %1\n v_mad_u64_u32 %0, s[28:29], %2, %3, %0\n v_mad_u64_u32 %1, s[30:31], %2, %3, %1\n"
                   "v_mad_u64_u32 %0, s[32:33], %2, %3, %0\n v_mad_u64_u32 %1, s[34:35], %2, %3, %1"
                   : "+v"(acc[0]), "+v"(acc[1]) : "v"(a), "v"(b)
                   : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s32", "s33", "s34", "s35");
    } else if (P == 8) {   // ONE dependent chain alternating between only TWO carry-out pairs
      asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0\n v_mad_u64_u32 %0, s[22:23], %1, %2, %0\n v_mad_u64_u32 %0, s[20:21], %1, %2, %0\n"
                   "v_mad_u64_u32 %0, s[22:23], %1, %2, %0\n v_mad_u64_u32 %0, s[20:21], %1, %2, %0\n v_mad_u64_u32 %0, s[22:23], %1, %2, %0\n"
                   "v_mad_u64_u32 %0, s[20:21], %1, %2, %0\n v_mad_u64_u32 %0, s[22:23], %1, %2, %0"
                   : "+v"(acc[0]) : "v"(a), "v"(b) : "s20", "s21", "s22", "s23");
    } else if (P == 9) {   // 8 independent accumulators alternating between only TWO carry-out pairs
      asm volatile("v_mad_u64_u32 %0, s[20:21], %8, %9, %0\n v_mad_u64_u32 %1, s[22:23], %8, %9, %1\n v_mad_u64_u32 %2, s[20:21], %8, %9, %2\n"
                   "v_mad_u64_u32 %3, s[22:23], %8, %9, %3\n v_mad_u64_u32 %4, s[20:21], %8, %9, %4\n v_mad_u64_u32 %5, s[22:23], %8, %9, %5\n"
                   "v_mad_u64_u32 %6, s[20:21], %8, %9, %6\n v_mad_u64_u32 %7, s[22:23], %8, %9, %7"
                   : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]) : "v"(a), "v"(b)
                   : "s20", "s21", "s22", "s23");
    } else if (P == 10) {   // one chain, FOUR carry-out pairs
      asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0\n v_mad_u64_u32 %0, s[22:23], %1, %2, %0\n v_mad_u64_u32 %0, s[24:25], %1, %2, %0\n"
                   "v_mad_u64_u32 %0, s[26:27], %1, %2, %0\n v_mad_u64_u32 %0, s[20:21], %1, %2, %0\n v_mad_u64_u32 %0, s[22:23], %1, %2, %0\n"
                   "v_mad_u64_u32 %0, s[24:25], %1, %2, %0\n v_mad_u64_u32 %0, s[26:27], %1, %2, %0"
                   : "+v"(acc[0]) : "v"(a), "v"(b) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
    } else if (P == 3) {   // 2 interleaved dependent chains
#pragma unroll
      for (int j = 0; j < 8; ++j) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[j & 1]) : "v"(a), "v"(b) : "vcc");
    } else if (P == 4) {   // dependent chain of plain C (compiler-chosen encoding)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[0] += (uint64_t)a * (b + j);
    } else if (P == 5) {   // independent plain C
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += (uint64_t)a * (b + j);
    }
    asm volatile("" : "+v"(a));
  }
  uint64_t s = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) s += acc[j];
  out[blockIdx.x * 64 + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32);
}
template <int P> static void run(const char* name, int w, uint32_t* d_out, int n_cu) {
  int blocks = n_cu * 4 * w;
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  kern<P><<<blocks, 64>>>(d_out, 12345u); CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  for (int r = 0; r < 4; ++r) kern<P><<<blocks, 64>>>(d_out, 12345u);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 4;
  printf("{\"pattern\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.3f, \"ns_per_mad_per_wave\": %.3f, \"ns_per_mad_per_simd\": %.3f}\n", name, w, ms,
         ms * 1e6 / (ITERS * 8.0), ms * 1e6 / (ITERS * 8.0) / w);
}
int main() {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  int n_cu = prop.multiProcessorCount;
  uint32_t* d_out; CHECK(hipMalloc(&d_out, 4 * 64 * n_cu * 4 * 8));
  for (int w : {1, 2}) {
    run<0>("dependent chain (vcc)", w, d_out, n_cu); run<1>("8 independent (vcc)", w, d_out, n_cu); run<2>("8 independent (distinct sgpr carry)", w, d_out, n_cu);
    run<3>("2 interleaved chains (vcc)", w, d_out, n_cu); run<4>("dependent chain, compiler", w, d_out, n_cu); run<5>("8 independent, compiler", w, d_out, n_cu);
    run<6>("1 chain, 8 rotating sgpr carries", w, d_out, n_cu); run<7>("2 chains, 8 rotating sgpr carries", w, d_out, n_cu);
    run<8>("1 chain, 2 alternating sgpr carries", w, d_out, n_cu); run<9>("8 independent, 2 alternating sgpr carries", w, d_out, n_cu);
    run<10>("1 chain, 4 rotating sgpr carries", w, d_out, n_cu);
  }
  return 0;
}
