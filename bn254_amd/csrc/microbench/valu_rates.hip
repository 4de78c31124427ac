// VALU issue-rate microbenchmark for gfx950 (MI355X).
//
// Measures, per SIMD and at 1/2/4/8 waves per SIMD, the sustained issue cost (cycles per
// wave64 instruction) of the integer / fp64 instructions a 254-bit Montgomery multiply can
// be built from.  The result calibrates the VALU roofline that bench.py reports
// (peak MAC32/s = SIMDs * 64 lanes * clock / cycles(v_mad_u64_u32)).
//
// Build: hipcc -O3 --offload-arch=gfx950 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

constexpr int ITERS = 4096;   // loop trips
constexpr int UNROLL = 16;    // independent chains per trip

enum Op { MAD_U64_U32 = 0, MUL_LO_U32, MUL_HI_U32, MAD_U32_U24, FMA_F64, ADD_CO_PAIR, ADD_U32, MAD_U64_DEP, NOPS };
static const char* op_names[NOPS] = {"v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u32_u24",
                                     "v_fma_f64", "v_add_co+v_addc_co", "v_add_u32", "v_mad_u64_u32(dep chain)"};

template <int OP>
__global__ void __launch_bounds__(64) rate_kernel(uint32_t* out, uint32_t seed, unsigned long long* cycles) {
  uint32_t a = seed + threadIdx.x * 2654435761u, b = seed ^ (threadIdx.x * 40503u + 977u);
  uint64_t acc[UNROLL];
  double facc[UNROLL];
#pragma unroll
  for (int j = 0; j < UNROLL; ++j) { acc[j] = a + j; facc[j] = (double)(a + j); }
  double fa = (double)a * 1e-9, fb = (double)b * 1e-9;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int j = 0; j < UNROLL; ++j) {
      if (OP == MAD_U64_U32) {
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b) : "vcc");
      } else if (OP == MUL_LO_U32) {
        uint32_t lo = (uint32_t)acc[j];
        asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(lo) : "v"(b));
        acc[j] = lo;
      } else if (OP == MUL_HI_U32) {
        uint32_t lo = (uint32_t)acc[j];
        asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(lo) : "v"(b));
        acc[j] = lo;
      } else if (OP == MAD_U32_U24) {
        uint32_t lo = (uint32_t)acc[j];
        asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(lo) : "v"(a), "v"(b));
        acc[j] = lo;
      } else if (OP == FMA_F64) {
        asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(facc[j]) : "v"(fa), "v"(fb));
      } else if (OP == ADD_CO_PAIR) {
        uint32_t lo = (uint32_t)acc[j], hi = (uint32_t)(acc[j] >> 32);
        asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc"
                     : "+v"(lo), "+v"(hi) : "v"(a), "v"(b) : "vcc");
        acc[j] = ((uint64_t)hi << 32) | lo;
      } else if (OP == ADD_U32) {
        uint32_t lo = (uint32_t)acc[j];
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(lo) : "v"(a));
        acc[j] = lo;
      } else if (OP == MAD_U64_DEP) {
        // one single dependent chain: every mad consumes the previous result
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[0]) : "v"(a), "v"(b) : "vcc");
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  uint64_t s = 0; double fs = 0;
#pragma unroll
  for (int j = 0; j < UNROLL; ++j) { s += acc[j]; fs += facc[j]; }
  out[blockIdx.x * 64 + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32) ^ (uint32_t)fs;
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int OP>
static void run(int waves_per_simd, uint32_t* d_out, unsigned long long* d_cyc, int n_cu) {
  // one 64-thread block = one wave; n_cu*4*w blocks -> w waves per SIMD if the dispatcher spreads evenly
  int blocks = n_cu * 4 * waves_per_simd;
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  rate_kernel<OP><<<blocks, 64>>>(d_out, 12345u, d_cyc);   // warm
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  rate_kernel<OP><<<blocks, 64>>>(d_out, 12345u, d_cyc);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long* h = (unsigned long long*)malloc(sizeof(unsigned long long) * blocks);
  CHECK(hipMemcpy(h, d_cyc, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost));
  double avg = 0; for (int i = 0; i < blocks; ++i) avg += (double)h[i]; avg /= blocks;
  free(h);
  double n_inst = (double)ITERS * UNROLL * (OP == ADD_CO_PAIR ? 2 : 1);
  // s_memtime ticks are 100 MHz "realtime"-like on some parts and shader clocks on others; report both
  double wave_inst_total = n_inst * blocks;
  double inst_per_s = wave_inst_total / (ms * 1e-3);
  double per_simd_cyc_at_2p4 = (n_cu * 4.0) * 2.4e9 / inst_per_s;  // cycles per wave-inst per SIMD assuming 2.4 GHz
  printf("{\"op\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.4f, \"memtime_ticks_per_inst_per_wave\": %.3f, "
         "\"wave_inst_per_s\": %.4e, \"cycles_per_wave_inst_per_simd_at_2.4GHz\": %.3f, \"lane_ops_per_s\": %.4e}\n",
         op_names[OP], waves_per_simd, ms, avg / n_inst, inst_per_s, per_simd_cyc_at_2p4, inst_per_s * 64);
  CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
}

int main() {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  int n_cu = prop.multiProcessorCount;
  printf("{\"device\": \"%s\", \"cus\": %d, \"clock_khz\": %d, \"arch\": \"%s\"}\n", prop.name, n_cu, prop.clockRate,
         prop.gcnArchName);
  uint32_t* d_out; unsigned long long* d_cyc;
  int max_blocks = n_cu * 4 * 8;
  CHECK(hipMalloc(&d_out, sizeof(uint32_t) * 64 * max_blocks));
  CHECK(hipMalloc(&d_cyc, sizeof(unsigned long long) * max_blocks));
  for (int w : {1, 2, 4, 8}) {
    run<MAD_U64_U32>(w, d_out, d_cyc, n_cu);
    run<MUL_LO_U32>(w, d_out, d_cyc, n_cu);
    run<MUL_HI_U32>(w, d_out, d_cyc, n_cu);
    run<MAD_U32_U24>(w, d_out, d_cyc, n_cu);
    run<FMA_F64>(w, d_out, d_cyc, n_cu);
    run<ADD_CO_PAIR>(w, d_out, d_cyc, n_cu);
    run<ADD_U32>(w, d_out, d_cyc, n_cu);
    run<MAD_U64_DEP>(w, d_out, d_cyc, n_cu);
  }
  CHECK(hipFree(d_out)); CHECK(hipFree(d_cyc));
  return 0;
}
