// Single-wave issue behaviour on gfx950: does interleaving v_mad_*64 with independent full-rate
// VALU ops shorten the stream?  (complements valu_rates.hip)   hipcc -O3 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
constexpr int ITERS = 2048;
enum Pat { MAD = 0, MADI, ADD, MAD_ADD, MAD_ADD2, MAD_ADD4, ALIGNBIT, LSHLADD64, ASHR64, MOV, MAD_SGPR, NP };
static const char* names[NP] = {"mad_u64_u32 x16", "mad_i64_i32 x16", "add_u32 x16", "(mad,add) x16", "(mad,add,add) x16", "(mad,4 add) x16",
                                "alignbit x16", "lshl_add_u64 x16", "ashrrev_i64 x16", "v_mov x16", "mad_u64_u32 sgpr-operand x16"};
template <int P>
__global__ void __launch_bounds__(64) kern(uint32_t* out, uint32_t seed, unsigned long long* cyc) {
  uint32_t a = seed + threadIdx.x * 2654435761u, b = seed ^ (threadIdx.x * 40503u + 977u);
  uint64_t acc[16]; uint32_t x[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) { acc[j] = a + j; x[j] = b + j; }
  uint32_t sconst = __builtin_amdgcn_readfirstlane(seed | 1);
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (P == MAD || P == MAD_ADD || P == MAD_ADD2 || P == MAD_ADD4) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b) : "vcc");
      if (P == MADI) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b) : "vcc");
      if (P == MAD_SGPR) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "s"(sconst) : "vcc");
      if (P == ADD || P == MAD_ADD || P == MAD_ADD2 || P == MAD_ADD4) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[j]) : "v"(a));
      if (P == MAD_ADD2 || P == MAD_ADD4) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[(j + 5) & 15]) : "v"(b));
      if (P == MAD_ADD4) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[(j + 9) & 15]) : "v"(b)); asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[(j + 3) & 15]) : "v"(a)); }
      if (P == ALIGNBIT) asm volatile("v_alignbit_b32 %0, %0, %1, 27" : "+v"(x[j]) : "v"(a));
      if (P == LSHLADD64) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[j]) : "v"(acc[(j + 1) & 15]));
      if (P == ASHR64) asm volatile("v_ashrrev_i64 %0, 27, %0" : "+v"(acc[j]));
      if (P == MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(x[j]) : "v"(x[(j + 1) & 15]));
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  uint64_t s = 0;
#pragma unroll
  for (int j = 0; j < 16; ++j) s += acc[j] + x[j];
  out[blockIdx.x * 64 + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32);
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int P> static void run(int w, uint32_t* d_out, unsigned long long* d_cyc, int n_cu) {
  int blocks = n_cu * 4 * w;
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  kern<P><<<blocks, 64>>>(d_out, 12345u, d_cyc); CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  for (int r = 0; r < 4; ++r) kern<P><<<blocks, 64>>>(d_out, 12345u, d_cyc);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 4;
  unsigned long long* h = (unsigned long long*)malloc(8 * blocks);
  CHECK(hipMemcpy(h, d_cyc, 8 * blocks, hipMemcpyDeviceToHost));
  double avg = 0; for (int i = 0; i < blocks; ++i) avg += h[i]; avg /= blocks; free(h);
  int per_group = (P == MAD_ADD) ? 2 : (P == MAD_ADD2) ? 3 : (P == MAD_ADD4) ? 5 : 1;
  printf("{\"pattern\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.4f, \"memtime_ticks_per_group\": %.2f, \"instr_per_group\": %d, \"wall_ns_per_group_per_wave\": %.3f}\n",
         names[P], w, ms, avg / (ITERS * 16.0), per_group, ms * 1e6 / (ITERS * 16.0));
}
int main() {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  int n_cu = prop.multiProcessorCount;
  uint32_t* d_out; unsigned long long* d_cyc;
  CHECK(hipMalloc(&d_out, 4 * 64 * n_cu * 4 * 8)); CHECK(hipMalloc(&d_cyc, 8 * n_cu * 4 * 8));
  for (int w : {1, 2, 4}) {
    run<MAD>(w, d_out, d_cyc, n_cu); run<MADI>(w, d_out, d_cyc, n_cu); run<MAD_SGPR>(w, d_out, d_cyc, n_cu); run<ADD>(w, d_out, d_cyc, n_cu);
    run<MAD_ADD>(w, d_out, d_cyc, n_cu); run<MAD_ADD2>(w, d_out, d_cyc, n_cu); run<MAD_ADD4>(w, d_out, d_cyc, n_cu);
    run<ALIGNBIT>(w, d_out, d_cyc, n_cu); run<LSHLADD64>(w, d_out, d_cyc, n_cu); run<ASHR64>(w, d_out, d_cyc, n_cu); run<MOV>(w, d_out, d_cyc, n_cu);
  }
  return 0;
}
