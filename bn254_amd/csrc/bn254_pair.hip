// Second device translation unit of libbn254hip.so: the Miller loop and the final exponentiation of a verify
// carried by a LANE PAIR (bn254_fp2_pair.h): lane 2i holds the real parts and lane 2i+1 the imaginary parts of
// every Fq2 value of item i.  Same tower / pairing source as bn254_hip.hip (bn254_field.h, bn254_pairing.h),
// compiled against the pair implementation of the fp2_* interface.  Per lane: half the multiplications, half the
// registers (<= 256, two waves per SIMD), half the LDS (an accumulator slot is 61 words).
// Replaces, for ECDSA::verify (/root/reference/src/ecdsa.rs:49-64), k_miller_verify + k_final_exp.
#include <hip/hip_runtime.h>

#define BN_SPLIT_FP2 1
#define bn254 bn254_pair   // own namespace: the Fq2 / Fq12 types differ from the other translation unit
#include "bn254_pairing.h"

using namespace bn254;

#include "bn254_ws.h"

// 256-thread workgroups: the four waves of a workgroup land on the four SIMDs of a CU, so two workgroups per CU
// give exactly two waves per SIMD.  With one-wave workgroups the dispatcher filled the SIMDs unevenly (1.54 resident
// waves per SIMD on average, rocprofv3 SQ_WAVE_CYCLES) and the pair layout gained nothing.
#ifndef BN_PAIR_WG
#define BN_PAIR_WG 256
#endif
#define KERNEL_PAIR __global__ __launch_bounds__(BN_PAIR_WG) __attribute__((amdgpu_waves_per_eu(2, 2)))

struct Fp12PairSlot { Fp12 v; int32_t pad; };
static_assert(sizeof(Fp12PairSlot) == 61 * 4, "LDS slot must be 61 words (odd stride: conflict-free)");

__device__ __forceinline__ Fp2 ws_load_fp2_own(const Ws& ws, int plane_re, size_t i) {
  Fp2 r;
  r.c[0] = ws_load_fp(ws, plane_re + (int)(threadIdx.x & 1u), i);
  return r;
}
__device__ __forceinline__ void ws_load_f12_own(const Ws& ws, size_t i, Fp12& f) {
  Fp2* c[6] = {&f.c0.c0, &f.c0.c1, &f.c0.c2, &f.c1.c0, &f.c1.c1, &f.c1.c2};
#pragma unroll
  for (int k = 0; k < 6; ++k) *c[k] = ws_load_fp2_own(ws, PL_F0 + 2 * k, i);
}
__device__ __forceinline__ void ws_store_f12_own(const Ws& ws, size_t i, const Fp12& f) {
  const Fp2* c[6] = {&f.c0.c0, &f.c0.c1, &f.c0.c2, &f.c1.c0, &f.c1.c1, &f.c1.c2};
#pragma unroll
  for (int k = 0; k < 6; ++k) ws_store_fp(ws, PL_F0 + 2 * k + (int)(threadIdx.x & 1u), i, c[k]->c[0]);
}

// f = miller(H(m), pk) * miller(sig, -G2); item = lane >> 1.  Both lanes of a pair take every branch together
// (item-level conditions only), so the DPP exchanges always find their partner active.
KERNEL_PAIR void k_miller_verify_pair(size_t n, Ws ws, const uint32_t* map, const uint32_t* count) {
  size_t i = ((size_t)blockIdx.x * BN_PAIR_WG + threadIdx.x) >> 1;
  if (i >= n) return;
  if (map) { if (i >= *count) return; i = map[i]; }
  G1Affine sig, h;
  G2Affine pk;
  ws_load_g1(ws, PL_P1X, BY_P1_INF, i, sig);
  ws_load_g1(ws, PL_P2X, BY_P2_INF, i, h);
  pk.x = ws_load_fp2_own(ws, PL_QX0, i);
  pk.y = ws_load_fp2_own(ws, PL_QY0, i);
  pk.inf = ws_byte(ws, BY_Q_INF, i) != 0;
  __shared__ Fp12PairSlot lds_f[BN_PAIR_WG];
  Fp12& f = lds_f[threadIdx.x].v;
  miller_loop<true, true>(f, h, pk, sig);
  ws_store_f12_own(ws, i, f);
}
KERNEL_PAIR void k_final_exp_pair(size_t n, Ws ws, int use_hash, uint8_t* status_out, const uint32_t* map, const uint32_t* count) {
  size_t i = ((size_t)blockIdx.x * BN_PAIR_WG + threadIdx.x) >> 1;
  if (i >= n) return;
  if (map) { if (i >= *count) return; i = map[i]; }
  Fp12 f;
  ws_load_f12_own(ws, i, f);
  uint8_t st = ws_byte(ws, BY_ST_DECODE, i);
  if (st == ST_OK && use_hash) st = ws_byte(ws, BY_ST_HASH, i);
  __shared__ Fp12PairSlot lds_acc[BN_PAIR_WG];
  final_exponentiation(f, f, lds_acc[threadIdx.x].v);
  const bool one = fp12_is_one(f);   // combined over the pair
  if ((threadIdx.x & 1u) == 0) status_out[i] = st != ST_OK ? st : (one ? (uint8_t)ST_OK : (uint8_t)ST_VERIFICATION_FAILED);
}

int bn254_pair_miller_verify(size_t n, Ws ws, const uint32_t* map, const uint32_t* count, hipStream_t s) {
  k_miller_verify_pair<<<(unsigned)((2 * n + BN_PAIR_WG - 1) / BN_PAIR_WG), BN_PAIR_WG, 0, s>>>(n, ws, map, count);
  HIP_TRY(hipGetLastError());
  return 0;
}
int bn254_pair_final_exp(size_t n, Ws ws, int use_hash, uint8_t* status_out, const uint32_t* map, const uint32_t* count, hipStream_t s) {
  k_final_exp_pair<<<(unsigned)((2 * n + BN_PAIR_WG - 1) / BN_PAIR_WG), BN_PAIR_WG, 0, s>>>(n, ws, use_hash, status_out, map, count);
  HIP_TRY(hipGetLastError());
  return 0;
}
