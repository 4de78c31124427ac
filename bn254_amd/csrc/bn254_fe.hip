// Translation unit of libbn254hip.so for the FINAL EXPONENTIATION of the lane-pair layout (bn254_fp2_pair.h): one verify /
// pairing per lane pair, k_final_exp_pair.  A code object of its own so that the placement of the Miller loops of
// bn254_pair.hip (62 KB of loop body against a 64 KB instruction cache) does not move when this kernel changes: with both
// in one object the same Miller source ran 1.3-2 % slower after the accumulator machine replaced the chains here
// (profiles/r03_b_ab_fe_machine.log).
// Replaces, for ECDSA::verify (/root/reference/src/ecdsa.rs:57-59), the final exponentiation and the == Gt::one() test.
#include <hip/hip_runtime.h>

#define BN_SPLIT_FP2 1
#if !defined(BN_NO_ASM_CSQR)
#define BN_ASM_CSQR_UNIT 1          // the accumulator machine of this unit runs on lane pairs with its accumulator in LDS (bn254_pairing.h: BN_FE_CSQR)
#if !defined(BN_NO_ASM_MUL)
#define BN_ASM_MUL 1                // ... and its MUL opcode is the generated block too (bn254_pairing.h: BN_FE_MUL)
#endif
#endif
#ifndef BN_PAIR_NO_SQR_DPP_ASM
#define BN_PAIR_SQR_DPP_ASM 1      // role prologue of the Fq2 squaring with folded DPP operands (bn254_fp2_pair.h)
#endif
#ifndef BN_PAIR_CALL_FP12_HOT
#define BN_INLINE_FP12_HOT 1       // fp12_sqr / fp12_mul_line2 inlined into the Miller loops (bn254_field.h: BN_DEVH)
#endif
// (BN_INLINE_MILLER — the Miller loops inlined into their kernels — is an A/B knob only: measured 8.3 instead of 5.8 ms)
#ifndef BN_PAIR_CALL_FE_HOT
#define BN_INLINE_FE_HOT 1         // fp12_cyclotomic_sqr / fp12_mul inlined into the loop of fp12_pow_u (bn254_field.h: BN_DEVF)
#endif
// Measured (same box): Miller 8.8-8.95 -> 8.2-8.3 ms, final exponentiation 6.5-6.6 -> 6.4-6.45 ms per 65 536.
#ifndef BN_PRIO_SHIFT
#define BN_PRIO_SHIFT 1            // priority changes every 2^shift steps, cycle of 4 levels (0..3 measured: 0 and 1 best)
#endif
#define BN_SET_STEP_PRIORITY(step)                                                        \
  do {                                                                                    \
    if (((step) & ((1 << BN_PRIO_SHIFT) - 1)) == 0) {                                     \
      int q_ = ((step) >> BN_PRIO_SHIFT) & 3;                                             \
      if (q_ == 0) __builtin_amdgcn_s_setprio(3);                                         \
      else if (q_ == 1) __builtin_amdgcn_s_setprio(2);                                    \
      else if (q_ == 2) __builtin_amdgcn_s_setprio(1);                                    \
      else __builtin_amdgcn_s_setprio(0);                                                 \
    }                                                                                     \
  } while (0)
#define bn254 bn254_fe   // own namespace: the Fq2 / Fq12 types differ from the one-lane translation unit
#include "bn254_pairing.h"
#include "bn254_codec_g2.h"

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
static_assert(sizeof(Fp12PairSlot) == (6 * BN_LIMBS + 1) * 4 && ((6 * BN_LIMBS + 1) & 1), "LDS slot: 6 x 9 limbs + 1 pad word (odd stride: conflict-free)");

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

// 32 big-endian bytes of the canonical value (4-byte aligned destination)
__device__ __forceinline__ void store_fp_be(uint8_t* b, const Fp& a) {
  U256 x = fp_to_u256(a);
  uint32_t* w = (uint32_t*)b;
#pragma unroll
  for (int k = 0; k < 8; ++k) w[k] = __builtin_bswap32(x.w[7 - k]);
}
// item i: product of the k Miller values at workspace indices base + i*item_stride + j*pair_stride, final
// exponentiation, comparison with one (status) and / or the canonical Gt bytes (each lane writes the 32-byte halves
// of its role).  Same contract as k_final_exp of bn254_hip.hip.
KERNEL_PAIR void k_final_exp_pair(size_t n, size_t k, size_t item_stride, size_t pair_stride, Ws ws, int use_hash, uint8_t* gt_out,
                                  uint8_t* status_out, int raw_only, size_t base, const uint32_t* map, const uint32_t* count) {
  size_t i = ((size_t)blockIdx.x * BN_PAIR_WG + threadIdx.x) >> 1;
  if (i >= n) return;
  if (count && i >= *count) return;                 // a device-side item count (with or without an index map)
  if (map) i = map[i];
  __shared__ Fp12PairSlot lds_acc[BN_PAIR_WG];
#if defined(BN_PAIR_FE_CHAINS)
  Fp12 f;
#else
  Fp12& f = lds_acc[threadIdx.x].v;                   // the accumulator of the machine below
#endif
  ws_load_f12_own(ws, base + i * item_stride, f);
  uint8_t st = ws_byte(ws, BY_ST_DECODE, base + i * item_stride);
  for (size_t j = 1; j < k; ++j) {
    size_t idx = base + i * item_stride + j * pair_stride;
    Fp12 g;
    ws_load_f12_own(ws, idx, g);
    fp12_mul(f, f, g);
    uint8_t sj = ws_byte(ws, BY_ST_DECODE, idx);
    if (st == ST_OK) st = sj;
  }
  if (st == ST_OK && use_hash) st = ws_byte(ws, BY_ST_HASH, i);
  BN_CLK_BEGIN(ws);
  if (!raw_only) {
#if defined(BN_PAIR_FE_CHAINS)
    // A/B knob: the chains as straight-line code with real Fq12 calls (rounds 1-2)
    if (gt_out) final_exponentiation<true>(f, f, lds_acc[threadIdx.x].v);
    else final_exponentiation_check<true>(f, f, lds_acc[threadIdx.x].v);
#else
    // canonical Gt bytes need the exact exponent; the == one test alone takes the shorter chain.  Both are programs of the
    // accumulator machine (bn254_pairing.h: fe_machine): ONE interpreter loop, so that every Fq12 routine is inlined once
    Fp12 slot[BN_FE_EXACT_SLOTS > BN_FE_CHECK_SLOTS ? BN_FE_EXACT_SLOTS : BN_FE_CHECK_SLOTS];
    fe_machine(f, slot, gt_out ? C_FE_EXACT : C_FE_CHECK);
#endif
  }
  BN_CLK_END(ws, 1);
  const unsigned role = threadIdx.x & 1u;
  if (gt_out) {
    const Fp2* c[6] = {&f.c0.c0, &f.c0.c1, &f.c0.c2, &f.c1.c0, &f.c1.c1, &f.c1.c2};
    for (int e = 0; e < 6; ++e) store_fp_be(gt_out + 384 * i + 64 * e + 32 * role, c[e]->c[0]);
  }
  const bool one = fp12_is_one(f);   // combined over the pair
  if (status_out && role == 0) status_out[i] = st != ST_OK ? st : (one ? (uint8_t)ST_OK : (uint8_t)ST_VERIFICATION_FAILED);
}

int bn254_pair_final_exp(size_t n, Ws ws, int use_hash, uint8_t* status_out, const uint32_t* map, const uint32_t* count, hipStream_t s, size_t base) {
  k_final_exp_pair<<<(unsigned)((2 * n + BN_PAIR_WG - 1) / BN_PAIR_WG), BN_PAIR_WG, 0, s>>>(n, 1, 1, 1, ws, use_hash, nullptr, status_out, 0, base, map, count);
  HIP_TRY(hipGetLastError());
  return 0;
}
int bn254_pair_final_exp_product(size_t n, size_t k, Ws ws, uint8_t* gt_out, uint8_t* status_out, int raw_only, hipStream_t s) {
  k_final_exp_pair<<<(unsigned)((2 * n + BN_PAIR_WG - 1) / BN_PAIR_WG), BN_PAIR_WG, 0, s>>>(n, k, k, 1, ws, 0, gt_out, status_out, raw_only, 0, nullptr, nullptr);
  HIP_TRY(hipGetLastError());
  return 0;
}
