// Measurement-only translation unit of libbn254hip.so (pair layout): kept apart from bn254_pair.hip so that the code object of the
// Miller kernels does not change with it (their loop body is ~62 KB against a 64 KB instruction cache: DESIGN.md section 3).
#include <hip/hip_runtime.h>

#define BN_SPLIT_FP2 1
#if !defined(BN_NO_ASM_CSQR)
#define BN_ASM_CSQR_UNIT 1          // the accumulator machine of this unit runs on lane pairs with its accumulator in LDS (bn254_pairing.h: BN_FE_CSQR)
#if !defined(BN_NO_ASM_MUL)
#define BN_ASM_MUL 1                // ... and its MUL opcode is the generated block too (bn254_pairing.h: BN_FE_MUL)
#endif
#endif
#define BN_PAIR_SQR_DPP_ASM 1
#define BN_PRIO_SHIFT 1
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
#define BN_INLINE_FP12_HOT 1        // the same inlining choices as bn254_fe.hip, so that fe_machine below is the code k_final_exp_pair runs
#define BN_INLINE_FE_HOT 1
#define bn254 bn254_probe   // own namespace, as in the other pair-layout translation units
#include "bn254_pairing.h"

using namespace bn254;

#include "bn254_ws.h"

#define BN_PAIR_WG 256
#define KERNEL_PAIR __global__ __launch_bounds__(BN_PAIR_WG) __attribute__((amdgpu_waves_per_eu(2, 2)))
struct Fp12PairSlot { Fp12 v; int32_t pad; };
__device__ __forceinline__ Fp2 ws_load_fp2_own(const Ws& ws, int plane_re, size_t i) {
  Fp2 r;
  r.c[0] = ws_load_fp(ws, plane_re + (int)(threadIdx.x & 1u), i);
  return r;
}

// Measurement only (bn254_probe_leaf_floor): the PRODUCT CALLS of one verify's Miller loop (or final exponentiation) and nothing else,
// on the same launch shape, with f's LDS slot allocated and the priority cycle running.  No twist point, no tower additions, no carries, no LDS traffic: what is left is the
// leaves themselves plus ~9 argument moves per call, i.e. a floor for ANY way of writing the code around them.
// mode 0: the Miller loop's leaves — 87 x (37 dual products + 5 squarings + 4 scalings) = 3 219 / 435 / 348 per lane against the loop's
//         3 194 / 430 / 348;  mode 1: the final exponentiation's — 189 x (5 dual + 9 squarings) = 945 / 1 701 against 975 / 1 714
__device__ __noinline__ void leaf_floor_loop(Fp2& x, const Fp2& y, const Fp& k, int steps, int n_dual, int n_sqr, int n_scale) {
  for (int d = 0; d < steps; ++d) {
    BN_SET_STEP_PRIORITY(d);
#pragma unroll 1
    for (int j = 0; j < n_dual; ++j) x = fp2_mul(x, y);
#pragma unroll 1
    for (int j = 0; j < n_sqr; ++j) x = fp2_sqr(x);
#pragma unroll 1
    for (int j = 0; j < n_scale; ++j) x = fp2_mul_fp(x, k);
  }
}
// modes 4 / 5: the product counts of modes 0 / 1 spread over FOUR independent chains.  In leaf_floor_loop every product waits for its
// predecessor's last digits at its own first instructions; with one wave per SIMD (batches of 16 385 .. 32 768 verifies) nothing hides that
// and the Miller kernel, whose consecutive products are mostly independent, ran 10 % FASTER than that "floor" (DESIGN.md section 9.6).
// A floor must not lose to what it bounds: the figure to quote is min(dependent, independent).
template <int CH, int N_DUAL, int N_SQR, int N_SCALE> __device__ __forceinline__ void leaf_floor_chains(Fp2 (&x)[CH], const Fp2& y, const Fp& k, int steps) {   // inlined: the chains stay in registers
  for (int d = 0; d < steps; ++d) {
    BN_SET_STEP_PRIORITY(d);
#pragma unroll 1
    for (int j = 0; j < N_DUAL / CH; ++j) {
#pragma unroll
      for (int c = 0; c < CH; ++c) x[c] = fp2_mul(x[c], y);
    }
#pragma unroll
    for (int j = 0; j < N_DUAL % CH; ++j) x[j] = fp2_mul(x[j], y);
#pragma unroll 1
    for (int j = 0; j < N_SQR / CH; ++j) {
#pragma unroll
      for (int c = 0; c < CH; ++c) x[c] = fp2_sqr(x[c]);
    }
#pragma unroll
    for (int j = 0; j < N_SQR % CH; ++j) x[j] = fp2_sqr(x[j]);
#pragma unroll 1
    for (int j = 0; j < N_SCALE / CH; ++j) {
#pragma unroll
      for (int c = 0; c < CH; ++c) x[c] = fp2_mul_fp(x[c], k);
    }
#pragma unroll
    for (int j = 0; j < N_SCALE % CH; ++j) x[j] = fp2_mul_fp(x[j], k);
  }
}
// mode 2: 87 x 37 dual products with the leaf INLINED into the loop (the body of fp_pair_mul_impl, bn254_fp2_pair.h) — no call, no return,
// no wait at a function entry, no argument moves: what the calling convention itself costs per product
__device__ __forceinline__ Fp2 fp2_mul_inlined(const Fp2& a, const Fp2& b) {
  const int32_t one = 1 - (int32_t)(threadIdx.x & 1u), mask = -one;
  int32_t ao[BN_LIMBS], ap[BN_LIMBS], x[BN_LIMBS], y[BN_LIMBS], r[BN_LIMBS];
#pragma unroll
  for (int i = 0; i < BN_LIMBS; ++i) {
    ao[i] = a.c[0].v[i];
    ap[i] = bn_partner_word(a.c[0].v[i]);
    x[i] = bn_pair_re_word(b.c[0].v[i]);
    y[i] = (bn_pair_im_word(b.c[0].v[i]) ^ mask) + one;
  }
  BN_MONT_DUAL_BODY(ao, x, ap, y, r);
  Fp2 z;
#pragma unroll
  for (int i = 0; i < BN_LIMBS; ++i) z.c[0].v[i] = r[i];
  return z;
}
__device__ __noinline__ void leaf_floor_loop_inlined(Fp2& x, const Fp2& y) {
  for (int d = 0; d < 87; ++d) {
    BN_SET_STEP_PRIORITY(d);
#pragma unroll 1
    for (int j = 0; j < 37; ++j) x = fp2_mul_inlined(x, y);
  }
}
KERNEL_PAIR void k_leaf_floor_pair(size_t n, Ws ws, int mode) {
  size_t i = ((size_t)blockIdx.x * BN_PAIR_WG + threadIdx.x) >> 1;
  if (i >= n) return;
  __shared__ Fp12PairSlot lds_f[BN_PAIR_WG];
  Fp2 x = ws_load_fp2_own(ws, PL_QX0, i);
  const Fp2 y = ws_load_fp2_own(ws, PL_QY0, i);
  const Fp k = ws_load_fp(ws, PL_P1X, i);
  lds_f[threadIdx.x].v.c0.c0 = x;
  BN_CLK_BEGIN(ws);
  if (mode == 0) leaf_floor_loop(x, y, k, 87, 37, 5, 4);
  else if (mode == 1) leaf_floor_loop(x, y, k, 189, 5, 9, 0);
  else if (mode == 2) leaf_floor_loop_inlined(x, y);
  else if (mode == 4 || mode == 5) {
    Fp2 c4[4] = {x, fp2_add(x, y), fp2_sub(x, y), fp2_add(x, x)};
    if (mode == 4) leaf_floor_chains<4, 37, 5, 4>(c4, y, k, 87);
    else leaf_floor_chains<4, 5, 9, 0>(c4, y, k, 189);
    x = fp2_add(fp2_add(c4[0], c4[1]), fp2_add(c4[2], c4[3]));
  } else if (mode == 6) {                           // controls for mode 4: the same inlined loop with ONE chain ...
    Fp2 c1[1] = {x};
    leaf_floor_chains<1, 37, 5, 4>(c1, y, k, 87);
    x = c1[0];
  } else if (mode == 7) {                           // ... and with two
    Fp2 c2[2] = {x, fp2_add(x, y)};
    leaf_floor_chains<2, 37, 5, 4>(c2, y, k, 87);
    x = fp2_add(c2[0], c2[1]);
  }
  else leaf_floor_loop(x, y, k, 87, 37, 0, 0);          // mode 3: the same 3 219 dual products as mode 2, called
  BN_CLK_END(ws, 2);                               // the probe's own clock slot
  x = fp2_add(x, lds_f[threadIdx.x].v.c0.c0);
  ws_store_fp(ws, PL_F0 + (int)(threadIdx.x & 1u), i, x.c[0]);
}
// Measurement only (bn254_probe_fe_program): the accumulator machine of the final exponentiation (bn254_pairing.h: fe_machine, the very
// interpreter of k_final_exp_pair) on a program handed over in global memory — programs of ONE operation kind time that operation in
// place (LDS accumulator, slot file in the private segment, two waves per SIMD): the measured split of the kernel by routine.
KERNEL_PAIR void k_fe_program_pair(size_t n, Ws ws, const unsigned char (*prog)[2]) {
  size_t i = ((size_t)blockIdx.x * BN_PAIR_WG + threadIdx.x) >> 1;
  if (i >= n) return;
  __shared__ Fp12PairSlot lds_acc[BN_PAIR_WG];
  Fp12& f = lds_acc[threadIdx.x].v;
  Fp2* c[6] = {&f.c0.c0, &f.c0.c1, &f.c0.c2, &f.c1.c0, &f.c1.c1, &f.c1.c2};
#pragma unroll
  for (int k = 0; k < 6; ++k) *c[k] = ws_load_fp2_own(ws, PL_F0 + 2 * k, i);
  Fp12 slot[BN_FE_EXACT_SLOTS > BN_FE_CHECK_SLOTS ? BN_FE_EXACT_SLOTS : BN_FE_CHECK_SLOTS];
#pragma unroll
  for (int k = 0; k < (int)(sizeof(slot) / sizeof(slot[0])); ++k) slot[k] = f;      // every slot readable
  fe_machine(f, slot, prog);
  Fp2 x = fp2_add(fp2_add(f.c0.c0, f.c0.c1), fp2_add(f.c1.c0, f.c1.c2));
  ws_store_fp(ws, PL_HASHX + (int)(threadIdx.x & 1u), i, x.c[0]);                    // keep the result alive (a plane no later kernel reads before rewriting it)
}
int bn254_pair_fe_program(size_t n, Ws ws, const unsigned char* prog, hipStream_t s) {
  k_fe_program_pair<<<(unsigned)((2 * n + BN_PAIR_WG - 1) / BN_PAIR_WG), BN_PAIR_WG, 0, s>>>(n, ws, (const unsigned char (*)[2])prog);
  HIP_TRY(hipGetLastError());
  return 0;
}
int bn254_pair_leaf_floor(size_t n, Ws ws, hipStream_t s, int mode) {
  k_leaf_floor_pair<<<(unsigned)((2 * n + BN_PAIR_WG - 1) / BN_PAIR_WG), BN_PAIR_WG, 0, s>>>(n, ws, mode);
  HIP_TRY(hipGetLastError());
  return 0;
}
