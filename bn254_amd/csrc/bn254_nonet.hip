// Device translation unit of libbn254hip.so: the FINAL EXPONENTIATION of ECDSA::verify (/root/reference/src/ecdsa.rs:57-59) for the
// SMALLEST batches, one verify per NONET of lane pairs (18 lanes; three verifies per wave, 12 per 256-lane workgroup).
//
// A wave that has its SIMD to itself issues a multiplier-class instruction only every ~10 cycles, so for a batch that cannot fill the chip
// latency is instructions per LANE.  The octet layout (bn254_trio.hip) spreads an Fq12 operation over four lane pairs: a cyclotomic
// squaring still costs three squarings in a row per pair, an Fq12 product six products in a row.  Here the NINE squarings of a
// Granger-Scott squaring run in nine pairs at once (one leaf call), and the 18 products of a Karatsuba Fq12 product in two rounds of
// nine; everything linear is distributed as well (the six outputs of a squaring in six pairs, the nine Fq6 coefficients of a product in
// nine, its six outputs in six) and exchanged through LDS inside the wave (no barrier: a wave's LDS instructions execute in order).
// All pairs of a wave run ONE instruction stream — which coefficient, product or output a pair handles is data (small index tables),
// never control flow.
//
// Formulas and carry sites are the pair layout's own (bn254_field.h: fp12_mul_body, fp6_mul<S>, fp12_cyclotomic_sqr_body, sites 20..43,
// 170..179) coefficient by coefficient; where the three Fq6 products of a multiplication have different site modes the strictest is
// applied to all.  The schedule is written once, in bn254_nonet.h, against a "box": the kernel below instantiates it on LDS, the CPU
// suite on plain arrays — equal to fe_machine coefficient for coefficient on every verify case and PROVEN under the bound tracker
// (tests/test_pair_layout.py::test_nonet_schedule_matches_fe_machine, ::test_nonet_schedule_bounds_hold).  Everything else of the chain (Frobenius, conjugation, the inversion of the easy part, slot
// moves) runs replicated in every pair, exactly as in the pair layout.  Same status bytes as every other layout:
// test_octet_and_pair_layouts_agree_with_oracle, the soak.
#include <hip/hip_runtime.h>

#define BN_SPLIT_FP2 1
#define BN_PAIR_SQR_DPP_ASM 1
#define BN_INLINE_FP12_HOT 1
#define BN_INLINE_FE_HOT 1
#define bn254 bn254_nonet   // own namespace: the pair layout's types and routines
#include "bn254_pairing.h"
#include "bn254_nonet.h"

using namespace bn254;

#include "bn254_ws.h"

#define BN_NONET_WG 256
#define BN_NONET_LANES 18                         // lanes per verify: nine lane pairs
#define BN_NONET_PER_WAVE 3                       // verifies per wave (54 of 64 lanes; lanes 54..63 follow along on copies)
#define BN_NONET_PER_WG (BN_NONET_PER_WAVE * BN_NONET_WG / BN_WAVE)
#define KERNEL_NONET __global__ __launch_bounds__(BN_NONET_WG) __attribute__((amdgpu_waves_per_eu(1, 1)))

// dynamic LDS of the kernel (words).  The nine pairs of a verify hold the SAME accumulator and the same slot file, so both exist once per
// verify and role (6 x 9 limbs + 1 pad word each), read by all pairs at once (one address: a broadcast) and written by the pair that
// forms a coefficient; the replicated operations (conjugation, Frobenius, inversion, slot moves) write identical words from every pair.
// Beside them per verify the exchange areas of products and Fq6 coefficients ([entry][role][9 limbs]), and one block of zeros.
#define NN_SLOT (6 * BN_LIMBS + 1)
#define NN_NSLOTS BN_FE_CHECK_SLOTS
#define NN_ACC_OFF 0
#define NN_FILE_OFF (NN_ACC_OFF + BN_NONET_PER_WG * 2 * NN_SLOT)
#define NN_XP_OFF (NN_FILE_OFF + BN_NONET_PER_WG * NN_NSLOTS * 2 * NN_SLOT)
#define NN_XP_STRIDE (18 * 2 * BN_LIMBS)
#define NN_X1_OFF (NN_XP_OFF + BN_NONET_PER_WG * NN_XP_STRIDE)
#define NN_X1_STRIDE (9 * 2 * BN_LIMBS)
#define NN_ZERO_OFF (NN_X1_OFF + BN_NONET_PER_WG * NN_X1_STRIDE)
#define NN_LDS_WORDS (NN_ZERO_OFF + 16)
static_assert(NN_LDS_WORDS * sizeof(int32_t) <= 160 * 1024, "nonet kernel: accumulators + exchange areas exceed the 160 KB of LDS of a gfx950 CU");

extern __shared__ int32_t nn_lds[];

#define NN_FENCE() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

// The layout itself — index tables, what a pair needs to know about its place, the phase functions of the two distributed operations —
// lives in bn254_nonet.h, written against a "box"; this is the device box: the workgroup's LDS, references = word offsets of the lane's
// verify and role.  tests/hostsim runs the same phase functions on a host box, also under the bound tracker.
struct NnLdsBox {
  typedef unsigned Ref;
  typedef unsigned Rel;
  typedef unsigned Base;
  unsigned acc_, file_, xp_, x1_;          // word offsets of the verify's accumulator, slot 0 of its file, its exchange areas — for this lane's role
  __device__ __forceinline__ Ref coef(unsigned idx) const { return idx < 6 ? acc_ + idx * BN_LIMBS : (unsigned)NN_ZERO_OFF; }
  __device__ __forceinline__ Ref xp(unsigned q) const { return xp_ + q * 2 * BN_LIMBS; }
  __device__ __forceinline__ Ref x1(unsigned q) const { return q < 9 ? x1_ + q * 2 * BN_LIMBS : (unsigned)NN_ZERO_OFF; }
  __device__ __forceinline__ Ref zero() const { return (unsigned)NN_ZERO_OFF; }
  __device__ __forceinline__ static Rel rel(unsigned idx) { return idx * BN_LIMBS; }
  __device__ __forceinline__ static Ref at(Base base, Rel r) { return base + r; }
  __device__ __forceinline__ Fp2 get(Ref off) const {
    Fp2 r;
#pragma unroll
    for (int i = 0; i < BN_LIMBS; ++i) r.c[0].v[i] = nn_lds[off + i];
    return r;
  }
  __device__ __forceinline__ void put(Ref off, const Fp2& x) const {
#pragma unroll
    for (int i = 0; i < BN_LIMBS; ++i) nn_lds[off + i] = x.c[0].v[i];
  }
};
struct NnDevLane : NnLane<NnLdsBox> {
  unsigned role, vslot;         // real / imaginary part, verify slot within the workgroup
  bool mul_writer;              // publishes its product of a multiplication round (wide form: the lanes of the second nine pairs too)
  NnLdsBox bx;
};
// WIDE form (batches that leave three quarters of the chip idle anyway: one verify per WAVE): EIGHTEEN lane pairs per verify, so that the 18
// products of a multiplication are ONE round — pair 9 + p takes product 9 + p, i.e. the second round's entry of pair p, and otherwise
// follows pair p without publishing anything (same values, same phase functions: the host emulation and its bound proof cover both forms).
template <bool WIDE> __device__ __forceinline__ NnDevLane nn_lane() {
  constexpr unsigned LANES = WIDE ? 2 * BN_NONET_LANES : BN_NONET_LANES, PER_WAVE = WIDE ? 1 : BN_NONET_PER_WAVE;
  NnDevLane L;
  const unsigned l = threadIdx.x & (BN_WAVE - 1), w = threadIdx.x / BN_WAVE;
  const unsigned v = l / LANES;
  const bool in_verify = v < PER_WAVE;
  const unsigned p = (l % LANES) >> 1;                         // 0..8, wide: 0..17
  const bool second = p >= BN_NONET_PAIRS;
  L.role = l & 1u;
  L.vslot = w * PER_WAVE + (in_verify ? v : PER_WAVE - 1);
  L.bx.acc_ = NN_ACC_OFF + (L.vslot * 2 + L.role) * NN_SLOT;
  L.bx.file_ = NN_FILE_OFF + (L.vslot * NN_NSLOTS * 2 + L.role) * NN_SLOT;          // slot k: + k * 2 * NN_SLOT
  L.bx.xp_ = NN_XP_OFF + L.vslot * NN_XP_STRIDE + L.role * BN_LIMBS;
  L.bx.x1_ = NN_X1_OFF + L.vslot * NN_X1_STRIDE + L.role * BN_LIMBS;
  nn_lane_roles<NnLdsBox>(L, L.bx, second ? p - BN_NONET_PAIRS : p, in_verify && !second);
  L.mul_writer = in_verify;
  if (WIDE && second) {
#pragma unroll
    for (unsigned j = 0; j < 4; ++j) { L.m_rel[0][j] = L.m_rel[1][j]; L.m_zero[0][j] = L.m_zero[1][j]; }
    L.m_pub[0] = L.m_pub[1];
  }
  return L;
}
// the two distributed operations: phase, publish, phase, publish — between wavefront-scope fences (one wave: its LDS instructions execute
// in order; the fences keep the compiler from moving them)
__device__ __forceinline__ void nn_csqr(NnDevLane& L) {
  {
    const Fp2 sq = nn_csqr_square(L, L.bx);
    NN_FENCE();
    if (L.writer) L.bx.put(L.sq_pub, sq);
    NN_FENCE();
  }
  const Fp2 o = nn_csqr_output(L, L.bx);
  NN_FENCE();                                     // every pair has read the old coefficients it needs
  if (L.publishes_out) L.bx.put(L.so_coef, o);    // straight into the verify's accumulator
  NN_FENCE();
}
template <bool WIDE> __device__ __forceinline__ void nn_mul(NnDevLane& L, unsigned bslot) {     // bslot: word offset of the second operand (a slot of the file), this lane's role
#pragma unroll
  for (unsigned r = 0; r < (WIDE ? 1u : 2u); ++r) {  // two rounds of nine products; wide form: one of eighteen
    const Fp2 pr = nn_mul_product(L, L.bx, L.bx.acc_, bslot, r);
    NN_FENCE();
    if (L.mul_writer) L.bx.put(L.m_pub[r], pr);
  }
  NN_FENCE();
  {
    const Fp2 c = nn_mul_level1(L, L.bx);
    NN_FENCE();
    if (L.writer) L.bx.put(L.l1_pub, c);
    NN_FENCE();
  }
  const Fp2 o = nn_mul_level2(L, L.bx);
  NN_FENCE();
  if (L.publishes_out) L.bx.put(L.out_coef, o);   // straight into the verify's accumulator (all operand reads are behind us)
  NN_FENCE();
}

// the accumulator machine of bn254_pairing.h (fe_machine) with the two hot operations distributed over the nine pairs; accumulator and
// slot file are the verify's shared copies in LDS.  LOAD / STORE / CONJ / FROB / INV are REPLICATED: every lane of the verify — and the
// follow-along lanes 54..63 on the wave's last verify — read-modify-writes the shared copy with identical words in lockstep (the
// invariant stated in bn254_nonet.h); `writer` / `publishes_out` gate only the publish steps of CSQR / MUL.
template <bool WIDE> __device__ __forceinline__ void nn_machine(NnDevLane& L, const unsigned char (*prog)[2]) {
  Fp12& acc = *(Fp12*)(nn_lds + L.bx.acc_);        // the first 54 words of a slot are an Fp12 in memory order
  BN_ASSUME_LDS(&acc);
#pragma clang loop unroll(disable)
  for (int pc = 0;; ++pc) {
    const int op = prog[pc][0], arg = prog[pc][1];
    if (op == FE_END) break;
    const unsigned sl = L.bx.file_ + (unsigned)arg * 2 * NN_SLOT;
    Fp12& slot = *(Fp12*)(nn_lds + sl);
    BN_ASSUME_LDS(&slot);
    NN_FENCE();
    switch (op) {
      case FE_LOAD: acc = slot; break;                       // identical words from every pair
      case FE_STORE: slot = acc; break;
      case FE_CSQR: nn_csqr(L); break;
      case FE_MUL: nn_mul<WIDE>(L, sl); break;
      case FE_CONJ: fp6_neg(acc.c1, acc.c1); break;
      case FE_FROB: fp12_frob_body(acc, acc, arg); break;
      default: fp12_inv(acc, acc); break;
    }
  }
  NN_FENCE();
}

// 32 big-endian bytes of the canonical value (4-byte aligned destination)
__device__ __forceinline__ void nn_store_fp_be(uint8_t* b, const Fp& a) {
  U256 x = fp_to_u256(a);
  uint32_t* w = (uint32_t*)b;
#pragma unroll
  for (int k = 0; k < 8; ++k) w[k] = __builtin_bswap32(x.w[7 - k]);
}
// item i: (PRODUCT: the product of the k Miller values at workspace indices i k .. i k + k - 1, then) the final exponentiation — program
// C_FE_CHECK for a status alone, C_FE_EXACT when canonical Gt bytes are wanted (bn254_batch_pairing*: same contract as k_final_exp_pair)
template <bool WIDE, bool PRODUCT> __device__ __forceinline__ void final_exp_nonet_body(size_t n, size_t k, const Ws& ws, int use_hash, uint8_t* gt_out, uint8_t* status_out) {
  constexpr unsigned PER_WG = WIDE ? BN_NONET_WG / BN_WAVE : BN_NONET_PER_WG;
  NnDevLane L = nn_lane<WIDE>();
  size_t i = (size_t)blockIdx.x * PER_WG + L.vslot;
  const bool live = L.writer && i < n;
  if (i >= n) i = n - 1;                                 // lanes without a verify of their own follow along on the last one
  const size_t first = PRODUCT ? i * k : i;
  if (threadIdx.x < 16) nn_lds[NN_ZERO_OFF + threadIdx.x] = 0;
  if (L.publishes_out) {                                 // pair k < 6 brings in coefficient k of the Miller value
    const Fp c = ws_load_fp(ws, PL_F0 + 2 * (int)L.pair + (int)L.role, first);
#pragma unroll
    for (int j = 0; j < BN_LIMBS; ++j) nn_lds[L.bx.acc_ + L.pair * BN_LIMBS + j] = c.v[j];
  }
  uint8_t st = ws_byte(ws, BY_ST_DECODE, first);
  if (st == ST_OK && use_hash) st = ws_byte(ws, BY_ST_HASH, i);
  __syncthreads();                                       // the zero block, the accumulators
  if (PRODUCT) {
    for (size_t j = 1; j < k; ++j) {                     // the other Miller values of the item through slot 0 (every program stores a slot before it reads it)
      if (L.publishes_out) {
        const Fp c = ws_load_fp(ws, PL_F0 + 2 * (int)L.pair + (int)L.role, first + j);
#pragma unroll
        for (int q = 0; q < BN_LIMBS; ++q) nn_lds[L.bx.file_ + L.pair * BN_LIMBS + q] = c.v[q];
      }
      NN_FENCE();
      nn_mul<WIDE>(L, L.bx.file_);
      const uint8_t sj = ws_byte(ws, BY_ST_DECODE, first + j);
      if (st == ST_OK) st = sj;
    }
  }
  nn_machine<WIDE>(L, (PRODUCT && gt_out) ? C_FE_EXACT : C_FE_CHECK);
  Fp12 f;
  {
    Fp2* c[6] = {&f.c0.c0, &f.c0.c1, &f.c0.c2, &f.c1.c0, &f.c1.c1, &f.c1.c2};
#pragma unroll
    for (int q = 0; q < 6; ++q) *c[q] = L.bx.get(L.bx.acc_ + q * BN_LIMBS);
  }
  if (PRODUCT && gt_out && live && L.publishes_out) nn_store_fp_be(gt_out + 384 * i + 64 * L.pair + 32 * L.role, L.bx.get(L.bx.acc_ + L.pair * BN_LIMBS).c[0]);
  const bool one = fp12_is_one(f);                       // combined over the pair
  if (status_out && live && L.pair == 0 && L.role == 0) status_out[i] = st != ST_OK ? st : (one ? (uint8_t)ST_OK : (uint8_t)ST_VERIFICATION_FAILED);
}
KERNEL_NONET void k_final_exp_nonet(size_t n, Ws ws, int use_hash, uint8_t* status_out) { final_exp_nonet_body<false, false>(n, 1, ws, use_hash, nullptr, status_out); }
KERNEL_NONET void k_final_exp_nonet_wide(size_t n, Ws ws, int use_hash, uint8_t* status_out) { final_exp_nonet_body<true, false>(n, 1, ws, use_hash, nullptr, status_out); }
KERNEL_NONET void k_final_exp_nonet_product(size_t n, size_t k, Ws ws, uint8_t* gt_out, uint8_t* status_out) { final_exp_nonet_body<true, true>(n, k, ws, 0, gt_out, status_out); }

bool bn254_nonet_fits_device() {
  int blocks = 0;
  hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, k_final_exp_nonet, BN_NONET_WG, NN_LDS_WORDS * sizeof(int32_t));
  if (e != hipSuccess) { (void)hipGetLastError(); return true; }
  return blocks > 0;
}
// `wide`: one verify per wave on eighteen lane pairs (a multiplication's products in one round) — for batches of up to one verify per SIMD
int bn254_nonet_final_exp(size_t n, Ws ws, int use_hash, uint8_t* status_out, hipStream_t s, int wide) {
  if (wide) {
    const unsigned per_wg = BN_NONET_WG / BN_WAVE;
    k_final_exp_nonet_wide<<<(unsigned)((n + per_wg - 1) / per_wg), BN_NONET_WG, NN_LDS_WORDS * sizeof(int32_t), s>>>(n, ws, use_hash, status_out);
  } else {
    const unsigned grid = (unsigned)((n + BN_NONET_PER_WG - 1) / BN_NONET_PER_WG);
    k_final_exp_nonet<<<grid, BN_NONET_WG, NN_LDS_WORDS * sizeof(int32_t), s>>>(n, ws, use_hash, status_out);
  }
  HIP_TRY(hipGetLastError());
  return 0;
}
// bn254_batch_pairing* for small batches: items of k pairs each, eighteen lane pairs per item (n <= NONET_WIDE_MAX_BATCH)
int bn254_nonet_final_exp_product(size_t n, size_t k, Ws ws, uint8_t* gt_out, uint8_t* status_out, hipStream_t s) {
  const unsigned per_wg = BN_NONET_WG / BN_WAVE;
  k_final_exp_nonet_product<<<(unsigned)((n + per_wg - 1) / per_wg), BN_NONET_WG, NN_LDS_WORDS * sizeof(int32_t), s>>>(n, k, ws, gt_out, status_out);
  HIP_TRY(hipGetLastError());
  return 0;
}
