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
// applied to all (a carry or a weak reduction never changes a value mod q and only tightens limbs), so the bound proofs of
// tests/test_pair_layout.py cover these flows.  Everything else of the chain (Frobenius, conjugation, the inversion of the easy part, slot
// moves) runs replicated in every pair, exactly as in the pair layout.  Same status bytes as every other layout:
// test_octet_and_pair_layouts_agree_with_oracle, the soak.
#include <hip/hip_runtime.h>

#define BN_SPLIT_FP2 1
#define BN_PAIR_SQR_DPP_ASM 1
#define BN_INLINE_FP12_HOT 1
#define BN_INLINE_FE_HOT 1
#define bn254 bn254_nonet   // own namespace: the pair layout's types and routines
#include "bn254_pairing.h"

using namespace bn254;

#include "bn254_ws.h"

#define BN_NONET_WG 256
#define BN_NONET_LANES 18                         // lanes per verify: nine lane pairs
#define BN_NONET_PER_WAVE 3                       // verifies per wave (54 of 64 lanes; lanes 54..63 follow along on copies)
#define BN_NONET_PER_WG (BN_NONET_PER_WAVE * BN_NONET_WG / BN_WAVE)
#define KERNEL_NONET __global__ __launch_bounds__(BN_NONET_WG) __attribute__((amdgpu_waves_per_eu(1, 1)))

// dynamic LDS of the kernel (words): per lane the accumulator and the second operand of a multiplication (6 x 9 limbs + 1 pad word:
// odd stride), per verify the exchange areas of products / Fq6 coefficients / outputs ([entry][role][9 limbs]), one block of zeros
#define NN_SLOT (6 * BN_LIMBS + 1)
#define NN_ACC_OFF 0
#define NN_BOP_OFF (NN_ACC_OFF + BN_NONET_WG * NN_SLOT)
#define NN_XP_OFF (NN_BOP_OFF + BN_NONET_WG * NN_SLOT)
#define NN_XP_STRIDE (18 * 2 * BN_LIMBS)
#define NN_X1_OFF (NN_XP_OFF + BN_NONET_PER_WG * NN_XP_STRIDE)
#define NN_X1_STRIDE (9 * 2 * BN_LIMBS)
#define NN_X2_OFF (NN_X1_OFF + BN_NONET_PER_WG * NN_X1_STRIDE)
#define NN_X2_STRIDE (6 * 2 * BN_LIMBS)
#define NN_ZERO_OFF (NN_X2_OFF + BN_NONET_PER_WG * NN_X2_STRIDE)
#define NN_LDS_WORDS (NN_ZERO_OFF + 16)
static_assert(NN_LDS_WORDS * sizeof(int32_t) <= 160 * 1024, "nonet kernel: accumulators + exchange areas exceed the 160 KB of LDS of a gfx950 CU");

extern __shared__ int32_t nn_lds[];

#define NN_FENCE() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

struct NnLane {
  unsigned pair;        // lane pair within the verify, 0..8
  unsigned role;        // 0 real part, 1 imaginary part
  unsigned vslot;       // verify slot within the workgroup, 0..11 (lanes 54..63 of a wave share the last one and never publish)
  bool writer;          // lane belongs to a verify (may publish to the exchange areas)
  unsigned acc, bop;    // word offsets of this lane's accumulator / second-operand slot
};
__device__ __forceinline__ NnLane nn_lane() {
  NnLane L;
  const unsigned l = threadIdx.x & (BN_WAVE - 1), w = threadIdx.x / BN_WAVE;
  const unsigned v = l / BN_NONET_LANES;                       // 0..3
  L.writer = v < BN_NONET_PER_WAVE;
  L.pair = (l % BN_NONET_LANES) >> 1;
  L.role = l & 1u;
  L.vslot = w * BN_NONET_PER_WAVE + (L.writer ? v : BN_NONET_PER_WAVE - 1);
  L.acc = NN_ACC_OFF + threadIdx.x * NN_SLOT;
  L.bop = NN_BOP_OFF + threadIdx.x * NN_SLOT;
  return L;
}
__device__ __forceinline__ Fp2 nn_get(unsigned off) {
  Fp2 r;
#pragma unroll
  for (int i = 0; i < BN_LIMBS; ++i) r.c[0].v[i] = nn_lds[off + i];
  return r;
}
__device__ __forceinline__ void nn_put(unsigned off, const Fp2& x) {
#pragma unroll
  for (int i = 0; i < BN_LIMBS; ++i) nn_lds[off + i] = x.c[0].v[i];
}
// coefficient `idx` (0..5 in the memory order c0.c0 c0.c1 c0.c2 c1.c0 c1.c1 c1.c2; 6 = zero) of the Fq12 value in a lane's slot
__device__ __forceinline__ Fp2 nn_coef(unsigned slot_off, unsigned idx) { return nn_get(idx < 6 ? slot_off + idx * BN_LIMBS : (unsigned)NN_ZERO_OFF); }
__device__ __forceinline__ unsigned nn_xp(const NnLane& L, unsigned q) { return NN_XP_OFF + L.vslot * NN_XP_STRIDE + (q * 2 + L.role) * BN_LIMBS; }
__device__ __forceinline__ unsigned nn_x1(const NnLane& L, unsigned q) { return q < 9 ? NN_X1_OFF + L.vslot * NN_X1_STRIDE + (q * 2 + L.role) * BN_LIMBS : (unsigned)NN_ZERO_OFF; }
__device__ __forceinline__ unsigned nn_x2(const NnLane& L, unsigned q) { return NN_X2_OFF + L.vslot * NN_X2_STRIDE + (q * 2 + L.role) * BN_LIMBS; }
// -x in the lanes where m is all ones, x where m is zero
__device__ __forceinline__ Fp2 nn_cond_neg(const Fp2& x, int32_t m) {
  Fp2 r;
#pragma unroll
  for (int i = 0; i < BN_LIMBS; ++i) r.c[0].v[i] = (x.c[0].v[i] ^ m) - m;
  return r;
}
constexpr int nn_max3(int a, int b, int c) { return a > b ? (a > c ? a : c) : (b > c ? b : c); }

// ---- Granger-Scott squaring of the accumulator (bn254_field.h: fp12_cyclotomic_sqr_body<170>, fp4_sqr<S>) -------------------------------
// groups k = 0, 1, 2: (x, y) = (c0.c0, c1.c1), (c1.c0, c0.c2), (c0.c1, c1.c2); pair 3k + j squares x (j = 0), y (j = 1), carry(x + y) (j = 2)
__device__ __constant__ const unsigned char NN_SQ_I1[9] = {0, 4, 0, 3, 2, 3, 1, 5, 1};
__device__ __constant__ const unsigned char NN_SQ_I2[9] = {6, 6, 4, 6, 6, 2, 6, 6, 5};
// output coefficient w (= the pair that forms it, 0..5): its group, and whether it is the odd (2xy) half
__device__ __constant__ const unsigned char NN_SQ_OUT_K[6] = {0, 1, 2, 2, 0, 1};
__device__ __constant__ const unsigned char NN_SQ_OUT_ODD[6] = {0, 0, 0, 1, 1, 1};
__device__ __noinline__ void nn_csqr(const NnLane& L) {
  constexpr int m_sum = nn_max3(BN_SITE_MODE(170, 1), BN_SITE_MODE(173, 1), BN_SITE_MODE(176, 1));
  constexpr int m_odd = nn_max3(BN_SITE_MODE(171, 1), BN_SITE_MODE(174, 1), BN_SITE_MODE(177, 1));
  constexpr int m_even = nn_max3(BN_SITE_MODE(172, 1), BN_SITE_MODE(175, 1), BN_SITE_MODE(178, 1));
  constexpr int m_xi = BN_SITE_MODE(179, 1);
  {
    const Fp2 u = fp2_site(fp2_add(nn_coef(L.acc, NN_SQ_I1[L.pair]), nn_coef(L.acc, NN_SQ_I2[L.pair])), m_sum);
    const Fp2 sq = fp2_sqr(u);
    NN_FENCE();
    if (L.writer) nn_put(nn_xp(L, L.pair), sq);
    NN_FENCE();
  }
  const unsigned w = L.pair < 6 ? L.pair : L.pair - 6;       // pairs 6..8 repeat outputs 0..2 and publish nothing
  const unsigned k = NN_SQ_OUT_K[w];
  const bool odd = NN_SQ_OUT_ODD[w] != 0;
  const Fp2 x2 = nn_get(nn_xp(L, 3 * k)), y2 = nn_get(nn_xp(L, 3 * k + 1)), s2 = nn_get(nn_xp(L, 3 * k + 2));
  const Fp2 t_even = fp2_site(fp2_add(x2, fp2_mul_xi(y2)), m_even);                     // x^2 + xi y^2
  Fp2 t_odd = fp2_site(fp2_sub(fp2_sub(s2, x2), y2), m_odd);                            // 2 x y
  const Fp2 t_odd_xi = fp2_site(fp2_mul_xi(t_odd), m_xi);                               // c1.c0 = 3 xi t5 + 2 a
  t_odd = fp2_select(k == 2, t_odd_xi, t_odd);
  const Fp2 t = fp2_select(odd, t_odd, t_even);
  const Fp2 o = fp2_lin2_reduce(t, 3, nn_coef(L.acc, w), odd ? 2 : -2);
  NN_FENCE();
  if (L.writer && L.pair < 6) nn_put(nn_x2(L, w), o);
  NN_FENCE();
#pragma unroll
  for (unsigned e = 0; e < 6; ++e) nn_put(L.acc + e * BN_LIMBS, nn_get(nn_x2(L, e)));
  NN_FENCE();
}

// ---- acc <- acc * b, Karatsuba (bn254_field.h: fp12_mul_body, fp6_mul<S>) ----------------------------------------------------------------
// product q = 6 g + K: group g = 0: t0 = a.c0 b.c0, 1: t1 = a.c1 b.c1, 2: u = carry(a.c0 + a.c1) carry(b.c0 + b.c1); Karatsuba operand
// K = 0, 1, 2: coefficient K; 3: c1 + c2; 4: c0 + c1; 5: c0 + c2.  An operand is carry(C[i1] + C[i2]) + carry(C[i3] + C[i4]) with
// 6 = zero: for g < 2 the carries act on tight values (no-ops in value, uniform code), for g = 2 they are sites 28..33.
__device__ __constant__ const unsigned char NN_MUL_IDX[18][4] = {
    {0, 6, 6, 6}, {1, 6, 6, 6}, {2, 6, 6, 6}, {1, 6, 2, 6}, {0, 6, 1, 6}, {0, 6, 2, 6},
    {3, 6, 6, 6}, {4, 6, 6, 6}, {5, 6, 6, 6}, {4, 6, 5, 6}, {3, 6, 4, 6}, {3, 6, 5, 6},
    {0, 3, 6, 6}, {1, 4, 6, 6}, {2, 5, 6, 6}, {1, 4, 2, 5}, {0, 3, 1, 4}, {0, 3, 2, 5}};
__device__ __forceinline__ Fp2 nn_mul_operand(unsigned slot_off, unsigned q, int mode) {
  const Fp2 p = fp2_site(fp2_add(nn_coef(slot_off, NN_MUL_IDX[q][0]), nn_coef(slot_off, NN_MUL_IDX[q][1])), mode);
  const Fp2 r = fp2_site(fp2_add(nn_coef(slot_off, NN_MUL_IDX[q][2]), nn_coef(slot_off, NN_MUL_IDX[q][3])), mode);
  return fp2_add(p, r);
}
// Fq6 coefficient K of group g from its four products (fp6_kfin_coef): pk - pA - pB, xi on it for K = 0, + pD (xi on it for K = 1)
__device__ __constant__ const unsigned char NN_L1_A[3] = {1, 0, 0};
__device__ __constant__ const unsigned char NN_L1_B[3] = {2, 1, 2};
__device__ __constant__ const unsigned char NN_L1_D[3] = {0, 2, 1};
// output coefficient w: A + [xi] B + C with B, C negated for the c1 half (u - t0 - t1); 9 = zero
__device__ __constant__ const unsigned char NN_L2_A[6] = {0, 1, 2, 6, 7, 8};
__device__ __constant__ const unsigned char NN_L2_B[6] = {5, 3, 4, 0, 1, 2};
__device__ __constant__ const unsigned char NN_L2_C[6] = {9, 9, 9, 3, 4, 5};
__device__ __noinline__ void nn_mul(const NnLane& L) {              // the second operand sits in the lane's BOP slot
  constexpr int m_a = nn_max3(BN_SITE_MODE(28, 1), BN_SITE_MODE(29, 1), BN_SITE_MODE(30, 1));
  constexpr int m_b = nn_max3(BN_SITE_MODE(31, 1), BN_SITE_MODE(32, 1), BN_SITE_MODE(33, 1));
  constexpr int m_in = nn_max3(BN_SITE_MODE(20, 1), BN_SITE_MODE(24, 1), BN_SITE_MODE(34, 1));                 // fp6_mul<S>: NS(S, ...) under the xi
  constexpr int m_c0 = nn_max3(BN_SITE_MODE(21, 1), BN_SITE_MODE(25, 1), BN_SITE_MODE(35, 1));
  constexpr int m_c1 = nn_max3(BN_SITE_MODE(22, 1), BN_SITE_MODE(26, 1), BN_SITE_MODE(36, 1));
  constexpr int m_c2 = nn_max3(BN_SITE_MODE(23, 1), BN_SITE_MODE(27, 1), BN_SITE_MODE(37, 1));
  constexpr int m_k = nn_max3(m_c0, m_c1, m_c2);
  constexpr int m_out = nn_max3(nn_max3(BN_SITE_MODE(38, 2), BN_SITE_MODE(39, 2), BN_SITE_MODE(40, 2)), nn_max3(BN_SITE_MODE(41, 2), BN_SITE_MODE(42, 2), BN_SITE_MODE(43, 2)), 0);
  // two rounds of nine products
#pragma unroll 1
  for (unsigned r = 0; r < 2; ++r) {
    const unsigned q = L.pair + 9 * r;
    const Fp2 x = nn_mul_operand(L.acc, q, m_a), y = nn_mul_operand(L.bop, q, m_b);
    const Fp2 pr = fp2_mul(x, y);
    NN_FENCE();
    if (L.writer) nn_put(nn_xp(L, q), pr);
  }
  NN_FENCE();
  // level 1: pair p forms Fq6 coefficient K = p % 3 of group g = p / 3
  {
    const unsigned g = L.pair / 3, K = L.pair % 3, base = 6 * g;
    const Fp2 pk = nn_get(nn_xp(L, base + 3 + K)), pa = nn_get(nn_xp(L, base + NN_L1_A[K])), pb = nn_get(nn_xp(L, base + NN_L1_B[K]));
    const Fp2 pd = nn_get(nn_xp(L, base + NN_L1_D[K]));
    const Fp2 wv = fp2_sub(fp2_sub(pk, pa), pb);
    const Fp2 w_xi = fp2_mul_xi(fp2_site(wv, m_in));
    const Fp2 d_xi = fp2_mul_xi(pd);
    const Fp2 c = fp2_site(fp2_add(fp2_select(K == 0, w_xi, wv), fp2_select(K == 1, d_xi, pd)), m_k);
    NN_FENCE();
    if (L.writer) nn_put(nn_x1(L, L.pair), c);
    NN_FENCE();
  }
  // level 2: pair w < 6 forms output coefficient w:  c0.cK = t0.cK + (v t1).cK,  c1.cK = u.cK - t0.cK - t1.cK
  {
    const unsigned w = L.pair < 6 ? L.pair : L.pair - 6;
    const Fp2 a = nn_get(nn_x1(L, NN_L2_A[w])), b = nn_get(nn_x1(L, NN_L2_B[w])), c = nn_get(nn_x1(L, NN_L2_C[w]));
    const Fp2 b_xi = fp2_mul_xi(b);
    const int32_t neg = w >= 3 ? -1 : 0;
    const Fp2 bs = nn_cond_neg(fp2_select(w == 0, b_xi, b), neg), cs = nn_cond_neg(c, neg);
    const Fp2 o = fp2_site(fp2_add(fp2_add(a, bs), cs), m_out);
    NN_FENCE();
    if (L.writer && L.pair < 6) nn_put(nn_x2(L, w), o);
    NN_FENCE();
  }
#pragma unroll
  for (unsigned e = 0; e < 6; ++e) nn_put(L.acc + e * BN_LIMBS, nn_get(nn_x2(L, e)));
  NN_FENCE();
}

// the accumulator machine of bn254_pairing.h (fe_machine) with the two hot operations distributed over the nine pairs
__device__ __forceinline__ void nn_store12(unsigned slot_off, const Fp12& x) {
  const Fp2* c[6] = {&x.c0.c0, &x.c0.c1, &x.c0.c2, &x.c1.c0, &x.c1.c1, &x.c1.c2};
#pragma unroll
  for (int k = 0; k < 6; ++k) nn_put(slot_off + k * BN_LIMBS, *c[k]);
}
__device__ __forceinline__ void nn_load12(Fp12& x, unsigned slot_off) {
  Fp2* c[6] = {&x.c0.c0, &x.c0.c1, &x.c0.c2, &x.c1.c0, &x.c1.c1, &x.c1.c2};
#pragma unroll
  for (int k = 0; k < 6; ++k) *c[k] = nn_get(slot_off + k * BN_LIMBS);
}
__device__ __noinline__ void nn_machine(const NnLane& L, const unsigned char (*prog)[2]) {
  Fp12 slot[BN_FE_CHECK_SLOTS];
  Fp12& acc = *(Fp12*)(nn_lds + L.acc);            // the slot's first 54 words are an Fp12 in memory order
  BN_ASSUME_LDS(&acc);
#pragma clang loop unroll(disable)
  for (int pc = 0;; ++pc) {
    const int op = prog[pc][0], arg = prog[pc][1];
    if (op == FE_END) break;
    switch (op) {
      case FE_LOAD: acc = slot[arg]; break;
      case FE_STORE: slot[arg] = acc; break;
      case FE_CSQR: NN_FENCE(); nn_csqr(L); break;
      case FE_MUL: NN_FENCE(); nn_store12(L.bop, slot[arg]); NN_FENCE(); nn_mul(L); break;
      case FE_CONJ: fp6_neg(acc.c1, acc.c1); break;
      case FE_FROB: fp12_frob_body(acc, acc, arg); break;
      default: fp12_inv(acc, acc); break;
    }
  }
}

KERNEL_NONET void k_final_exp_nonet(size_t n, Ws ws, int use_hash, uint8_t* status_out) {
  const NnLane L = nn_lane();
  size_t i = (size_t)blockIdx.x * BN_NONET_PER_WG + L.vslot;
  const bool live = L.writer && i < n;
  if (i >= n) i = n - 1;                                 // lanes without a verify of their own follow along on the last one
  if (threadIdx.x < 16) nn_lds[NN_ZERO_OFF + threadIdx.x] = 0;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const Fp c = ws_load_fp(ws, PL_F0 + 2 * k + (int)L.role, i);
#pragma unroll
    for (int j = 0; j < BN_LIMBS; ++j) nn_lds[L.acc + k * BN_LIMBS + j] = c.v[j];
  }
  nn_lds[L.acc + 6 * BN_LIMBS] = 0;
  uint8_t st = ws_byte(ws, BY_ST_DECODE, i);
  if (st == ST_OK && use_hash) st = ws_byte(ws, BY_ST_HASH, i);
  __syncthreads();                                       // the zero block
  nn_machine(L, C_FE_CHECK);
  Fp12 f;
  nn_load12(f, L.acc);
  const bool one = fp12_is_one(f);                       // combined over the pair
  if (live && L.pair == 0 && L.role == 0) status_out[i] = st != ST_OK ? st : (one ? (uint8_t)ST_OK : (uint8_t)ST_VERIFICATION_FAILED);
}

bool bn254_nonet_fits_device() {
  int blocks = 0;
  hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, k_final_exp_nonet, BN_NONET_WG, NN_LDS_WORDS * sizeof(int32_t));
  if (e != hipSuccess) { (void)hipGetLastError(); return true; }
  return blocks > 0;
}
int bn254_nonet_final_exp(size_t n, Ws ws, int use_hash, uint8_t* status_out, hipStream_t s) {
  const unsigned grid = (unsigned)((n + BN_NONET_PER_WG - 1) / BN_NONET_PER_WG);
  k_final_exp_nonet<<<grid, BN_NONET_WG, NN_LDS_WORDS * sizeof(int32_t), s>>>(n, ws, use_hash, status_out);
  HIP_TRY(hipGetLastError());
  return 0;
}
