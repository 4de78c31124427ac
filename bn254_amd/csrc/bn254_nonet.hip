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

// Everything a lane needs to know about its place — which coefficients it squares, which products it forms, which it combines — as
// LDS word offsets worked out ONCE at kernel entry from the small index tables below and kept in registers (the routines are inlined
// into the interpreter loop): an indexed table read inside an operation would be a global-memory round trip on the latency path.
__device__ __constant__ const unsigned char NN_SQ_I1[9] = {0, 4, 0, 3, 2, 3, 1, 5, 1};        // pair 3k + j squares x (j = 0), y (j = 1), carry(x + y) (j = 2)
__device__ __constant__ const unsigned char NN_SQ_I2[9] = {6, 6, 4, 6, 6, 2, 6, 6, 5};        // of group k: (c0.c0, c1.c1), (c1.c0, c0.c2), (c0.c1, c1.c2); 6 = zero
__device__ __constant__ const unsigned char NN_SQ_OUT_K[6] = {0, 1, 2, 2, 0, 1};              // output coefficient w: its group ...
__device__ __constant__ const unsigned char NN_SQ_OUT_ODD[6] = {0, 0, 0, 1, 1, 1};            // ... and whether it is the odd (2xy) half
// product q = 6 g + K: group g = 0: t0 = a.c0 b.c0, 1: t1 = a.c1 b.c1, 2: u = carry(a.c0 + a.c1) carry(b.c0 + b.c1); Karatsuba operand
// K = 0, 1, 2: coefficient K; 3: c1 + c2; 4: c0 + c1; 5: c0 + c2.  An operand is carry(C[i1] + C[i2]) + carry(C[i3] + C[i4]) with
// 6 = zero: for g < 2 the carries act on tight values (no-ops in value, uniform code), for g = 2 they are sites 28..33.
__device__ __constant__ const unsigned char NN_MUL_IDX[18][4] = {
    {0, 6, 6, 6}, {1, 6, 6, 6}, {2, 6, 6, 6}, {1, 6, 2, 6}, {0, 6, 1, 6}, {0, 6, 2, 6},
    {3, 6, 6, 6}, {4, 6, 6, 6}, {5, 6, 6, 6}, {4, 6, 5, 6}, {3, 6, 4, 6}, {3, 6, 5, 6},
    {0, 3, 6, 6}, {1, 4, 6, 6}, {2, 5, 6, 6}, {1, 4, 2, 5}, {0, 3, 1, 4}, {0, 3, 2, 5}};
// Fq6 coefficient K of group g from its four products (fp6_kfin_coef): pk - pA - pB, xi on it for K = 0, + pD (xi on it for K = 1)
__device__ __constant__ const unsigned char NN_L1_A[3] = {1, 0, 0};
__device__ __constant__ const unsigned char NN_L1_B[3] = {2, 1, 2};
__device__ __constant__ const unsigned char NN_L1_D[3] = {0, 2, 1};
// output coefficient w: A + [xi] B + C with B, C negated for the c1 half (u - t0 - t1); 9 = zero
__device__ __constant__ const unsigned char NN_L2_A[6] = {0, 1, 2, 6, 7, 8};
__device__ __constant__ const unsigned char NN_L2_B[6] = {5, 3, 4, 0, 1, 2};
__device__ __constant__ const unsigned char NN_L2_C[6] = {9, 9, 9, 3, 4, 5};

struct NnLane {
  unsigned pair, role, vslot;   // lane pair within the verify (0..8), real / imaginary part, verify slot within the workgroup (0..11)
  bool writer;                  // the lane belongs to a verify (lanes 54..63 of a wave follow the wave's last verify and never publish)
  bool publishes_out;           // ... and its pair forms an output coefficient (pairs 0..5)
  unsigned acc, file;           // word offsets of the verify's accumulator / slot 0 of its slot file, for this lane's role
  // cyclotomic squaring
  unsigned sq_a, sq_b, sq_pub;  // operand = carry(LDS[sq_a] + LDS[sq_b]); its square goes to sq_pub
  unsigned so_x2, so_y2, so_s2, so_coef;
  bool so_odd, so_xi;
  int32_t so_sign;
  // multiplication: operand coefficient j of round r at acc + m_rel[r][j] / slot + m_rel[r][j], or the zero block
  unsigned m_rel[2][4];
  bool m_zero[2][4];
  unsigned m_pub[2];
  unsigned l1_k, l1_a, l1_b, l1_d, l1_pub;
  bool l1_xi_w, l1_xi_d;
  unsigned l2_a, l2_b, l2_c, out_coef;
  bool l2_xi;
  int32_t l2_neg;
};
__device__ __forceinline__ NnLane nn_lane() {
  NnLane L;
  const unsigned l = threadIdx.x & (BN_WAVE - 1), w = threadIdx.x / BN_WAVE;
  const unsigned v = l / BN_NONET_LANES;                       // 0..3
  L.writer = v < BN_NONET_PER_WAVE;
  L.pair = (l % BN_NONET_LANES) >> 1;
  L.role = l & 1u;
  L.vslot = w * BN_NONET_PER_WAVE + (L.writer ? v : BN_NONET_PER_WAVE - 1);
  L.publishes_out = L.writer && L.pair < 6;
  L.acc = NN_ACC_OFF + (L.vslot * 2 + L.role) * NN_SLOT;
  L.file = NN_FILE_OFF + (L.vslot * NN_NSLOTS * 2 + L.role) * NN_SLOT;          // slot k: + k * 2 * NN_SLOT
  const unsigned xp = NN_XP_OFF + L.vslot * NN_XP_STRIDE + L.role * BN_LIMBS, x1 = NN_X1_OFF + L.vslot * NN_X1_STRIDE + L.role * BN_LIMBS;
  auto XP = [&](unsigned q) { return xp + q * 2 * BN_LIMBS; };
  auto X1 = [&](unsigned q) { return q < 9 ? x1 + q * 2 * BN_LIMBS : (unsigned)NN_ZERO_OFF; };
  auto COEF = [&](unsigned idx) { return idx < 6 ? L.acc + idx * BN_LIMBS : (unsigned)NN_ZERO_OFF; };
  const unsigned p = L.pair, wq = p < 6 ? p : p - 6;           // pairs 6..8 repeat outputs 0..2 and publish nothing
  L.sq_a = COEF(NN_SQ_I1[p]); L.sq_b = COEF(NN_SQ_I2[p]); L.sq_pub = XP(p);
  const unsigned k = NN_SQ_OUT_K[wq];
  L.so_odd = NN_SQ_OUT_ODD[wq] != 0; L.so_xi = L.so_odd && k == 2; L.so_sign = L.so_odd ? 2 : -2;
  L.so_x2 = XP(3 * k); L.so_y2 = XP(3 * k + 1); L.so_s2 = XP(3 * k + 2); L.so_coef = L.acc + wq * BN_LIMBS;
  for (unsigned r = 0; r < 2; ++r) {
    const unsigned q = p + 9 * r;
    for (unsigned j = 0; j < 4; ++j) { const unsigned idx = NN_MUL_IDX[q][j]; L.m_zero[r][j] = idx >= 6; L.m_rel[r][j] = idx < 6 ? idx * BN_LIMBS : 0u; }
    L.m_pub[r] = XP(q);
  }
  const unsigned g = p / 3, K = p % 3;
  L.l1_k = XP(6 * g + 3 + K); L.l1_a = XP(6 * g + NN_L1_A[K]); L.l1_b = XP(6 * g + NN_L1_B[K]); L.l1_d = XP(6 * g + NN_L1_D[K]); L.l1_pub = X1(p);
  L.l1_xi_w = K == 0; L.l1_xi_d = K == 1;
  L.l2_a = X1(NN_L2_A[wq]); L.l2_b = X1(NN_L2_B[wq]); L.l2_c = X1(NN_L2_C[wq]); L.out_coef = L.acc + wq * BN_LIMBS;
  L.l2_xi = wq == 0; L.l2_neg = wq >= 3 ? -1 : 0;
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
// -x in the lanes where m is all ones, x where m is zero
__device__ __forceinline__ Fp2 nn_cond_neg(const Fp2& x, int32_t m) {
  Fp2 r;
#pragma unroll
  for (int i = 0; i < BN_LIMBS; ++i) r.c[0].v[i] = (x.c[0].v[i] ^ m) - m;
  return r;
}
constexpr int nn_max3(int a, int b, int c) { return a > b ? (a > c ? a : c) : (b > c ? b : c); }
// ---- Granger-Scott squaring of the accumulator (bn254_field.h: fp12_cyclotomic_sqr_body<170>, fp4_sqr<S>) -------------------------------
__device__ __forceinline__ void nn_csqr(const NnLane& L) {
  constexpr int m_sum = nn_max3(BN_SITE_MODE(170, 1), BN_SITE_MODE(173, 1), BN_SITE_MODE(176, 1));
  constexpr int m_odd = nn_max3(BN_SITE_MODE(171, 1), BN_SITE_MODE(174, 1), BN_SITE_MODE(177, 1));
  constexpr int m_even = nn_max3(BN_SITE_MODE(172, 1), BN_SITE_MODE(175, 1), BN_SITE_MODE(178, 1));
  constexpr int m_xi = BN_SITE_MODE(179, 1);
  {
    const Fp2 sq = fp2_sqr(fp2_site(fp2_add(nn_get(L.sq_a), nn_get(L.sq_b)), m_sum));
    NN_FENCE();
    if (L.writer) nn_put(L.sq_pub, sq);
    NN_FENCE();
  }
  const Fp2 x2 = nn_get(L.so_x2), y2 = nn_get(L.so_y2), s2 = nn_get(L.so_s2);
  // even half: x^2 + xi y^2;  odd half: 2xy = s^2 - x^2 - y^2, for c1.c0 times xi — ONE multiplication by xi serves both: of y^2 in the
  // lanes that form an even output, of 2xy in the others (used by the c1.c0 lanes only)
  const Fp2 t_odd = fp2_site(fp2_sub(fp2_sub(s2, x2), y2), m_odd);
  const Fp2 xi_part = fp2_mul_xi(fp2_select(L.so_odd, t_odd, y2));
  const Fp2 t_even = fp2_site(fp2_add(x2, xi_part), m_even);
  const Fp2 t_odd_sel = fp2_select(L.so_xi, fp2_site(xi_part, m_xi), t_odd);
  const Fp2 o = fp2_lin2_reduce(fp2_select(L.so_odd, t_odd_sel, t_even), 3, nn_get(L.so_coef), L.so_sign);
  NN_FENCE();                                     // every pair has read the old coefficients it needs (one wave: in order)
  if (L.publishes_out) nn_put(L.so_coef, o);      // straight into the verify's accumulator
  NN_FENCE();
}

// ---- acc <- acc * slot, Karatsuba (bn254_field.h: fp12_mul_body, fp6_mul<S>) ------------------------------------------------------------
__device__ __forceinline__ Fp2 nn_mul_operand(const NnLane& L, unsigned base, unsigned r, int mode) {
  Fp2 c[4];
#pragma unroll
  for (unsigned j = 0; j < 4; ++j) c[j] = nn_get(L.m_zero[r][j] ? (unsigned)NN_ZERO_OFF : base + L.m_rel[r][j]);
  return fp2_add(fp2_site(fp2_add(c[0], c[1]), mode), fp2_site(fp2_add(c[2], c[3]), mode));
}
__device__ __forceinline__ void nn_mul(const NnLane& L, unsigned bslot) {     // bslot: word offset of the second operand (a slot of the file), this lane's role
  constexpr int m_a = nn_max3(BN_SITE_MODE(28, 1), BN_SITE_MODE(29, 1), BN_SITE_MODE(30, 1));
  constexpr int m_b = nn_max3(BN_SITE_MODE(31, 1), BN_SITE_MODE(32, 1), BN_SITE_MODE(33, 1));
  constexpr int m_in = nn_max3(BN_SITE_MODE(20, 1), BN_SITE_MODE(24, 1), BN_SITE_MODE(34, 1));                 // fp6_mul<S>: NS(S, ...) under the xi
  constexpr int m_k = nn_max3(nn_max3(BN_SITE_MODE(21, 1), BN_SITE_MODE(25, 1), BN_SITE_MODE(35, 1)), nn_max3(BN_SITE_MODE(22, 1), BN_SITE_MODE(26, 1), BN_SITE_MODE(36, 1)),
                              nn_max3(BN_SITE_MODE(23, 1), BN_SITE_MODE(27, 1), BN_SITE_MODE(37, 1)));
  constexpr int m_out = nn_max3(nn_max3(BN_SITE_MODE(38, 2), BN_SITE_MODE(39, 2), BN_SITE_MODE(40, 2)), nn_max3(BN_SITE_MODE(41, 2), BN_SITE_MODE(42, 2), BN_SITE_MODE(43, 2)), 0);
  // two rounds of nine products
#pragma unroll
  for (unsigned r = 0; r < 2; ++r) {
    const Fp2 x = nn_mul_operand(L, L.acc, r, m_a), y = nn_mul_operand(L, bslot, r, m_b);
    const Fp2 pr = fp2_mul(x, y);
    NN_FENCE();
    if (L.writer) nn_put(L.m_pub[r], pr);
  }
  NN_FENCE();
  {  // level 1: pair p forms Fq6 coefficient K = p % 3 of group g = p / 3
    const Fp2 pk = nn_get(L.l1_k), pa = nn_get(L.l1_a), pb = nn_get(L.l1_b), pd = nn_get(L.l1_d);
    const Fp2 wv = fp2_sub(fp2_sub(pk, pa), pb);
    // K = 0: xi (pk - p1 - p2) + p0;  K = 1: (pk - p0 - p1) + xi p2;  K = 2: (pk - p0 - p2) + p1 — one multiplication by xi, of the bracket or of pD
    const Fp2 xi_part = fp2_mul_xi(fp2_select(L.l1_xi_w, fp2_site(wv, m_in), pd));
    const Fp2 c = fp2_site(fp2_add(fp2_select(L.l1_xi_w, xi_part, wv), fp2_select(L.l1_xi_w, pd, fp2_select(L.l1_xi_d, xi_part, pd))), m_k);
    NN_FENCE();
    if (L.writer) nn_put(L.l1_pub, c);
    NN_FENCE();
  }
  {  // level 2: pair w < 6 forms output coefficient w:  c0.cK = t0.cK + (v t1).cK,  c1.cK = u.cK - t0.cK - t1.cK
    const Fp2 a = nn_get(L.l2_a), b = nn_get(L.l2_b), c = nn_get(L.l2_c);
    const Fp2 bs = nn_cond_neg(fp2_select(L.l2_xi, fp2_mul_xi(b), b), L.l2_neg), cs = nn_cond_neg(c, L.l2_neg);
    const Fp2 o = fp2_site(fp2_add(fp2_add(a, bs), cs), m_out);
    NN_FENCE();
    if (L.publishes_out) nn_put(L.out_coef, o);   // straight into the verify's accumulator (all operand reads are behind us)
    NN_FENCE();
  }
}

// the accumulator machine of bn254_pairing.h (fe_machine) with the two hot operations distributed over the nine pairs; accumulator and
// slot file are the verify's shared copies in LDS
__device__ __forceinline__ void nn_machine(const NnLane& L, const unsigned char (*prog)[2]) {
  Fp12& acc = *(Fp12*)(nn_lds + L.acc);            // the first 54 words of a slot are an Fp12 in memory order
  BN_ASSUME_LDS(&acc);
#pragma clang loop unroll(disable)
  for (int pc = 0;; ++pc) {
    const int op = prog[pc][0], arg = prog[pc][1];
    if (op == FE_END) break;
    const unsigned sl = L.file + (unsigned)arg * 2 * NN_SLOT;
    Fp12& slot = *(Fp12*)(nn_lds + sl);
    BN_ASSUME_LDS(&slot);
    NN_FENCE();
    switch (op) {
      case FE_LOAD: acc = slot; break;                       // identical words from every pair
      case FE_STORE: slot = acc; break;
      case FE_CSQR: nn_csqr(L); break;
      case FE_MUL: nn_mul(L, sl); break;
      case FE_CONJ: fp6_neg(acc.c1, acc.c1); break;
      case FE_FROB: fp12_frob_body(acc, acc, arg); break;
      default: fp12_inv(acc, acc); break;
    }
  }
  NN_FENCE();
}

KERNEL_NONET void k_final_exp_nonet(size_t n, Ws ws, int use_hash, uint8_t* status_out) {
  const NnLane L = nn_lane();
  size_t i = (size_t)blockIdx.x * BN_NONET_PER_WG + L.vslot;
  const bool live = L.writer && i < n;
  if (i >= n) i = n - 1;                                 // lanes without a verify of their own follow along on the last one
  if (threadIdx.x < 16) nn_lds[NN_ZERO_OFF + threadIdx.x] = 0;
  if (L.publishes_out) {                                 // pair k < 6 brings in coefficient k of the Miller value
    const Fp c = ws_load_fp(ws, PL_F0 + 2 * (int)L.pair + (int)L.role, i);
#pragma unroll
    for (int j = 0; j < BN_LIMBS; ++j) nn_lds[L.acc + L.pair * BN_LIMBS + j] = c.v[j];
  }
  uint8_t st = ws_byte(ws, BY_ST_DECODE, i);
  if (st == ST_OK && use_hash) st = ws_byte(ws, BY_ST_HASH, i);
  __syncthreads();                                       // the zero block, the accumulators
  nn_machine(L, C_FE_CHECK);
  Fp12 f;
  {
    Fp2* c[6] = {&f.c0.c0, &f.c0.c1, &f.c0.c2, &f.c1.c0, &f.c1.c1, &f.c1.c2};
#pragma unroll
    for (int k = 0; k < 6; ++k) *c[k] = nn_get(L.acc + k * BN_LIMBS);
  }
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
