// Second device translation unit of libbn254hip.so: the Miller loop and the final exponentiation of a verify
// carried by a LANE PAIR (bn254_fp2_pair.h): lane 2i holds the real parts and lane 2i+1 the imaginary parts of
// every Fq2 value of item i.  Same tower / pairing source as bn254_hip.hip (bn254_field.h, bn254_pairing.h),
// compiled against the pair implementation of the fp2_* interface.  Per lane: half the multiplications, half the
// registers (<= 256, two waves per SIMD), half the LDS (an accumulator slot is 55 words).
// Replaces, for ECDSA::verify (/root/reference/src/ecdsa.rs:49-64), k_miller_verify + k_final_exp.
#include <hip/hip_runtime.h>

#define BN_SPLIT_FP2 1
#if defined(BN_PAIR_FP6_LAZY) && !defined(BN_FP6_LAZY)
#define BN_FP6_LAZY 1              // Fq6-level lazy reduction in fp12_sqr / fp12_mul_line2 of this translation unit (bn254_field.h: fp6_mul_lazy)
#endif
#ifndef BN_PAIR_NO_SQR_DPP_ASM
#define BN_PAIR_SQR_DPP_ASM 1      // role prologue of the Fq2 squaring with folded DPP operands (bn254_fp2_pair.h)
#endif
#ifndef BN_PAIR_CALL_FP12_HOT
#define BN_INLINE_FP12_HOT 1       // fp12_sqr / fp12_mul_line2 inlined into the Miller loops (bn254_field.h: BN_DEVH)
#endif
#ifndef BN_PAIR_CALL_MUL_LINE
#define BN_INLINE_MUL_LINE 1       // fp12_mul_line inlined into the single-pair Miller loops too (pairing workload 7.30 -> 7.52 M pairings/s, same box)
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
#define bn254 bn254_pair   // own namespace: the Fq2 / Fq12 types differ from the other translation unit
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

// Decode n uncompressed G2 points (/root/reference/src/utils.rs:107-116) on lane pairs: each lane reads, range-checks
// and converts the two 32-byte words of its role; curve equation and (flag bit0) the subgroup test — one 63-bit
// ladder on the twist — run in the pair layout.  Same statuses and workspace outputs as k_decode_g2.
__device__ __forceinline__ bool load_fp_be_checked(Fp& r, const uint8_t* p, uint32_t& any) {
  const uint32_t* w = (const uint32_t*)p;
  U256 t;
#pragma unroll
  for (int k = 0; k < 8; ++k) { t.w[7 - k] = __builtin_bswap32(w[k]); any |= w[k]; }
  bool ok = !u256_geq(t.w, C_Q);
  r = fp_from_u256(t);
  return ok;
}
KERNEL_PAIR void k_decode_g2_pair(const uint8_t* pts, size_t n, uint32_t flags, Ws ws, int accumulate) {
  const unsigned role = threadIdx.x & 1u;
  size_t i = ((size_t)blockIdx.x * BN_PAIR_WG + threadIdx.x) >> 1;
  const bool live = i < n;                          // no early return: jac_accumulate votes across the wave
  const size_t ii = live ? i : n - 1;
  const uint8_t* b = pts + 128 * ii;
  G2Affine q;
  uint32_t any = 0;
  bool ok = load_fp_be_checked(q.x.c[0], b + 32 * role, any);
  ok = load_fp_be_checked(q.y.c[0], b + 64 + 32 * role, any) && ok;
  any |= (uint32_t)bn_partner_word((int32_t)any);
  ok = bn_pair_and(ok);
  q.inf = any == 0;
  uint8_t st = ST_OK;
  if (q.inf) st = (flags & FLAG_REJECT_IDENTITY) ? (uint8_t)ST_INVALID_GROUP_POINT : (uint8_t)ST_OK;
  else if (!ok) st = ST_NOT_MEMBER;
  const bool on_curve = g2_on_curve(q);             // pair-combined; evaluated by every lane
  if (st == ST_OK && !q.inf && !on_curve) st = ST_INVALID_GROUP_POINT;
  G2Affine gen;
  gen.x = fp2_load_const(C_G2_GEN[0]); gen.y = fp2_load_const(C_G2_GEN[1]); gen.inf = false;
  if (st != ST_OK) q = gen;                          // failed lanes walk on with the generator (pairs decide together)
  if (flags & FLAG_G2_SUBGROUP_CHECK) {              // wave-uniform
    __shared__ G2Jac lds_up[BN_PAIR_WG];             // the ladder's accumulator, 27 words per lane
#if defined(BN_SUBGROUP_PRIVATE)
    bool in = g2_in_subgroup(q);
#else
    bool in = g2_in_subgroup_lds(q, lds_up[threadIdx.x]);
#endif
    if (st == ST_OK && !in) { st = ST_INVALID_GROUP_POINT; q = gen; }
  }
  if (!live) return;
  ws_store_fp(ws, PL_QX0 + (int)role, i, q.x.c[0]);
  ws_store_fp(ws, PL_QY0 + (int)role, i, q.y.c[0]);
  if (role == 0) {
    ws_byte(ws, BY_Q_INF, i) = q.inf;
    uint8_t prev = accumulate ? ws_byte(ws, BY_ST_DECODE, i) : (uint8_t)ST_OK;
    ws_byte(ws, BY_ST_DECODE, i) = prev != ST_OK ? prev : st;
  }
}
// Compressed public keys (65 B, bn::G2::from_compressed: Fq2 square root + subgroup test) into the Q planes of a
// verify, on lane pairs; same statuses as k_decompress_g2_ws.
KERNEL_PAIR void k_decompress_g2_pair(const uint8_t* in, size_t n, Ws ws, int skip_subgroup_test) {     // skip_subgroup_test: the caller runs it (bn254_lm_g2_subgroup)
  const unsigned role = threadIdx.x & 1u;
  size_t i = ((size_t)blockIdx.x * BN_PAIR_WG + threadIdx.x) >> 1;
  const bool live = i < n;                          // no early return: the subgroup ladder votes across the wave
  const size_t ii = live ? i : n - 1;
  G2Affine q;
  uint8_t st = decompress_g2(q, in + 65 * ii);
  G2Affine gen;
  gen.x = fp2_load_const(C_G2_GEN[0]); gen.y = fp2_load_const(C_G2_GEN[1]); gen.inf = false;
  if (st != ST_OK) q = gen;
  __shared__ G2Jac lds_up[BN_PAIR_WG];               // the subgroup ladder's accumulator, 27 words per lane
  bool in_sub = true;
  if (!skip_subgroup_test) {                         // wave-uniform
#if defined(BN_SUBGROUP_PRIVATE)
  in_sub = g2_in_subgroup(q);
#else
  in_sub = g2_in_subgroup_lds(q, lds_up[threadIdx.x]);
#endif
  }
  if (st == ST_OK && !in_sub) { st = ST_NOT_MEMBER; q = gen; }
  if (!live) return;
  ws_store_fp(ws, PL_QX0 + (int)role, i, q.x.c[0]);
  ws_store_fp(ws, PL_QY0 + (int)role, i, q.y.c[0]);
  if (role == 0) {
    ws_byte(ws, BY_Q_INF, i) = q.inf;
    uint8_t prev = ws_byte(ws, BY_ST_DECODE, i);
    ws_byte(ws, BY_ST_DECODE, i) = prev != ST_OK ? prev : st;
  }
}
int bn254_pair_decompress_g2(const uint8_t* in, size_t n, Ws ws, hipStream_t s, int skip_subgroup_test) {
  k_decompress_g2_pair<<<(unsigned)((2 * n + BN_PAIR_WG - 1) / BN_PAIR_WG), BN_PAIR_WG, 0, s>>>(in, n, ws, skip_subgroup_test);
  HIP_TRY(hipGetLastError());
  return 0;
}
int bn254_pair_decode_g2(const uint8_t* pts, size_t n, uint32_t flags, Ws ws, int accumulate, hipStream_t s) {
  k_decode_g2_pair<<<(unsigned)((2 * n + BN_PAIR_WG - 1) / BN_PAIR_WG), BN_PAIR_WG, 0, s>>>(pts, n, flags, ws, accumulate);
  HIP_TRY(hipGetLastError());
  return 0;
}

// f = miller(H(m), pk) * miller(sig, -G2); item = lane >> 1.  Both lanes of a pair take every branch together
// (item-level conditions only), so the DPP exchanges always find their partner active.
// mode 1 = check_public_keys (/root/reference/src/ecdsa.rs:80-86): miller(G1::one(), pk_g2) * miller(pk_g1, -G2) with
// pk_g1 in the P1 planes
KERNEL_PAIR void k_miller_verify_pair(size_t n, Ws ws, const uint32_t* map, const uint32_t* count, int mode) {
  size_t i = ((size_t)blockIdx.x * BN_PAIR_WG + threadIdx.x) >> 1;
  if (i >= n) return;
  if (map) { if (i >= *count) return; i = map[i]; }
  G1Affine sig, h;
  G2Affine pk;
  ws_load_g1(ws, PL_P1X, BY_P1_INF, i, sig);
  if (mode == 1) { h.x = fp_load_const(C_G1_GEN[0]); h.y = fp_load_const(C_G1_GEN[1]); h.inf = false; }   // wave-uniform
  else ws_load_g1(ws, PL_P2X, BY_P2_INF, i, h);
  pk.x = ws_load_fp2_own(ws, PL_QX0, i);
  pk.y = ws_load_fp2_own(ws, PL_QY0, i);
  pk.inf = ws_byte(ws, BY_Q_INF, i) != 0;
  __shared__ Fp12PairSlot lds_f[BN_PAIR_WG];
  Fp12& f = lds_f[threadIdx.x].v;
  BN_CLK_BEGIN(ws);
  miller_loop<true, true, true>(f, h, pk, sig);
  BN_CLK_END(ws, 0);
  ws_store_f12_own(ws, i, f);
}
// Keyed verify: f = miller(H(m), pk[key_idx[i]]) * miller(sig, -G2) with BOTH line sequences read from tables — pair A from
// the registered key's 87 lines in HBM (each lane loads the 2 x 9 words of its role per line: 12.5 KB per verify, L2-resident
// for a validator set), pair B from the constant table.  No twist-point arithmetic: 36 product slots per doubling step
// against 48 (bn254_pairing.h: miller_loop_keyed).  Status: the signature's decode status stays first; then the key's
// (index >= n_keys -> IndexOutOfBounds, else what registration found), written back for k_final_exp_pair.
// `base`: tuple i lives at workspace index base + i (the per-group tuples of the keyed randomised verify sit behind the items);
// `count` / `map`: a device-side tuple count and index map (exact re-check of the items of failed groups); key_idx is indexed
// by the tuple's own number in every case.
KERNEL_PAIR void k_miller_verify_keyed_pair(size_t n, Ws ws, const uint32_t* key_idx, KeyTable kt, size_t base, const uint32_t* map,
                                            const uint32_t* count) {
  size_t i = ((size_t)blockIdx.x * BN_PAIR_WG + threadIdx.x) >> 1;
  if (i >= n) return;
  if (count && i >= *count) return;
  if (map) i = map[i];
  uint32_t key = key_idx[i];
  i += base;
  uint8_t kst = ST_OK;
  if (key >= kt.n_keys) { kst = ST_INDEX_OOB; key = 0; }
  else kst = kt.st[key];
  const bool key_inf = kst != ST_OK || kt.inf[key] != 0;      // a refused key walks the loop as a skipped pair
  if ((threadIdx.x & 1u) == 0) {
    const uint8_t prev = ws_byte(ws, BY_ST_DECODE, i);
    ws_byte(ws, BY_ST_DECODE, i) = prev != ST_OK ? prev : kst;
  }
  G1Affine sig, h;
  ws_load_g1(ws, PL_P1X, BY_P1_INF, i, sig);
  ws_load_g1(ws, PL_P2X, BY_P2_INF, i, h);
  __shared__ Fp12PairSlot lds_f[BN_PAIR_WG];
  Fp12& f = lds_f[threadIdx.x].v;
  typedef const int32_t (*LinePtr)[2][2][BN_LIMBS];
  BN_CLK_BEGIN(ws);
  miller_loop_keyed<true>(f, h, key_inf, (LinePtr)(kt.lines + (size_t)key * BN_N_FIXED_LINES * BN_KEY_LINE_WORDS), sig);
  BN_CLK_END(ws, 0);
  ws_store_f12_own(ws, i, f);
}
int bn254_pair_miller_verify_keyed(size_t n, Ws ws, const uint32_t* key_idx, KeyTable kt, hipStream_t s, size_t base, const uint32_t* map,
                                   const uint32_t* count) {
  k_miller_verify_keyed_pair<<<(unsigned)((2 * n + BN_PAIR_WG - 1) / BN_PAIR_WG), BN_PAIR_WG, 0, s>>>(n, ws, key_idx, kt, base, map, count);
  HIP_TRY(hipGetLastError());
  return 0;
}
// generic single pair per lane pair: f = miller(P1, Q)   (bn254_batch_pairing*)
KERNEL_PAIR void k_miller_var_pair(size_t n, Ws ws) {
  size_t i = ((size_t)blockIdx.x * BN_PAIR_WG + threadIdx.x) >> 1;
  if (i >= n) return;
  G1Affine p;
  G2Affine q;
  ws_load_g1(ws, PL_P1X, BY_P1_INF, i, p);
  q.x = ws_load_fp2_own(ws, PL_QX0, i);
  q.y = ws_load_fp2_own(ws, PL_QY0, i);
  q.inf = ws_byte(ws, BY_Q_INF, i) != 0;
  __shared__ Fp12PairSlot lds_f[BN_PAIR_WG];
  Fp12& f = lds_f[threadIdx.x].v;
  miller_loop<true, false, true>(f, p, q, p);
  ws_store_f12_own(ws, i, f);
}
// ---- randomised batch verification (see bn254_hip.hip) on lane pairs -------------------------------------------
// A 256-lane workgroup = 128 lane pairs.  k_miller_rand_pair: one item per pair -> 2 groups of 64 items per
// workgroup; k_miller_rand2_pair: two items per pair (shared f^2, merged lines) -> 4 groups per workgroup.
// The Miller values of a group are multiplied by an LDS tree; both lanes of a pair take every branch together.
template <int ITEMS_PER_PAIR>
__device__ __forceinline__ void miller_rand_pair_body(size_t n, size_t n_groups, Ws ws, size_t gbase) {
  constexpr unsigned PAIRS_PER_GROUP = BN_WAVE / ITEMS_PER_PAIR;                 // 64 or 32
  constexpr unsigned GROUPS_PER_WG = (BN_PAIR_WG / 2) / PAIRS_PER_GROUP;           // 2 or 4
  const unsigned pair = threadIdx.x >> 1, gp = pair / PAIRS_PER_GROUP, tp = pair % PAIRS_PER_GROUP;
  const size_t group = (size_t)blockIdx.x * GROUPS_PER_WG + gp;
  const size_t i0 = group * BN_WAVE + (size_t)tp * ITEMS_PER_PAIR;
  __shared__ Fp12PairSlot lds_f[BN_PAIR_WG];
  Fp12& f = lds_f[threadIdx.x].v;
  G1Affine a0;
  G2Affine pk0;
  const size_t j0 = i0 < n ? i0 : n - 1;
  ws_load_g1(ws, PL_HASHX, BY_A_INF, j0, a0);
  if (i0 >= n) a0.inf = true;
  pk0.x = ws_load_fp2_own(ws, PL_QX0, j0); pk0.y = ws_load_fp2_own(ws, PL_QY0, j0); pk0.inf = ws_byte(ws, BY_Q_INF, j0) != 0;
  if constexpr (ITEMS_PER_PAIR == 2) {
    G1Affine a1;
    G2Affine pk1;
    const size_t i1 = i0 + 1, j1 = i1 < n ? i1 : n - 1;
    ws_load_g1(ws, PL_HASHX, BY_A_INF, j1, a1);
    if (i1 >= n) a1.inf = true;
    pk1.x = ws_load_fp2_own(ws, PL_QX0, j1); pk1.y = ws_load_fp2_own(ws, PL_QY0, j1); pk1.inf = ws_byte(ws, BY_Q_INF, j1) != 0;
    miller_loop_2var<true>(f, a0, pk0, a1, pk1);
  } else {
    miller_loop<true, false, true>(f, a0, pk0, a0);
  }
  __syncthreads();
  for (unsigned stride = PAIRS_PER_GROUP / 2; stride >= 1; stride >>= 1) {
    if (tp < stride) fp12_mul(f, f, lds_f[threadIdx.x + 2 * stride].v);
    __syncthreads();
  }
  if (tp == 0 && group < n_groups) ws_store_f12_own(ws, gbase + group, f);
}
KERNEL_PAIR void k_miller_rand_pair(size_t n, size_t n_groups, Ws ws, size_t gbase) { miller_rand_pair_body<1>(n, n_groups, ws, gbase); }
KERNEL_PAIR void k_miller_rand2_pair(size_t n, size_t n_groups, Ws ws, size_t gbase) { miller_rand_pair_body<2>(n, n_groups, ws, gbase); }
// per group: F_g * miller(S_g, -G2) through the line table
KERNEL_PAIR void k_rand_tail_pair(size_t n_groups, Ws ws, size_t gbase) {
  size_t g = ((size_t)blockIdx.x * BN_PAIR_WG + threadIdx.x) >> 1;
  if (g >= n_groups) return;
  G1Affine s, unused_g1;
  G2Affine unused_g2;
  ws_load_g1(ws, PL_P1X, BY_P1_INF, gbase + g, s);
  unused_g1.x = fp_load_const(C_G1_GEN[0]); unused_g1.y = fp_load_const(C_G1_GEN[1]); unused_g1.inf = false;
  unused_g2.x = fp2_load_const(C_G2_GEN[0]); unused_g2.y = fp2_load_const(C_G2_GEN[1]); unused_g2.inf = false;
  Fp12 fg;
  ws_load_f12_own(ws, gbase + g, fg);
  __shared__ Fp12PairSlot lds_f[BN_PAIR_WG];
  Fp12& f = lds_f[threadIdx.x].v;
  miller_loop<false, true, true>(f, unused_g1, unused_g2, s);
  fp12_mul(f, f, fg);
  ws_store_f12_own(ws, gbase + g, f);
}
int bn254_pair_miller_rand(size_t n, size_t n_groups, int items_per_pair, Ws ws, size_t gbase, hipStream_t s) {
  if (items_per_pair == 2) k_miller_rand2_pair<<<(unsigned)((n_groups + 3) / 4), BN_PAIR_WG, 0, s>>>(n, n_groups, ws, gbase);
  else k_miller_rand_pair<<<(unsigned)((n_groups + 1) / 2), BN_PAIR_WG, 0, s>>>(n, n_groups, ws, gbase);
  HIP_TRY(hipGetLastError());
  return 0;
}
int bn254_pair_rand_tail(size_t n_groups, Ws ws, size_t gbase, hipStream_t s) {
  k_rand_tail_pair<<<(unsigned)((2 * n_groups + BN_PAIR_WG - 1) / BN_PAIR_WG), BN_PAIR_WG, 0, s>>>(n_groups, ws, gbase);
  HIP_TRY(hipGetLastError());
  return 0;
}

// Aggregation for config 3 on lane pairs (see k_aggregate in bn254_hip.hip): a pair walks the signer list of its
// tuple two entries per iteration; each lane adds the signature of "its" entry (2t + role) to its own partial G1 sum and the
// two partial sums are added at the end.  The public keys take one of two routes, chosen per wave:
//   * direct: both keys of the iteration are added to the G2 sum in the pair layout (lists shorter than n_groups);
//   * subset sums (k_pool_subsets_g2): the walk only sets bit (signer mod 8) in the tuple's mask byte of group signer / 8 (LDS,
//     atomic OR: the two lanes of a pair may meet in one word), then ONE table entry per group is added — n_signers / 8
//     additions whatever the list length.  A tuple that names a signer twice (the OR would swallow the second copy) is
//     detected by the bit already being set and takes the direct route afterwards; sums are commutative, so order is free.
//     With the per-message signature tables (k_pool_subsets_g1, groups4 != 0) the walk adds no signature either: the same
//     mask bytes, read as two nibbles, select one table entry per group of 4 signers — the real-part lane takes the low
//     nibbles, the imaginary-part lane the high ones, one G1 and one G2 addition per lane and group of 8.
// a table record as the source of an in-place addition (bn254_curve.h: jac_madd_inplace_from): the lane's half of a G2 record or a G1
// record (bn254_ws.h: Pool — x at word 0, y at word 9), loaded inside the addition; pointer and flag travel in registers
struct PoolRec {
  const int32_t* p;
  bool inf;
  __device__ __forceinline__ void operator()(G1Affine& q) const {
#pragma unroll
    for (int k = 0; k < BN_LIMBS; ++k) { q.x.v[k] = p[k]; q.y.v[k] = p[BN_LIMBS + k]; }
    q.inf = inf;
  }
  __device__ __forceinline__ void operator()(G2Affine& q) const {
#pragma unroll
    for (int k = 0; k < BN_LIMBS; ++k) { q.x.c[0].v[k] = p[k]; q.y.c[0].v[k] = p[BN_LIMBS + k]; }
    q.inf = inf;
  }
};
extern __shared__ uint32_t bn_agg_masks[];     // [BN_PAIR_WG / 2 tuples][mask_stride words], mask_stride odd
KERNEL_PAIR void k_aggregate_pair(const uint32_t* tuple_msg, const uint64_t* tuple_off, const uint32_t* signer_idx, size_t n, size_t n_signers,
                                  size_t n_msgs, Pool pk_pool, Pool sig_pool, Pool h_pool, Pool sub_pool, unsigned n_groups, Pool sub1_pool, unsigned groups4,
                                  unsigned mask_stride, Ws ws, const uint32_t* perm, Pool wide2_pool, int wide2, Pool wide1_pool, int wide1) {
  const unsigned role = threadIdx.x & 1u;
  // `perm` (tuples bucketed by message, bn254_hip.hip: k_agg_sort_*): slot -> tuple.  Workgroups are dispatched to the 8 XCDs round
  // robin and every XCD has its own L2: with the map, XCD x takes a CONTIGUOUS eighth of the slots, so that the ~8 workgroups that
  // share a message's table share an L2 (64 resident workgroups per XCD = ~8 tables of 0.3 MB in 4 MB).
  size_t blk = blockIdx.x;
  if (perm) {
    const size_t nb = gridDim.x, per = (nb + 7) / 8, x = blk & 7u, k = blk >> 3;
    blk = x * per + k;                     // a bijection onto [0, 8 * per) >= nb: slots past the end are idle
  }
  size_t i = (blk * BN_PAIR_WG + threadIdx.x) >> 1;
  const bool live = i < n;                 // no early return: the wave-level votes and shuffles below need every lane
  size_t ii = live ? i : n - 1;
  if (perm) { ii = perm[ii]; i = ii; }     // every read and write below goes to the tuple's own index
  uint32_t m = tuple_msg[ii];
  uint64_t lo = tuple_off[ii], hi = live ? tuple_off[ii + 1] : lo;
  // the two running sums live in LDS (27 words each, odd stride): the additions are real functions that take them by reference
  __shared__ G1Jac lds_acc1[BN_PAIR_WG];
  __shared__ G2Jac lds_acc2[BN_PAIR_WG];
  static_assert(sizeof(G1Jac) == 3 * BN_LIMBS * 4 && sizeof(G2Jac) == 3 * BN_LIMBS * 4, "accumulators: 27 words per lane in the pair layout");
  G1Jac& acc1 = lds_acc1[threadIdx.x];
  G2Jac& acc2 = lds_acc2[threadIdx.x];
  jac_set_identity(acc1);
  jac_set_identity(acc2);
  uint8_t st = ST_OK;
  // as in k_aggregate: an out-of-range message index is IndexOutOfBounds, a decreasing offset pair an empty list
  if (m >= n_msgs) { st = ST_INDEX_OOB; m = 0; }
  if (hi < lo) { if (st == ST_OK) st = ST_INDEX_OOB; hi = lo; }
  uint64_t longest = hi - lo;
  for (int off = 32; off > 0; off >>= 1) {
    uint64_t other = __shfl_xor((unsigned long long)longest, off, BN_WAVE);
    longest = other > longest ? other : longest;
  }
  const bool use_sub = n_groups != 0 && longest > n_groups;             // wave-uniform
  const bool use_sub1 = use_sub && groups4 != 0;                        // ... signatures from the per-message tables as well
  uint32_t* my_masks = bn_agg_masks + (threadIdx.x >> 1) * mask_stride;
  bool dup = false;
  if (use_sub) {
    for (unsigned w = role; w < mask_stride; w += 2) my_masks[w] = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  }
#if !defined(BN_AGG_NO_BATCHED_WALK)
  if (use_sub1) {                          // wave-uniform: keys AND signatures come from tables — the walk only checks entries and sets mask bits
    // Round 6: the walk in BATCHES.  The old loop below took two entries per iteration, every lane loading both, each load's result needed
    // at once: index -> status bytes -> mask bit is a chain of dependent round trips, ~256 times per lane with nothing in flight beside it
    // (there are no product calls on this path, so loads CAN stay in flight).  Here a lane takes only ITS entries (list position = role mod 2),
    // eight at a time: eight index loads together, then sixteen status bytes together, then the bookkeeping (k_aggregate_pair 35.6 -> 30.8 ms
    // per 1 Mi tuples same box, profiles/r06_g_ab_agg_batched_walk.log).  The tuple's status is that of the
    // FIRST failing entry in list order: each lane keeps the position of its first failure and the pair takes the smaller of the two.
#ifndef BN_AGG_WALK_BATCH
#define BN_AGG_WALK_BATCH 8       // 2 / 4 / 8 measured: 31.0 / 30.9 / 30.7 ms (profiles/r06_g_ab_agg_walk_batch_size.log) — halving the loads per lane is what pays, not the depth
#endif
    constexpr int WB = BN_AGG_WALK_BATCH;
    uint64_t bad_pos = ~(uint64_t)0;
    uint32_t bad_st = ST_OK;
    for (uint64_t t = 0; t < longest; t += 2 * WB) {
      uint32_t sgn[WB];
      bool act[WB], valid[WB];
      uint8_t s1[WB], s2[WB];
#pragma unroll
      for (int j = 0; j < WB; ++j) {
        const uint64_t pos = lo + t + 2 * j + role;
        act[j] = pos < hi;
        sgn[j] = act[j] ? signer_idx[pos] : 0u;
      }
#pragma unroll
      for (int j = 0; j < WB; ++j) {
        valid[j] = act[j] && sgn[j] < n_signers;
        const uint32_t g = valid[j] ? sgn[j] : 0u;
        s1[j] = sig_pool.st[(size_t)m * n_signers + g];
        s2[j] = pk_pool.st[g];
      }
#pragma unroll
      for (int j = 0; j < WB; ++j) {
        uint32_t e_st = ST_OK;
        if (act[j] && !valid[j]) e_st = ST_INDEX_OOB;                         // IndexOutOfBounds
        else if (valid[j] && (s1[j] & 0x7f)) e_st = s1[j] & 0x7f;             // the signature's decode status first, then the key's
        else if (valid[j] && (s2[j] & 0x7f)) e_st = s2[j] & 0x7f;
        const uint64_t pos = lo + t + 2 * j + role;
        if (e_st != ST_OK && pos < bad_pos) { bad_pos = pos; bad_st = e_st; }
        if (valid[j]) {
          const uint32_t bit = 1u << (8u * ((sgn[j] >> 3) & 3u) + (sgn[j] & 7u));
          const uint32_t old = atomicOr(&my_masks[sgn[j] >> 5], bit);
          dup = dup || (old & bit) != 0;
        }
      }
    }
    const uint32_t p_lo = (uint32_t)bn_partner_word((int32_t)(uint32_t)bad_pos), p_hi = (uint32_t)bn_partner_word((int32_t)(uint32_t)(bad_pos >> 32));
    const uint32_t p_st = (uint32_t)bn_partner_word((int32_t)bad_st);
    const uint64_t p_pos = ((uint64_t)p_hi << 32) | p_lo;
    if (st == ST_OK) st = (uint8_t)(p_pos < bad_pos ? p_st : bad_st);
  } else
#endif
  for (uint64_t t = 0; t < longest; t += 2) {
    G2Affine pp[2];
    G1Affine sp;
    sp.x = fp_load_const(C_G1_GEN[0]); sp.y = fp_load_const(C_G1_GEN[1]); sp.inf = true;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      bool active = lo + t + e < hi;
      uint32_t sgn = active ? signer_idx[lo + t + e] : 0u;
      bool valid = active && sgn < n_signers;
      if (active && !valid && st == ST_OK) st = ST_INDEX_OOB;             // IndexOutOfBounds
      if (!valid) sgn = 0;
      size_t sj = (size_t)m * n_signers + sgn;
      uint8_t s1 = sig_pool.st[sj], s2 = pk_pool.st[sgn];
      if (valid && st == ST_OK && (s1 & 0x7f)) st = s1 & 0x7f;
      if (valid && st == ST_OK && (s2 & 0x7f)) st = s2 & 0x7f;
      if (!use_sub) {
        pp[e].x.c[0] = pool_load_fp(pk_pool, 0 + (int)role, sgn);
        pp[e].y.c[0] = pool_load_fp(pk_pool, 2 + (int)role, sgn);
        pp[e].inf = !valid || (s2 & 0x80);
      }
      if ((unsigned)e == role) {            // this lane's entry of the iteration: its signature, and its bit of the key masks
        if (!use_sub1) {
          sp.x = pool_load_fp(sig_pool, 0, sj); sp.y = pool_load_fp(sig_pool, 1, sj);
          sp.inf = !valid || (s1 & 0x80);
        }
        if (use_sub && valid) {
          const uint32_t bit = 1u << (8u * ((sgn >> 3) & 3u) + (sgn & 7u));
          const uint32_t old = atomicOr(&my_masks[sgn >> 5], bit);
          dup = dup || (old & bit) != 0;
        }
      }
    }
    if (!use_sub) {
      jac_accumulate_mem(acc2, pp[0]);
      jac_accumulate_mem(acc2, pp[1]);
    }
    if (!use_sub1) jac_accumulate_mem(acc1, sp);
  }
  if (use_sub) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const int32_t partner_dup = bn_partner_word((int32_t)dup);           // unconditionally: a DPP fetch must not sit behind a short-circuit
    dup = dup || partner_dup != 0;                                       // either lane of the pair saw a repeated signer
    // Two mask bytes (16 signers) per iteration.  Keys: ONE entry of the 16-signer table (wide2: k_pool_widen_g2) or the two 8-signer
    // entries; signatures (use_sub1): with the per-message 8-signer tables (wide1: k_pool_widen_g1) each lane of the pair takes ONE entry —
    // the real-part lane the low byte's, the imaginary-part lane the high byte's — else its nibble of both bytes from the 4-signer tables.
    // The records are named, not loaded: the additions fetch them themselves (PoolRec) — an Affine handed to a real function by reference
    // went through the private segment.  One word at either end of each record is touched first, all together, so that the fetches of an
    // iteration overlap (a real function waits for every outstanding load at its entry: the records then sit in the L1 when it reads them).
    const unsigned n_chunks = (n_groups + 1) / 2;
    // (Round 6, measured and not kept: this loop SOFTWARE-PIPELINED for the route of the largest batches — the records of chunk k + 1 loaded into
    // registers while chunk k is added, the additions reading registers — 35.85-35.92 against 35.48-35.53 ms per 1 Mi tuples same box,
    // profiles/r06_f_ab_agg_pipeline.log: the 36 more live registers cost more than the round trips they were meant to hide.)
    for (unsigned k = 0; k < n_chunks; ++k) {                              // wave-uniform
      const uint32_t m16 = dup ? 0u : (my_masks[k >> 1] >> (16u * (k & 1u))) & 0xFFFFu;
      const uint32_t byte0 = m16 & 255u, byte1 = m16 >> 8;
      const bool second = 2 * k + 1 < n_groups;                            // wave-uniform (false only in the last chunk of an odd group count: byte1 = 0)
      PoolRec k2[2], s1[2];
      unsigned nk2, ns1 = 0;
      if (wide2) {
        const size_t j = (size_t)k * 65536 + m16;
        k2[0] = {wide2_pool.planes + pool_word(wide2_pool, (int)role, j), m16 == 0 || (wide2_pool.st[j] & 0x80)};
        nk2 = 1;
      } else {
        const size_t j0 = (size_t)(2 * k) * 256 + byte0, j1 = (size_t)(second ? 2 * k + 1 : 2 * k) * 256 + byte1;
        k2[0] = {sub_pool.planes + pool_word(sub_pool, (int)role, j0), byte0 == 0 || (sub_pool.st[j0] & 0x80)};
        k2[1] = {sub_pool.planes + pool_word(sub_pool, (int)role, j1), byte1 == 0 || (sub_pool.st[j1] & 0x80)};
        nk2 = second ? 2 : 1;
      }
      if (use_sub1 && wide1) {
        const unsigned g = second ? 2 * k + role : 2 * k;                  // this lane's byte
        const uint32_t byte = (role && second) ? byte1 : (role ? 0u : byte0);
        const size_t j = ((size_t)m * n_groups + g) * 256 + byte;
        s1[0] = {wide1_pool.planes + pool_word(wide1_pool, 0, j), byte == 0 || (wide1_pool.st[j] & 0x80)};
        ns1 = 1;
      } else if (use_sub1) {
        const uint32_t nib0 = (byte0 >> (4u * role)) & 15u, nib1 = (byte1 >> (4u * role)) & 15u;
        const size_t j0 = (((size_t)m * groups4) + 2u * (2 * k) + role) * 16 + nib0;
        const size_t j1 = (((size_t)m * groups4) + 2u * (second ? 2 * k + 1 : 2 * k) + role) * 16 + nib1;
        s1[0] = {sub1_pool.planes + pool_word(sub1_pool, 0, j0), nib0 == 0 || (sub1_pool.st[j0] & 0x80)};
        s1[1] = {sub1_pool.planes + pool_word(sub1_pool, 0, j1), nib1 == 0 || (sub1_pool.st[j1] & 0x80)};
        ns1 = second ? 2 : 1;
      }
      {
        int32_t t = k2[0].p[0] ^ k2[0].p[2 * BN_LIMBS - 1];
        if (nk2 > 1) t ^= k2[1].p[0] ^ k2[1].p[2 * BN_LIMBS - 1];
        if (ns1 > 0) t ^= s1[0].p[0] ^ s1[0].p[2 * BN_LIMBS - 1];
        if (ns1 > 1) t ^= s1[1].p[0] ^ s1[1].p[2 * BN_LIMBS - 1];
        asm volatile("" ::"v"(t));
      }
      if (k == 0) {
        // the first entries SEED the sums (a load instead of an addition to the identity, which the streamed addition reports as exceptional)
        { G2Affine e; k2[0](e); jac_from_affine(acc2, e); }
        if (ns1 > 0) { G1Affine e; s1[0](e); jac_from_affine(acc1, e); }
      } else {
        jac_accumulate_from(acc2, k2[0]);
        if (ns1 > 0) jac_accumulate_from(acc1, s1[0]);
      }
      if (nk2 > 1) jac_accumulate_from(acc2, k2[1]);
      if (ns1 > 1) jac_accumulate_from(acc1, s1[1]);
    }
    if (__builtin_amdgcn_ballot_w64(dup) != 0) {                          // rare: the tuples with a repeated signer add their keys one by one
      uint64_t longest2 = dup ? hi - lo : 0;
      for (int off = 32; off > 0; off >>= 1) {
        uint64_t other = __shfl_xor((unsigned long long)longest2, off, BN_WAVE);
        longest2 = other > longest2 ? other : longest2;
      }
      for (uint64_t t = 0; t < longest2; ++t) {
        const bool active = dup && lo + t < hi;
        uint32_t sgn = active ? signer_idx[lo + t] : 0u;
        const bool valid = active && sgn < n_signers;
        if (!valid) sgn = 0;
        G2Affine p;
        p.x.c[0] = pool_load_fp(pk_pool, 0 + (int)role, sgn);
        p.y.c[0] = pool_load_fp(pk_pool, 2 + (int)role, sgn);
        p.inf = !valid || pk_pool.st[sgn] != 0;                            // decode errors are in `st` already; identity entries add nothing
        jac_accumulate_mem(acc2, p);
        if (use_sub1) {                                                    // their signatures too: entry t goes to the lane of parity t
          const size_t sj = (size_t)m * n_signers + sgn;
          G1Affine q;
          q.x = pool_load_fp(sig_pool, 0, sj); q.y = pool_load_fp(sig_pool, 1, sj);
          q.inf = !valid || sig_pool.st[sj] != 0 || (unsigned)(t & 1u) != role;
          jac_accumulate_mem(acc1, q);
        }
      }
    }
  }
  // G1: own partial sum + the partner's
  G1Jac other;
#pragma unroll
  for (int k = 0; k < BN_LIMBS; ++k) {
    other.x.v[k] = bn_partner_word(acc1.x.v[k]); other.y.v[k] = bn_partner_word(acc1.y.v[k]); other.z.v[k] = bn_partner_word(acc1.z.v[k]);
  }
  jac_add(acc1, acc1, other);
  G1Affine asig, h;
  G2Affine apk;
  jac_to_affine(asig, acc1);
  jac_to_affine(apk, acc2);
  if (!live) return;
  h.x = pool_load_fp(h_pool, 0, m); h.y = pool_load_fp(h_pool, 1, m); h.inf = false;
  ws_store_fp(ws, PL_QX0 + (int)role, i, apk.x.c[0]);
  ws_store_fp(ws, PL_QY0 + (int)role, i, apk.y.c[0]);
  if (role == 0) {
    ws_store_g1(ws, PL_P1X, BY_P1_INF, i, asig);
    ws_store_g1(ws, PL_P2X, BY_P2_INF, i, h);
    ws_byte(ws, BY_Q_INF, i) = apk.inf;
    ws_byte(ws, BY_ST_DECODE, i) = st;
    ws_byte(ws, BY_ST_HASH, i) = h_pool.st[m];
  }
}
int bn254_pair_aggregate(const uint32_t* tuple_msg, const uint64_t* tuple_off, const uint32_t* signer_idx, size_t n, size_t n_signers, size_t n_msgs,
                         Pool pk_pool, Pool sig_pool, Pool h_pool, Pool sub_pool, size_t n_groups, Pool sub1_pool, size_t groups4, Ws ws, hipStream_t s,
                         const uint32_t* perm, const Pool* wide2_pool, const Pool* wide1_pool) {
  const unsigned mask_stride = n_groups ? (unsigned)(((n_groups + 3) / 4) | 1u) : 1u;     // words per tuple, odd: the tuples of a wave hit different banks
  const size_t lds = n_groups ? (size_t)(BN_PAIR_WG / 2) * mask_stride * sizeof(uint32_t) : 0;
  unsigned blocks = (unsigned)((2 * n + BN_PAIR_WG - 1) / BN_PAIR_WG);
  if (perm) blocks = (blocks + 7u) & ~7u;            // the XCD-contiguous slot mapping needs a multiple of 8 (extra workgroups are idle)
  k_aggregate_pair<<<blocks, BN_PAIR_WG, lds, s>>>(tuple_msg, tuple_off, signer_idx, n, n_signers, n_msgs, pk_pool, sig_pool, h_pool, sub_pool,
                                                   (unsigned)n_groups, sub1_pool, (unsigned)groups4, mask_stride, ws, perm, wide2_pool ? *wide2_pool : sub_pool,
                                                   wide2_pool != nullptr, wide1_pool ? *wide1_pool : sub1_pool, wide1_pool != nullptr);
  HIP_TRY(hipGetLastError());
  return 0;
}
// ---- fixed-base scalar multiplication sk * G2::one() on lane pairs (PublicKey::from_private_key, /root/reference/src/types.rs:85-87) ----------
// Key derivation multiplies the FIXED generator, so the doublings can be tabulated: sk = sum d_j 16^j with signed digits d_j in [-8, 8]
// (65 of them) and sk * G = sum d_j (16^j G) — 65 mixed additions from the table T[j][d - 1] = d 16^j G (bn254_group.hip: g2_comb_build,
// once per context) where the 256-step ladder of k_g2_mul spends 256 doublings + 65 complete additions: 0.7 k Fq2 products per key instead of
// 3 k, on TWO lanes instead of one.  The scalar is a private key: a window's entry is found by reading all eight records of the window and
// keeping one by selects (no address and no branch depends on a digit), the digit's sign is a conditional negation, a zero digit adds the
// identity.  The accumulator starts at a fixed BLINDING point B (record 520) that is subtracted at the end, so that it is never the identity
// and never +- a table entry for any scalar an honest caller holds — the in-place addition's exceptional route (P = +-Q, handled by the
// complete formula behind a wave vote) then never runs, which keeps the schedule independent of the key; it still gives the right point if
// it ever does.
#define BN_G2_COMB_WINDOWS 65
#define BN_G2_COMB_BLIND (BN_G2_COMB_WINDOWS * 8)
template <class A> struct AffineValue {
  A v;
  __device__ __forceinline__ void operator()(A& q) const { q = v; }
};
__device__ __forceinline__ void store_fp_be_pair(uint8_t* b, const Fp& a) {
  U256 x = fp_to_u256(a);
  uint32_t* w = (uint32_t*)b;
#pragma unroll
  for (int k = 0; k < 8; ++k) w[k] = __builtin_bswap32(x.w[7 - k]);
}
KERNEL_PAIR void k_g2_mul_fixed_pair(const uint8_t* scalars, size_t n, int reduce, Pool comb, uint8_t* out, uint8_t* status) {
  const unsigned role = threadIdx.x & 1u;
  size_t i = ((size_t)blockIdx.x * BN_PAIR_WG + threadIdx.x) >> 1;
  const bool live = i < n;                         // no early return: the additions vote across the wave
  const size_t ii = live ? i : n - 1;
  uint32_t k[8];
  {
    const uint32_t* w = (const uint32_t*)(scalars + 32 * ii);
#pragma unroll
    for (int j = 0; j < 8; ++j) k[7 - j] = __builtin_bswap32(w[j]);
    if (reduce) {
      for (int it = 0; it < 6; ++it) {             // 2^256 / r < 6; the subtraction is applied by selects
        const bool ge = u256_geq(k, C_ORDER_R);
        uint32_t bw = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const uint64_t d = (uint64_t)k[j] - C_ORDER_R[j] - bw;
          bw = (uint32_t)(d >> 63);
          k[j] = ge ? (uint32_t)d : k[j];
        }
      }
    }
  }
  __shared__ G2Jac lds_acc[BN_PAIR_WG];
  G2Jac& acc = lds_acc[threadIdx.x];
  {
    G2Affine b;
    PoolRec{comb.planes + pool_word(comb, (int)role, BN_G2_COMB_BLIND), false}(b);
    jac_from_affine(acc, b);
  }
  int carry = 0;
#pragma unroll 1
  for (int j = 0; j < BN_G2_COMB_WINDOWS; ++j) {
    int v = (j < 64 ? (int)((k[j >> 3] >> (4 * (j & 7))) & 15u) : 0) + carry;
    carry = v > 8;
    const int d = v - 16 * carry, m = d < 0 ? -d : d;
    G2Affine q;
    PoolRec{comb.planes + pool_word(comb, (int)role, (size_t)j * 8), false}(q);
#pragma unroll 1
    for (int e = 1; e < 8; ++e) {                  // every record of the window is read; the wanted one is kept by selects
      G2Affine t;
      PoolRec{comb.planes + pool_word(comb, (int)role, (size_t)j * 8 + e), false}(t);
      q.x = fp2_select(m == e + 1, t.x, q.x);
      q.y = fp2_select(m == e + 1, t.y, q.y);
    }
    q.y = fp2_select(d < 0, fp2_neg(q.y), q.y);
    q.inf = m == 0;
    jac_accumulate_from(acc, AffineValue<G2Affine>{q});
  }
  {
    G2Affine b;
    PoolRec{comb.planes + pool_word(comb, (int)role, BN_G2_COMB_BLIND), false}(b);
    b.y = fp2_neg(b.y);
    jac_accumulate_from(acc, AffineValue<G2Affine>{b});
  }
  G2Affine r;
  jac_to_affine(r, acc);
  if (!live) return;
  uint8_t* o = out + 128 * i;
  if (r.inf) {
    uint32_t* w = (uint32_t*)(o + 32 * role);
    uint32_t* w2 = (uint32_t*)(o + 64 + 32 * role);
#pragma unroll
    for (int t = 0; t < 8; ++t) { w[t] = 0; w2[t] = 0; }
  } else {
    store_fp_be_pair(o + 32 * role, r.x.c[0]);
    store_fp_be_pair(o + 64 + 32 * role, r.y.c[0]);
  }
  if (role == 0) status[i] = ST_OK;
}
// The same for sk * G1::one() (PublicKeyG1::from_private_key, /root/reference/src/types.rs:155-157).  G1 lives over Fq, so the two lanes of a
// pair are two independent adders: the real-part lane takes the even windows, the imaginary-part lane the odd ones, each into its own
// blinded accumulator (records 520 / 521), the partner's sum is added at the end and record 522 = -(B0 + B1) removes the blinding.
#define BN_G1_COMB_BLIND0 (BN_G2_COMB_WINDOWS * 8)
KERNEL_PAIR void k_g1_mul_fixed_pair(const uint8_t* scalars, size_t n, int reduce, Pool comb, uint8_t* out, uint8_t* status) {
  const unsigned role = threadIdx.x & 1u;
  size_t i = ((size_t)blockIdx.x * BN_PAIR_WG + threadIdx.x) >> 1;
  const bool live = i < n;
  const size_t ii = live ? i : n - 1;
  uint32_t k[8];
  {
    const uint32_t* w = (const uint32_t*)(scalars + 32 * ii);
#pragma unroll
    for (int j = 0; j < 8; ++j) k[7 - j] = __builtin_bswap32(w[j]);
    if (reduce) {
      for (int it = 0; it < 6; ++it) {
        const bool ge = u256_geq(k, C_ORDER_R);
        uint32_t bw = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const uint64_t d = (uint64_t)k[j] - C_ORDER_R[j] - bw;
          bw = (uint32_t)(d >> 63);
          k[j] = ge ? (uint32_t)d : k[j];
        }
      }
    }
  }
  __shared__ G1Jac lds_acc[BN_PAIR_WG];
  G1Jac& acc = lds_acc[threadIdx.x];
  {
    G1Affine b;
    PoolRec{comb.planes + pool_word(comb, 0, BN_G1_COMB_BLIND0 + role), false}(b);
    jac_from_affine(acc, b);
  }
  int carry = 0;
#pragma unroll 1
  for (int j = 0; j < BN_G2_COMB_WINDOWS; ++j) {     // both lanes recode every digit (the carries run through all windows); a lane ADDS its own windows only
    int v = (j < 64 ? (int)((k[j >> 3] >> (4 * (j & 7))) & 15u) : 0) + carry;
    carry = v > 8;
    const int d = v - 16 * carry, m = d < 0 ? -d : d;
    if (((unsigned)j & 1u) != role) continue;        // a constant of the lane's position, not of the key
    G1Affine q;
    PoolRec{comb.planes + pool_word(comb, 0, (size_t)j * 8), false}(q);
#pragma unroll 1
    for (int e = 1; e < 8; ++e) {
      G1Affine t;
      PoolRec{comb.planes + pool_word(comb, 0, (size_t)j * 8 + e), false}(t);
      q.x = fp_select(m == e + 1, t.x, q.x);
      q.y = fp_select(m == e + 1, t.y, q.y);
    }
    q.y = fp_select(d < 0, fp_neg(q.y), q.y);
    q.inf = m == 0;
    jac_accumulate_from(acc, AffineValue<G1Affine>{q});
  }
  G1Jac other;
#pragma unroll
  for (int t = 0; t < BN_LIMBS; ++t) {
    other.x.v[t] = bn_partner_word(acc.x.v[t]); other.y.v[t] = bn_partner_word(acc.y.v[t]); other.z.v[t] = bn_partner_word(acc.z.v[t]);
  }
  jac_add(acc, acc, other);
  {
    G1Affine b;
    PoolRec{comb.planes + pool_word(comb, 0, BN_G1_COMB_BLIND0 + 2), false}(b);
    jac_accumulate_from(acc, AffineValue<G1Affine>{b});
  }
  G1Affine r;
  jac_to_affine(r, acc);
  if (!live) return;
  uint8_t* o = out + 64 * i;
  if (role == 0) {                                   // both lanes hold the same point: the real-part lane writes x, the other y
    if (r.inf) { uint32_t* w = (uint32_t*)o; for (int t = 0; t < 8; ++t) w[t] = 0; } else store_fp_be_pair(o, r.x);
    status[i] = ST_OK;
  } else {
    if (r.inf) { uint32_t* w = (uint32_t*)(o + 32); for (int t = 0; t < 8; ++t) w[t] = 0; } else store_fp_be_pair(o + 32, r.y);
  }
}
int bn254_pair_g1_mul_fixed(const uint8_t* d_scalars, size_t n, int reduce, Pool comb, uint8_t* d_out, uint8_t* d_status, hipStream_t s) {
  k_g1_mul_fixed_pair<<<(unsigned)((2 * n + BN_PAIR_WG - 1) / BN_PAIR_WG), BN_PAIR_WG, 0, s>>>(d_scalars, n, reduce, comb, d_out, d_status);
  HIP_TRY(hipGetLastError());
  return 0;
}
int bn254_pair_g2_mul_fixed(const uint8_t* d_scalars, size_t n, int reduce, Pool comb, uint8_t* d_out, uint8_t* d_status, hipStream_t s) {
  k_g2_mul_fixed_pair<<<(unsigned)((2 * n + BN_PAIR_WG - 1) / BN_PAIR_WG), BN_PAIR_WG, 0, s>>>(d_scalars, n, reduce, comb, d_out, d_status);
  HIP_TRY(hipGetLastError());
  return 0;
}
int bn254_pair_miller_verify(size_t n, Ws ws, const uint32_t* map, const uint32_t* count, hipStream_t s, int mode) {
  k_miller_verify_pair<<<(unsigned)((2 * n + BN_PAIR_WG - 1) / BN_PAIR_WG), BN_PAIR_WG, 0, s>>>(n, ws, map, count, mode);
  HIP_TRY(hipGetLastError());
  return 0;
}
int bn254_pair_miller_var(size_t n, Ws ws, hipStream_t s) {
  k_miller_var_pair<<<(unsigned)((2 * n + BN_PAIR_WG - 1) / BN_PAIR_WG), BN_PAIR_WG, 0, s>>>(n, ws);
  HIP_TRY(hipGetLastError());
  return 0;
}
