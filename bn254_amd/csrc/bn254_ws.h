// Workspace layout shared by the two device translation units of libbn254hip.so (bn254_hip.hip: one item per
// lane; bn254_pair.hip: one verify per lane PAIR).  Device-side only; include after the field headers.
#pragma once
#include <stddef.h>

#define BN_WAVE 64
#define BN_SPLIT_MAX_N ((size_t)98304)   // <= 1.5 waves per SIMD with one lane per verify
#define TRIO_MAX_BATCH_DEFAULT 16384               // octet layout up to TWO passes of one wave per SIMD (1024 SIMDs x 8 verifies): 3.5 ms at 8192, 6.5 ms at 16384 (lane pairs: 7.1 / 7.9 ms)
#define NONET_WIDE_MAX_BATCH 1024                  // ... and up to this size on EIGHTEEN lane pairs, one verify per wave: the 18 products of a multiplication in one round (0.63 -> 0.56 ms)
#define NONET_MAX_BATCH_DEFAULT 3072               // final exponentiation on nine lane pairs per verify up to this batch size: ONE pass of 3 verifies per wave on 1024 SIMDs (0.64 ms against the octet layout's 1.09; a second pass would cost 1.3)
#define LM_MAX_BATCH_DEFAULT 1536                   // Miller loop as the lane machine (bn254_lmiller.hip: nine lane pairs in each of four waves per verify, 3 verifies per workgroup, one workgroup per CU) up to this batch size: two passes over 256 CUs (0.43 ms one pass, 0.86 two; eight wave roles: 1.0); 0 = never
// ---- batch size -> kernel layout: the ONE routing table of the verify-shaped entry points ---------------------------------------------
// (verify, verify from compressed encodings, check_public_keys, aggregate verify's pairing part, keyed verify below its table route.)
// Four thresholds — the *_DEFAULT values above, overridden per context by BN254_OPT_LM_MAX_BATCH / _NONET_MAX_BATCH / _NONET_WIDE /
// _TRIO_MAX_BATCH — span the table; bn_route() is the only place that turns a batch size into layouts, bn_route_table() lists its rows
// (what bn254_debug_route_table hands to the tests, which generate every boundary +-1 from it instead of listing sizes by hand).
//   default rows:  n <=  1 024  lane machine      + eighteen lane pairs      (one verify per SIMD)
//                  n <=  1 536  lane machine      + nine lane pairs
//                  n <=  3 072  eight wave roles  + nine lane pairs
//                  n <= 16 384  eight wave roles  + octets
//                  above        lane pairs        + lane pairs               (two waves per SIMD: throughput)
enum BnMillerLayout { BN_ML_LANE_MACHINE = 0, BN_ML_WAVE_ROLES = 1, BN_ML_LANE_PAIRS = 2 };
enum BnFeLayout { BN_FE_NONET_WIDE = 0, BN_FE_NONET = 1, BN_FE_OCTET = 2, BN_FE_LANE_PAIRS = 3 };
struct BnRouteLimits { size_t lm_max, nonet_wide_max, nonet_max, small_max; };   // 0 = that layout is off; small_max bounds the whole small-batch family
struct BnRoute { int miller, fe; };
static inline BnRoute bn_route(const BnRouteLimits& L, size_t n) {
  BnRoute r = {BN_ML_LANE_PAIRS, BN_FE_LANE_PAIRS};
  if (n > L.small_max) return r;
  r.miller = n <= L.lm_max ? BN_ML_LANE_MACHINE : BN_ML_WAVE_ROLES;
  r.fe = n <= L.nonet_max ? (n <= L.nonet_wide_max ? BN_FE_NONET_WIDE : BN_FE_NONET) : BN_FE_OCTET;
  return r;
}
// rows (max_n[i], route[i]) in ascending order of max_n, the last one max_n = SIZE_MAX; returns the number of rows (<= 5)
static inline int bn_route_table(const BnRouteLimits& L, size_t* max_n, BnRoute* route, int cap) {
  size_t b[4] = {L.lm_max, L.nonet_wide_max, L.nonet_max, L.small_max};
  for (int i = 0; i < 4; ++i) if (b[i] > L.small_max) b[i] = L.small_max;
  for (int i = 0; i < 4; ++i) for (int j = i + 1; j < 4; ++j) if (b[j] < b[i]) { size_t t = b[i]; b[i] = b[j]; b[j] = t; }
  int rows = 0;
  for (int i = 0; i < 4 && rows < cap - 1; ++i) {
    if (b[i] == 0 || (i && b[i] == b[i - 1])) continue;
    const BnRoute r = bn_route(L, b[i]);
    if (rows && route[rows - 1].miller == r.miller && route[rows - 1].fe == r.fe) { max_n[rows - 1] = b[i]; continue; }
    max_n[rows] = b[i]; route[rows] = r; ++rows;
  }
  max_n[rows] = (size_t)-1; route[rows] = bn_route(L, (size_t)-1); ++rows;
  return rows;
}
#ifndef BN_WS_ROUTE_ONLY      /* (a host-side test includes the part above alone: tests/test_abi.py::test_routing_table_rows) */
#define TRIO_WAVE_ROLES_DEFAULT 2                  // ... with the Miller loop as wave roles: 2 = eight waves per 32 verifies (k_miller_verify_w8), 1 = four
#define AGG_SUBSET_MIN_TUPLES_DEFAULT 4096       // aggregate verify: subset-sum table of the key pool from this many tuples on (the table costs ~0.3 ms)
#define AGG_SUBSET_MAX_SIGNERS ((size_t)2048)     // ... one mask byte per group of 8 keys and tuple in LDS: 256 groups at most
#define AGG_SUBSET_G1_TUPLES_PER_MSG ((size_t)64)  // ... signature tables (per message) once a message is shared by this many tuples on average
#define AGG_SUBSET_G1_MAX_BYTES ((size_t)8 << 30)  // ... and while they stay below 8 GB of HBM
#define AGG_WIDE_MIN_TUPLES_DEFAULT 262144         // ... tables of twice the window (keys: 16 signers per entry, signatures: 8) from this many tuples on: half the additions
                                                   //     per tuple for ~6 ms of table building per call (k_pool_widen_g2 / _g1); 0 = never
#define AGG_WIDE_G1_TUPLES_PER_MSG ((size_t)512)   // ... the signature half only when a message's (16 x larger) table is shared by this many tuples
#define AGG_WIDE_G2_MAX_BYTES ((size_t)2 << 30)    // ... and the tables stay below these budgets of HBM
#define PINNED_STAGING_DEFAULT 0                    // host-pointer verify: threads copying through the pinned staging buffer; 0 = off (A/B in profiles/r04_*)
#define PINNED_STAGING_MIN_N ((size_t)8192)        // ... only batches whose transfer is worth overlapping
#define RAND_MIN_BATCH_DEFAULT 131072              // randomised verify pays off from about here
#define RAND_TWO_PER_LANE_MIN_N ((size_t)131072)   // randomised verify: two items per lane once that still fills 1024 SIMDs
// Register budget: amdgpu_waves_per_eu(W, W) on the kernels is propagated to every device function
// they call (AMDGPU attributor), capping VGPR+AGPR at 512/W so that W waves fit on each SIMD.
// In a pure-VALU microbenchmark two co-resident waves each keep the full single-wave issue rate
// (profiles/r01_issue_mix_microbench.jsonl), but the Fq12 bodies need ~400 live registers: at W = 2
// they spill to the private segment and every kernel got slower (profiles/r01_c_ab_occupancy.log).
// W = 1 (512 registers per lane) is the measured optimum for this code shape.
#ifndef BN_WAVES_PER_EU
#define BN_WAVES_PER_EU 1
#endif
#define KERNEL __global__ __launch_bounds__(BN_WAVE) __attribute__((amdgpu_waves_per_eu(BN_WAVES_PER_EU, BN_WAVES_PER_EU)))
// kernels whose per-lane state is a few field elements (hash rounds, decoders, encoders) fit 256
// registers without spilling and gain from a second wave per SIMD (hash: 1.34 -> 1.15 ms per 65 536)
#define KERNEL_SMALL __global__ __launch_bounds__(BN_WAVE) __attribute__((amdgpu_waves_per_eu(2, 2)))

// ------------------------------------------------------------------------------------------
// workspace planes
// ------------------------------------------------------------------------------------------
struct Ws {
  int32_t* planes;    // [N_PLANES * BN_LIMBS][stride] i32
  uint8_t* bytes;     // [N_BYTE_PLANES][stride]
  size_t stride;
  // hash-to-G1 round state (see k_hash_round)
  uint32_t* h_best;   // [stride]  smallest successful counter of the current round, or HASH_NONE
  uint8_t* h_next;    // [stride]  first counter not yet tried
  uint32_t* h_list;   // [2][stride] compacted indices of the messages still without a point
  uint32_t* h_cnt;    // [HASH_MAX_ROUNDS + 1] number of entries of the list feeding round r
  // clock probe (BN254_OPT_CLOCK_PROBE; nullptr = off): [3 slots: Miller loop, final exponentiation, probes][BN_CLK_MAX_WG][2] — per workgroup
  // index the shader-clock cycles (s_memtime) and the constant-rate ticks (s_memrealtime) its first lane saw between entry and exit,
  // ACCUMULATED over every launch since bn254_ctx_last_clocks last read (and cleared) them: their ratio is the clock the chip actually
  // ran those kernels at — over exactly the launches in between, e.g. the timed steps of bench.py (roofline.effective_sclk_mhz)
  unsigned long long* clk;
};
#define BN_CLK_MAX_WG 4096
#define BN_CLK_BEGIN(ws)                                                                         \
  unsigned long long clk0_ = 0, wall0_ = 0;                                                      \
  if ((ws).clk && threadIdx.x == 0) { clk0_ = clock64(); wall0_ = wall_clock64(); }
#define BN_CLK_END(ws, slot)                                                                     \
  do {                                                                                           \
    if ((ws).clk && threadIdx.x == 0 && blockIdx.x < BN_CLK_MAX_WG) {                            \
      unsigned long long* p_ = (ws).clk + ((size_t)(slot) * BN_CLK_MAX_WG + blockIdx.x) * 2;    \
      p_[0] += clock64() - clk0_; p_[1] += wall_clock64() - wall0_;                              \
    }                                                                                            \
  } while (0)
#define HASH_NONE 0xFFFFFFFFu
#define HASH_DONE 0xFFFFFFFEu                    // k_hash_direct has already written the point of this message
#define HASH_DIRECT_WIDTH_DEFAULT 32             // counters tried at once per message by k_hash_direct (lanes of one wave; 1, 2, .. 32)
#define HASH_DIRECT_MAX_N ((size_t)4096)         // ... for batches that leave the chip mostly idle: 32 n lanes <= two waves per SIMD
#define HASH_MAX_ROUNDS 64
#define HASH_MAX_GRID_LANES ((size_t)1 << 24)   // lanes launched per round at most (grid-stride beyond)
#ifndef HASH_TARGET_LANES
#define HASH_TARGET_LANES ((size_t)1 << 17)    // ~2 waves per SIMD
#endif
enum { PL_P1X = 0, PL_P1Y, PL_QX0, PL_QX1, PL_QY0, PL_QY1, PL_P2X, PL_P2Y, PL_HASHX, PL_HASHY, PL_F0, N_PLANES = PL_F0 + 12 };
enum { BY_ST_DECODE = 0, BY_ST_HASH, BY_P1_INF, BY_Q_INF, BY_P2_INF, BY_A_INF, N_BYTE_PLANES };

__device__ __forceinline__ Fp ws_load_fp(const Ws& ws, int plane, size_t i) {
  Fp r;
#pragma unroll
  for (int k = 0; k < BN_LIMBS; ++k) r.v[k] = ws.planes[((size_t)plane * BN_LIMBS + k) * ws.stride + i];
  return r;
}
__device__ __forceinline__ void ws_store_fp(const Ws& ws, int plane, size_t i, const Fp& a) {
#pragma unroll
  for (int k = 0; k < BN_LIMBS; ++k) ws.planes[((size_t)plane * BN_LIMBS + k) * ws.stride + i] = a.v[k];
}
__device__ __forceinline__ uint8_t& ws_byte(const Ws& ws, int plane, size_t i) { return ws.bytes[(size_t)plane * ws.stride + i]; }

__device__ __forceinline__ void ws_store_g1(const Ws& ws, int px, int inf_plane, size_t i, const G1Affine& p) {
  ws_store_fp(ws, px, i, p.x); ws_store_fp(ws, px + 1, i, p.y); ws_byte(ws, inf_plane, i) = p.inf;
}
__device__ __forceinline__ void ws_load_g1(const Ws& ws, int px, int inf_plane, size_t i, G1Affine& p) {
  p.x = ws_load_fp(ws, px, i); p.y = ws_load_fp(ws, px + 1, i); p.inf = ws_byte(ws, inf_plane, i) != 0;
}

// Pools (aggregate verify) are decoded once into RECORDS: everything one lane reads of an entry is contiguous, because the
// aggregation kernels GATHER entries by index — with the limb-major planes of the workspace a point cost 18 cache lines per
// lane (1 152 distinct lines per wave-load, more than the L1 holds), with records it costs one or two.
//   G1 entry (2 coordinates):  [x (9 words) | y (9) | 2 pad]                                   = 20 words (80 B, 16-byte aligned)
//   G2 entry (4 coordinates):  [x.re (9) | y.re (9) | 2 pad] [x.im (9) | y.im (9) | 2 pad]     = 40 words: a lane of the pair
//                              layout reads the half of its role
// coordinate index e as before: G1 0 = x, 1 = y; G2 0 = x.re, 1 = x.im, 2 = y.re, 3 = y.im.  Status byte per entry.
#define BN_POOL_HALF_WORDS 20
struct Pool { int32_t* planes; uint8_t* st; size_t stride; uint32_t g2; };
__device__ __forceinline__ size_t pool_word(const Pool& p, int e, size_t j) {
  return p.g2 ? j * (2 * BN_POOL_HALF_WORDS) + (size_t)((e & 1) * BN_POOL_HALF_WORDS + (e >> 1) * BN_LIMBS) : j * BN_POOL_HALF_WORDS + (size_t)(e * BN_LIMBS);
}
__device__ __forceinline__ Fp pool_load_fp(const Pool& p, int e, size_t j) {
  Fp r;
  const int32_t* w = p.planes + pool_word(p, e, j);
#pragma unroll
  for (int k = 0; k < BN_LIMBS; ++k) r.v[k] = w[k];
  return r;
}
__device__ __forceinline__ void pool_store_fp(const Pool& p, int e, size_t j, const Fp& a) {
  int32_t* w = p.planes + pool_word(p, e, j);
#pragma unroll
  for (int k = 0; k < BN_LIMBS; ++k) w[k] = a.v[k];
}
#define HIP_TRY(expr)                                      \
  do {                                                     \
    hipError_t e_ = (expr);                                \
    if (e_ != hipSuccess) return -(int)e_;                 \
  } while (0)

// entry points of bn254_pair.hip (internal to the library)
__attribute__((visibility("hidden"))) int bn254_pair_miller_verify(size_t n, Ws ws, const uint32_t* map, const uint32_t* count, hipStream_t s,
                                                                   int mode = 0);
__attribute__((visibility("hidden"))) int bn254_pair_miller_var(size_t n, Ws ws, hipStream_t s);
__attribute__((visibility("hidden"))) int bn254_pair_leaf_floor(size_t n, Ws ws, hipStream_t s, int mode);
__attribute__((visibility("hidden"))) int bn254_pair_fe_program(size_t n, Ws ws, const unsigned char* prog, hipStream_t s);   // measurement only      // measurement only
__attribute__((visibility("hidden"))) int bn254_pair_final_exp_product(size_t n, size_t k, Ws ws, uint8_t* gt_out, uint8_t* status_out, int raw_only,
                                                                       hipStream_t s);
__attribute__((visibility("hidden"))) int bn254_pair_final_exp(size_t n, Ws ws, int use_hash, uint8_t* status_out, const uint32_t* map,
                                                               const uint32_t* count, hipStream_t s, size_t base = 0);
__attribute__((visibility("hidden"))) int bn254_pair_miller_rand(size_t n, size_t n_groups, int items_per_pair, Ws ws, size_t gbase, hipStream_t s);
__attribute__((visibility("hidden"))) int bn254_pair_rand_tail(size_t n_groups, Ws ws, size_t gbase, hipStream_t s);
__attribute__((visibility("hidden"))) int bn254_pair_aggregate(const uint32_t* tuple_msg, const uint64_t* tuple_off, const uint32_t* signer_idx, size_t n,
                                                               size_t n_signers, size_t n_msgs, Pool pk_pool, Pool sig_pool, Pool h_pool, Pool sub_pool,
                                                               size_t n_groups, Pool sub1_pool, size_t groups4, Ws ws, hipStream_t s,
                                                               const uint32_t* perm = nullptr, const Pool* wide2_pool = nullptr,
                                                               const Pool* wide1_pool = nullptr);
__attribute__((visibility("hidden"))) int bn254_pair_g2_mul_fixed(const uint8_t* d_scalars, size_t n, int reduce, Pool comb, uint8_t* d_out, uint8_t* d_status,
                                                                  hipStream_t s);      // sk * G2::one() from the comb table of the generator (520 + 1 records)
#define BN_G2_COMB_RECORDS (65 * 8 + 1)
__attribute__((visibility("hidden"))) int bn254_pair_g1_mul_fixed(const uint8_t* d_scalars, size_t n, int reduce, Pool comb, uint8_t* d_out, uint8_t* d_status,
                                                                  hipStream_t s);      // sk * G1::one(): 520 records + the two blinding points + minus their sum
#define BN_G1_COMB_RECORDS (65 * 8 + 3)
__attribute__((visibility("hidden"))) int bn254_pair_decode_g2(const uint8_t* pts, size_t n, uint32_t flags, Ws ws, int accumulate, hipStream_t s);
__attribute__((visibility("hidden"))) int bn254_pair_decompress_g2(const uint8_t* in, size_t n, Ws ws, hipStream_t s, int skip_subgroup_test = 0);

// keyed verify: the registered keys of a context — per key the 87 lines of its Miller loop in the c2 = 1 form,
// lines[key][line][coefficient c0 / c1][re / im][limb] (canonical limbs; 12.5 KB per key), its decode status and identity flag
#define BN_KEY_LINE_WORDS (2 * 2 * BN_LIMBS)
struct KeyTable { const int32_t* lines; const uint8_t* st; const uint8_t* inf; uint32_t n_keys; };
__attribute__((visibility("hidden"))) int bn254_lm_miller_verify_keyed(size_t n, Ws ws, const uint32_t* key_idx, KeyTable kt, hipStream_t s);   // bn254_lmiller.hip: the smallest batches
__attribute__((visibility("hidden"))) int bn254_pair_miller_verify_keyed(size_t n, Ws ws, const uint32_t* key_idx, KeyTable kt, hipStream_t s, size_t base = 0,
                                                                         const uint32_t* map = nullptr, const uint32_t* count = nullptr);

// entry points of bn254_trio.hip (octet layout for small batches)
__attribute__((visibility("hidden"))) int bn254_trio_miller_verify(size_t n, Ws ws, hipStream_t s, int mode = 0);
// entry points of bn254_quad.hip (the same with the four lane pairs of a verify as four waves with roles)
__attribute__((visibility("hidden"))) int bn254_quad_miller_verify(size_t n, Ws ws, hipStream_t s, int mode = 0);
__attribute__((visibility("hidden"))) int bn254_w8_miller_verify(size_t n, Ws ws, hipStream_t s, int mode = 0);      // eight waves per 32 verifies
// can the current device hold one workgroup of the small-batch kernels with the dynamic LDS they ask for (up to 156 KB)?
// Asked of the runtime's occupancy calculator at context creation; a runtime that cannot answer counts as "yes".
__attribute__((visibility("hidden"))) bool bn254_trio_fits_device();
__attribute__((visibility("hidden"))) bool bn254_quad_fits_device(int eight_waves);
__attribute__((visibility("hidden"))) int bn254_trio_final_exp(size_t n, Ws ws, int use_hash, uint8_t* status_out, hipStream_t s);
// entry points of bn254_lmiller.hip (Miller loop of the smallest batches as the lane machine: nine lane pairs in each of four waves per verify)
__attribute__((visibility("hidden"))) bool bn254_lm_fits_device();
__attribute__((visibility("hidden"))) int bn254_lm_miller_verify(size_t n, Ws ws, hipStream_t s, int mode = 0);
__attribute__((visibility("hidden"))) int bn254_lm_g2_subgroup(size_t n, Ws ws, hipStream_t s, int fail_status = 4 /* ST_INVALID_GROUP_POINT; the compressed decoders report ST_NOT_MEMBER */);   // the G2 subgroup test of points already decoded into the Q planes
// entry points of bn254_nonet.hip (final exponentiation of the smallest batches on nine lane pairs per verify)
__attribute__((visibility("hidden"))) bool bn254_nonet_fits_device();
__attribute__((visibility("hidden"))) int bn254_nonet_final_exp(size_t n, Ws ws, int use_hash, uint8_t* status_out, hipStream_t s, int wide = 0);
__attribute__((visibility("hidden"))) int bn254_nonet_final_exp_product(size_t n, size_t k, Ws ws, uint8_t* gt_out, uint8_t* status_out, hipStream_t s);   // bn254_batch_pairing*, n <= NONET_WIDE_MAX_BATCH
#endif  // BN_WS_ROUTE_ONLY
