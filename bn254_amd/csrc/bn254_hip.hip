// libbn254hip.so — HIP kernels for gfx950 (MI355X) + the C ABI declared in include/bn254_hip.h.
//
// Execution model: a batch is processed by a short chain of kernels that hand per-item state to each other
// through an HBM workspace laid out limb-major ("planes"): word k of field element e of item i lives at
//   ws[(e*9 + k) * stride + i]       (9 x 29-bit balanced limbs per field element)
// so a wave reads/writes contiguous bytes per limb (fully coalesced), and the caller-facing byte formats (AoS,
// big-endian) are touched exactly once on the way in/out.  Fq-level work (decoding, hash-to-G1, G1 arithmetic)
// runs one item per lane in this translation unit; everything built on the Fq2 tower (Miller loops, final
// exponentiation, G2 sums and subgroup tests) runs one item per LANE PAIR in bn254_pair.hip, with the
// one-lane-per-item kernels of this file kept behind BN254_OPT_PAIR_LANES = 0.
//
//   batch_verify:  k_decode_g1, k_decode_g2 -> k_hash_init/round/resolve/finish -> k_miller_verify_pair -> k_final_exp_pair
//
// HBM traffic per verify is 225 B of input/output + 2 x ~1 KB of workspace hand-off against ~19 k Montgomery
// products: the path is bound by VALU integer-multiply issue, not by HBM.
#include <hip/hip_runtime.h>
#include <thread>
#include <vector>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "../../include/bn254_hip.h"
#include "bn254_hash.h"
#include "bn254_io.h"
#include "bn254_pairing.h"

using namespace bn254;

#include "bn254_ws.h"

__device__ __forceinline__ void ws_store_f12(const Ws& ws, size_t i, const Fp12& f) {
  const Fp2* c[6] = {&f.c0.c0, &f.c0.c1, &f.c0.c2, &f.c1.c0, &f.c1.c1, &f.c1.c2};
#pragma unroll
  for (int k = 0; k < 6; ++k) { ws_store_fp(ws, PL_F0 + 2 * k, i, c[k]->c0); ws_store_fp(ws, PL_F0 + 2 * k + 1, i, c[k]->c1); }
}
__device__ __forceinline__ void ws_load_f12(const Ws& ws, size_t i, Fp12& f) {
  Fp2* c[6] = {&f.c0.c0, &f.c0.c1, &f.c0.c2, &f.c1.c0, &f.c1.c1, &f.c1.c2};
#pragma unroll
  for (int k = 0; k < 6; ++k) { c[k]->c0 = ws_load_fp(ws, PL_F0 + 2 * k, i); c[k]->c1 = ws_load_fp(ws, PL_F0 + 2 * k + 1, i); }
}
__device__ __forceinline__ void ws_store_g2(const Ws& ws, size_t i, const G2Affine& q) {
  ws_store_fp(ws, PL_QX0, i, q.x.c0); ws_store_fp(ws, PL_QX1, i, q.x.c1);
  ws_store_fp(ws, PL_QY0, i, q.y.c0); ws_store_fp(ws, PL_QY1, i, q.y.c1);
  ws_byte(ws, BY_Q_INF, i) = q.inf;
}
__device__ __forceinline__ void ws_load_g2(const Ws& ws, size_t i, G2Affine& q) {
  q.x.c0 = ws_load_fp(ws, PL_QX0, i); q.x.c1 = ws_load_fp(ws, PL_QX1, i);
  q.y.c0 = ws_load_fp(ws, PL_QY0, i); q.y.c1 = ws_load_fp(ws, PL_QY1, i);
  q.inf = ws_byte(ws, BY_Q_INF, i) != 0;
}
// a lane whose input failed to decode walks the rest of the pipeline on the generators so that
// every wave stays convergent; its status byte keeps the decode error.
__device__ __forceinline__ void g1_set_generator(G1Affine& p) { p.x = fp_load_const(C_G1_GEN[0]); p.y = fp_load_const(C_G1_GEN[1]); p.inf = false; }
__device__ __forceinline__ void g2_set_generator(G2Affine& q) { q.x = fp2_load_const(C_G2_GEN[0]); q.y = fp2_load_const(C_G2_GEN[1]); q.inf = false; }
// the coordinates of the generator with the identity flag untouched (a stand-in for arithmetic that must not meet (0, 0))
__device__ __forceinline__ void g2_set_generator_keep_inf(G2Affine& q) { q.x = fp2_load_const(C_G2_GEN[0]); q.y = fp2_load_const(C_G2_GEN[1]); }

// The Miller accumulator f (12 field elements = 432 B per lane) is the hottest per-lane state: every
// Fq12 squaring / line multiplication reads and rewrites it.  It is staged in LDS, one padded slot per
// lane (109 words: an odd word stride keeps the 64 lanes of a wave on distinct banks), so those
// accesses never leave the CU.  28 KB per 64-lane workgroup -> 5 workgroups per 160 KB CU.
struct Fp12Slot { Fp12 v; int32_t pad; };
static_assert(sizeof(Fp12Slot) == (12 * BN_LIMBS + 1) * 4 && ((12 * BN_LIMBS + 1) & 1), "LDS slot: 12 x 9 limbs + 1 pad word (odd stride)");

// ------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------
// decode n G1 points (64 B each) into planes (px, px+1) + inf byte plane; status into st_plane
// (first error wins if `accumulate`)
KERNEL_SMALL void k_decode_g1(const uint8_t* pts, size_t n, uint32_t flags, Ws ws, int px, int inf_plane, int accumulate) {
  size_t i = (size_t)blockIdx.x * BN_WAVE + threadIdx.x;
  if (i >= n) return;
  G1Affine p;
  uint8_t st = decode_g1(p, pts + 64 * i, flags);
  if (st != ST_OK) g1_set_generator(p);
  ws_store_g1(ws, px, inf_plane, i, p);
  uint8_t prev = accumulate ? ws_byte(ws, BY_ST_DECODE, i) : (uint8_t)ST_OK;
  ws_byte(ws, BY_ST_DECODE, i) = prev != ST_OK ? prev : st;
}
KERNEL void k_decode_g2(const uint8_t* pts, size_t n, uint32_t flags, Ws ws, int accumulate) {
  size_t i = (size_t)blockIdx.x * BN_WAVE + threadIdx.x;
  if (i >= n) return;
  G2Affine q;
  uint8_t st = decode_g2(q, pts + 128 * i, flags);
  if (st != ST_OK) g2_set_generator(q);
  if (flags & FLAG_G2_SUBGROUP_CHECK) {   // wave-uniform branch; every lane runs the ladder
    bool in = g2_in_subgroup(q);
    if (st == ST_OK && !in) { st = ST_INVALID_GROUP_POINT; g2_set_generator(q); }
  }
  ws_store_g2(ws, i, q);
  uint8_t prev = accumulate ? ws_byte(ws, BY_ST_DECODE, i) : (uint8_t)ST_OK;
  ws_byte(ws, BY_ST_DECODE, i) = prev != ST_OK ? prev : st;
}

// Keyed verify, registration: key j is decoded like PublicKey::from_uncompressed does (/root/reference/src/types.rs:96-99
// -> src/utils.rs:107-116; the subgroup check of AffineG2::new ALWAYS runs here, whatever the caller's flags: the table form
// below relies on it) and the 87 lines of its Miller loop are written in the c2 = 1 form (bn254_pairing.h: g2_line_table).
// One key per lane; a refused key walks on with the generator so that the wave stays convergent.  One-time work per key
// set (87 Fq2 inversions per key: ~15 ms for 256 keys), not part of any verify.
KERNEL void k_register_keys(const uint8_t* pks, size_t n_keys, uint32_t flags, int32_t* lines, uint8_t* key_st, uint8_t* key_inf, int32_t* key_xy) {
  const size_t j = (size_t)blockIdx.x * BN_WAVE + threadIdx.x;
  const bool live = j < n_keys;
  G2Affine q;
  uint8_t st = decode_g2(q, pks + 128 * (live ? j : n_keys - 1), flags);
  if (st != ST_OK || q.inf) g2_set_generator_keep_inf(q);
  const bool in = g2_in_subgroup(q);
  if (st == ST_OK && !q.inf && !in) { st = ST_INVALID_GROUP_POINT; g2_set_generator_keep_inf(q); }
  int32_t* out = lines + (live ? j : 0) * (size_t)BN_N_FIXED_LINES * BN_KEY_LINE_WORDS;
  const bool ok = g2_line_table(q, [&](int idx, const KeyLine& kl) {
    if (!live) return;
    const Fp c[4] = {fp_canon(kl.c0.c0), fp_canon(kl.c0.c1), fp_canon(kl.c1.c0), fp_canon(kl.c1.c1)};
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int k = 0; k < BN_LIMBS; ++k) out[((size_t)idx * 4 + e) * BN_LIMBS + k] = c[e].v[k];
  });
  if (!live) return;
  if (st == ST_OK && !q.inf && !ok) st = ST_INVALID_GROUP_POINT;   // a line with c2 = 0: not reachable from the order-r subgroup (~2^-250)
  key_st[j] = st;
  key_inf[j] = q.inf;
  // the point itself (x.re, x.im, y.re, y.im; 4 x 9 words): small keyed batches run the small-batch kernels on expanded keys
  const Fp xy[4] = {q.x.c0, q.x.c1, q.y.c0, q.y.c1};
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int k = 0; k < BN_LIMBS; ++k) key_xy[(j * 4 + e) * BN_LIMBS + k] = xy[e].v[k];
}
// keyed verify of a SMALL batch: the registered key of every tuple written into the Q planes (with the status rule of the keyed
// kernel: signature first, then index out of range, then the key's own), after which the batch is an ordinary verify
KERNEL_SMALL void k_keyed_expand(size_t n, Ws ws, const uint32_t* key_idx, KeyTable kt, const int32_t* key_xy) {
  const size_t i = (size_t)blockIdx.x * BN_WAVE + threadIdx.x;
  if (i >= n) return;
  uint32_t key = key_idx[i];
  uint8_t kst = ST_OK;
  if (key >= kt.n_keys) { kst = ST_INDEX_OOB; key = 0; }
  else kst = kt.st[key];
  const uint8_t prev = ws_byte(ws, BY_ST_DECODE, i);
  ws_byte(ws, BY_ST_DECODE, i) = prev != ST_OK ? prev : kst;
  G2Affine q;
  const int32_t* w = key_xy + (size_t)key * 4 * BN_LIMBS;
  q.x.c0 = fp_load_const(w); q.x.c1 = fp_load_const(w + BN_LIMBS); q.y.c0 = fp_load_const(w + 2 * BN_LIMBS); q.y.c1 = fp_load_const(w + 3 * BN_LIMBS);
  q.inf = kt.inf[key] != 0;
  if (kst != ST_OK) g2_set_generator(q);               // a refused key: the tuple's status is set, the arithmetic walks on with the generator
  ws_store_g2(ws, i, q);
}

// keyed verify against an EMPTY key set: every index is out of range — the signature's decode status first, else IndexOutOfBounds
KERNEL_SMALL void k_keyed_no_keys(size_t n, Ws ws, uint8_t* status_out) {
  const size_t i = (size_t)blockIdx.x * BN_WAVE + threadIdx.x;
  if (i >= n) return;
  const uint8_t st = ws_byte(ws, BY_ST_DECODE, i);
  status_out[i] = st != ST_OK ? st : (uint8_t)ST_INDEX_OOB;
}

// hash_to_try_and_increment (hash.rs:29-63) in ROUNDS.  The reference tries counters 0,1,2,... per
// message until one yields a point (p = 0.4726 per try, 2.12 tries on average, 20+ for the unluckiest
// message of a 65 536 batch).  One-message-per-lane with a retry loop makes every wave wait for its
// slowest lane and the kernel for the slowest message, and every failed try pays for a square-root
// exponentiation.  Instead:
//   * a try is first only TESTED: SHA-256, range rules, x^3 + 3, and its Jacobi symbol (binary algorithm,
//     no multiplications) — ~8 % of the cost of the exponentiation;
//   * a round tests only the messages that still have no counter (compacted index list), `width`
//     consecutive counters at once in `width` different lanes (speculation; width grows as the survivors
//     thin out so every round fills the SIMDs); atomicMin keeps the SMALLEST passing counter, exactly the
//     one the sequential loop stops at;
//   * k_hash_finish then computes ONE square root per message, for the winning counter.
// Lane w of a round: slot = w % n_act (message), j = w / n_act (counter offset) — consecutive lanes
// work on consecutive messages with the same offset.
// Message i of an offsets array: the bytes [off[i], off[i+1]) of a buffer of msgs_len bytes.  A pair that is reversed or
// runs past the buffer — only a *_device caller can hand one over: the host entry points validate their arrays, Rust slices
// cannot express one (/root/reference/src/ecdsa.rs:49) — is hashed as the EMPTY message, never dereferenced, and the item
// reports InvalidLength (5) in its hash status.  msgs_len = UINT64_MAX when the caller did not declare the buffer size
// (bn254_ctx_expect_msgs_len): then only reversed pairs can be caught.
__device__ __forceinline__ bool msg_span(const uint64_t* off, size_t i, uint64_t msgs_len, uint64_t& lo, uint64_t& len) {
  lo = off[i];
  const uint64_t hi = off[i + 1];
  const bool ok = lo <= hi && hi <= msgs_len;
  len = ok ? hi - lo : 0;
  if (!ok) lo = 0;
  return ok;
}
KERNEL_SMALL void k_hash_init(size_t n, Ws ws) {
  size_t i = (size_t)blockIdx.x * BN_WAVE + threadIdx.x;
  if (i < n) { ws.h_best[i] = HASH_NONE; ws.h_next[i] = 0; }
  if (i <= HASH_MAX_ROUNDS) ws.h_cnt[i] = (i == 0) ? (uint32_t)n : 0u;
}
KERNEL_SMALL void k_hash_round(const uint8_t* msgs, const uint64_t* off, uint64_t msgs_len, Ws ws, int round, uint32_t width, uint32_t max_ctr) {
  const uint32_t n_act = ws.h_cnt[round];
  if (n_act == 0) return;
  const uint32_t* list = round == 0 ? nullptr : ws.h_list + (size_t)(round & 1) * ws.stride;
  const size_t total = (size_t)n_act * width;
  for (size_t w = (size_t)blockIdx.x * BN_WAVE + threadIdx.x; w < total; w += (size_t)gridDim.x * BN_WAVE) {
    uint32_t slot = (uint32_t)(w % n_act), j = (uint32_t)(w / n_act);
    uint32_t i = list ? list[slot] : slot;
    uint32_t ctr = (uint32_t)ws.h_next[i] + j;
    if (ctr >= max_ctr) continue;                                  // hash.rs:40: counters 0..=254
    uint64_t lo, len;
    msg_span(off, i, msgs_len, lo, len);
    const uint8_t* msg = msgs + lo;
    HashState hs;
    hash_state_init(hs, msg, len);
    if (hash_try_filter(hs, msg, len, ctr)) atomicMin(&ws.h_best[i], ctr);
  }
}
// after a round: messages without a passing counter are queued for the next round (or give up at max_ctr).
// Queue positions: ONE atomic per 1024-slot tile.  An atomic per wave (what `atomicAdd` on a wave-uniform address compiles to anyway:
// the backend already reduces it across the wave) is 262 144 same-address atomics in the first round of configs[4], which the L2 channel
// that owns the counter serialises at ~10 ns each: 3 ms for that one launch, 6.3 ms per 16 Mi step, 99 % wait
// (profiles/r04_m_kernel_stats_hash.csv).  A 1024-lane workgroup counts its survivors through LDS, its first lane reserves the tile's
// range, every survivor takes base + the waves before it + its rank in its own wave.
// Rounds of up to HASH_RESOLVE_TILES_MIN slots keep one wave per 64 slots (k_hash_resolve: a few thousand atomics at most, and no
// workgroup barriers on the latency path of the headline's 65 536-message hash).
KERNEL_SMALL void k_hash_resolve(Ws ws, int round, uint32_t width, uint32_t max_ctr) {
  const uint32_t n_act = ws.h_cnt[round];
  if (n_act == 0) return;
  const uint32_t* list = round == 0 ? nullptr : ws.h_list + (size_t)(round & 1) * ws.stride;
  uint32_t* list_out = ws.h_list + (size_t)((round + 1) & 1) * ws.stride;
  const size_t span = (size_t)gridDim.x * BN_WAVE;
  for (size_t base = (size_t)blockIdx.x * BN_WAVE; base < n_act; base += span) {        // wave-uniform trip count: the vote needs every lane
    const size_t slot = base + threadIdx.x;
    bool survivor = false;
    uint32_t i = 0;
    if (slot < n_act) {
      i = list ? list[slot] : (uint32_t)slot;
      if (ws.h_best[i] == HASH_NONE) {
        const uint32_t next = (uint32_t)ws.h_next[i] + width;
        if (next < max_ctr) { ws.h_next[i] = (uint8_t)next; survivor = true; }           // else hash.rs:62: HashToPointError (k_hash_finish)
      }
    }
    const uint64_t votes = __ballot(survivor);
    if (votes == 0) continue;
    uint32_t first = 0;
    if (threadIdx.x == 0) first = atomicAdd(&ws.h_cnt[round + 1], (uint32_t)__popcll(votes));
    first = __shfl(first, 0, BN_WAVE);
    if (survivor) list_out[first + (uint32_t)__popcll(votes & ((1ull << threadIdx.x) - 1ull))] = i;
  }
}
#define HASH_RESOLVE_WG 1024
#define HASH_RESOLVE_TILES_MIN ((size_t)1 << 20)
__global__ void __launch_bounds__(HASH_RESOLVE_WG) k_hash_resolve_tiles(Ws ws, int round, uint32_t width, uint32_t max_ctr) {
  const uint32_t n_act = ws.h_cnt[round];
  if (n_act == 0) return;
  const uint32_t* list = round == 0 ? nullptr : ws.h_list + (size_t)(round & 1) * ws.stride;
  uint32_t* list_out = ws.h_list + (size_t)((round + 1) & 1) * ws.stride;
  __shared__ uint32_t wave_cnt[HASH_RESOLVE_WG / BN_WAVE];
  __shared__ uint32_t tile_base;
  const unsigned wave = threadIdx.x / BN_WAVE, lane = threadIdx.x % BN_WAVE;
  const size_t span = (size_t)gridDim.x * HASH_RESOLVE_WG;
  for (size_t base = (size_t)blockIdx.x * HASH_RESOLVE_WG; base < n_act; base += span) {    // workgroup-uniform trip count: barriers inside
    const size_t slot = base + threadIdx.x;
    bool survivor = false;
    uint32_t i = 0;
    if (slot < n_act) {
      i = list ? list[slot] : (uint32_t)slot;
      if (ws.h_best[i] == HASH_NONE) {
        const uint32_t next = (uint32_t)ws.h_next[i] + width;
        if (next < max_ctr) { ws.h_next[i] = (uint8_t)next; survivor = true; }           // else hash.rs:62: HashToPointError (k_hash_finish)
      }
    }
    const uint64_t votes = __ballot(survivor);
    if (lane == 0) wave_cnt[wave] = (uint32_t)__popcll(votes);
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t total = 0;
      for (unsigned w = 0; w < HASH_RESOLVE_WG / BN_WAVE; ++w) { const uint32_t c = wave_cnt[w]; wave_cnt[w] = total; total += c; }   // -> exclusive prefix
      tile_base = total ? atomicAdd(&ws.h_cnt[round + 1], total) : 0u;
    }
    __syncthreads();
    if (survivor) list_out[tile_base + wave_cnt[wave] + (uint32_t)__popcll(votes & ((1ull << lane) - 1ull))] = i;
    __syncthreads();                                                                      // wave_cnt / tile_base are rewritten by the next tile
  }
}
// SMALL batches (n <= HASH_DIRECT_MAX_N): latency, not work, is what counts — the first `width` counters of a message
// (a power of two <= 32, default 32) are tried in as many lanes of one wave with the square root itself (no filter pass
// first: SHA-256 + one exponentiation instead of SHA-256 + Jacobi symbol, then SHA-256 + exponentiation in a second
// kernel), the lowest passing counter writes its point.  A message without one (p = 0.5274^32 = 1.3e-9) is queued as a
// survivor of "round 0" for the ordinary rounds, which start at counter `width` (and cost it a second exponentiation).
KERNEL_SMALL void k_hash_direct(const uint8_t* msgs, const uint64_t* off, uint64_t msgs_len, size_t n, Ws ws, uint32_t width, uint32_t max_ctr, int px,
                                int inf_plane, uint8_t* tries_out) {
  const size_t w = (size_t)blockIdx.x * BN_WAVE + threadIdx.x;
  const size_t i = w / width;
  const uint32_t ctr = (uint32_t)(w % width);
  bool ok = false;
  G1Affine p;
  g1_set_generator(p);
  bool span_ok = true;
  if (i < n && ctr < max_ctr) {
    uint64_t lo, len;
    span_ok = msg_span(off, i, msgs_len, lo, len);
    const uint8_t* msg = msgs + lo;
    HashState hs;
    hash_state_init(hs, msg, len);
    ok = hash_try(p, hs, msg, len, ctr);
  }
  const uint64_t pass = __ballot(ok);
  if (i >= n) return;
  if (!span_ok) {                                                   // the whole counter group of the message agrees
    if (ctr == 0) {
      g1_set_generator(p);
      ws_store_g1(ws, px, inf_plane, i, p);
      ws_byte(ws, BY_ST_HASH, i) = (uint8_t)ST_INVALID_LENGTH;
      if (tries_out) tries_out[i] = 0;
      ws.h_best[i] = HASH_DONE;
    }
    return;
  }
  const uint32_t group = (uint32_t)(pass >> (threadIdx.x & ~(width - 1u))) & (uint32_t)((1ull << width) - 1u);
  if (group != 0) {
    if (ctr == (uint32_t)__builtin_ctz(group)) {                    // hash.rs:40-59: the first counter that yields a point
      ws_store_g1(ws, px, inf_plane, i, p);
      ws_byte(ws, BY_ST_HASH, i) = (uint8_t)ST_OK;
      if (tries_out) tries_out[i] = (uint8_t)(ctr + 1);
      ws.h_best[i] = HASH_DONE;
    }
  } else if (ctr == 0) {
    ws.h_best[i] = HASH_NONE;
    ws.h_next[i] = (uint8_t)width;
    if (width < max_ctr) {
      const uint32_t pos = atomicAdd(&ws.h_cnt[1], 1u);
      (ws.h_list + ws.stride)[pos] = (uint32_t)i;                   // the list that feeds round 1
    }
  }
}
// the point of every message: the even root for its winning counter (or the error status)
KERNEL_SMALL void k_hash_finish(const uint8_t* msgs, const uint64_t* off, uint64_t msgs_len, size_t n, Ws ws, uint32_t max_ctr, int px, int inf_plane,
                                uint8_t* tries_out) {
  size_t i = (size_t)blockIdx.x * BN_WAVE + threadIdx.x;
  if (i >= n) return;
  const uint32_t best = ws.h_best[i];
  if (best == HASH_DONE) return;                                   // k_hash_direct
  uint64_t lo, len;
  const bool span_ok = msg_span(off, i, msgs_len, lo, len);
  const uint8_t* msg = msgs + lo;
  HashState hs;
  hash_state_init(hs, msg, len);
  G1Affine p;
  // the filter and the exponentiation agree by construction (Euler's criterion); a disagreement would be
  // reported as an error status, never as a wrong point
  bool ok = span_ok && best != HASH_NONE && hash_try(p, hs, msg, len, best);
  if (!ok) g1_set_generator(p);
  ws_store_g1(ws, px, inf_plane, i, p);
  ws_byte(ws, BY_ST_HASH, i) = ok ? (uint8_t)ST_OK : !span_ok ? (uint8_t)ST_INVALID_LENGTH : (uint8_t)ST_HASH_TO_POINT;
  if (tries_out) tries_out[i] = ok ? (uint8_t)(best + 1) : !span_ok ? (uint8_t)0 : (uint8_t)max_ctr;
}

// ECDSA::verify Miller loop: f = miller(H(m), pk) * miller(sig, -G2)   (ecdsa.rs:53-57)
// P1 planes hold sig, P2 planes hold H(m), Q planes hold pk.
// With `map` (randomised batch verification, exact re-check of failed groups) lane j works on item
// map[j] for j < *count and leaves at once otherwise.
KERNEL void k_miller_verify(size_t n, Ws ws, const uint32_t* map, const uint32_t* count) {
  size_t i = (size_t)blockIdx.x * BN_WAVE + threadIdx.x;
  if (i >= n) return;
  if (map) { if (i >= *count) return; i = map[i]; }
  G1Affine sig, h;
  G2Affine pk;
  ws_load_g1(ws, PL_P1X, BY_P1_INF, i, sig);
  ws_load_g1(ws, PL_P2X, BY_P2_INF, i, h);
  ws_load_g2(ws, i, pk);
  __shared__ Fp12Slot lds_f[BN_WAVE];
  Fp12& f = lds_f[threadIdx.x].v;
  miller_loop<true, true>(f, h, pk, sig);
  ws_store_f12(ws, i, f);
}
// The same, one PAIRING per lane: the two Miller loops of a verify run in different waves so that a
// 65 536-verify batch puts two waves on every SIMD (two co-resident waves each keep the full
// single-wave issue rate on gfx950).  Workgroups [0, nblk) take pair A = (H(m), pk) with a variable
// twist point, workgroups [nblk, 2 nblk) take pair B = (sig, -G2::one()) through the line table;
// f_A lands at workspace index i, f_B at index f_stride + i; k_final_exp multiplies them.
__device__ __noinline__ void miller_role_a(size_t i, Ws ws) {
  G1Affine h, unused_g1;
  G2Affine pk;
  ws_load_g1(ws, PL_P2X, BY_P2_INF, i, h);
  ws_load_g2(ws, i, pk);
  g1_set_generator(unused_g1);
  Fp12 f;
  miller_loop<true, false>(f, h, pk, unused_g1);
  ws_store_f12(ws, i, f);
}
__device__ __noinline__ void miller_role_b(size_t i, size_t f_stride, Ws ws) {
  G1Affine sig, unused_g1;
  G2Affine unused_g2;
  ws_load_g1(ws, PL_P1X, BY_P1_INF, i, sig);
  g1_set_generator(unused_g1);
  g2_set_generator(unused_g2);
  Fp12 f;
  miller_loop<false, true>(f, unused_g1, unused_g2, sig);
  ws_store_f12(ws, f_stride + i, f);
  ws_byte(ws, BY_ST_DECODE, f_stride + i) = ST_OK;
}
KERNEL void k_miller_verify_split(size_t n, size_t f_stride, unsigned nblk, Ws ws) {
  bool role_b = blockIdx.x >= nblk;     // wave-uniform
  size_t i = (size_t)(blockIdx.x - (role_b ? nblk : 0u)) * BN_WAVE + threadIdx.x;
  if (i >= n) return;
  if (!role_b) miller_role_a(i, ws); else miller_role_b(i, f_stride, ws);
}
// generic single pair per lane: f = miller(P1, Q)
KERNEL void k_miller_var(size_t n, Ws ws) {
  size_t i = (size_t)blockIdx.x * BN_WAVE + threadIdx.x;
  if (i >= n) return;
  G1Affine p;
  G2Affine q;
  ws_load_g1(ws, PL_P1X, BY_P1_INF, i, p);
  ws_load_g2(ws, i, q);
  Fp12 f;
  miller_loop<true, false>(f, p, q, p);
  ws_store_f12(ws, i, f);
}
// check_public_keys Miller loop: miller(G1::one(), pk_g2) * miller(pk_g1, -G2)   (ecdsa.rs:80-86)
KERNEL void k_miller_cpk(size_t n, Ws ws) {
  size_t i = (size_t)blockIdx.x * BN_WAVE + threadIdx.x;
  if (i >= n) return;
  G1Affine pk1, g;
  G2Affine pk2;
  ws_load_g1(ws, PL_P1X, BY_P1_INF, i, pk1);
  ws_load_g2(ws, i, pk2);
  g1_set_generator(g);
  Fp12 f;
  miller_loop<true, true>(f, g, pk2, pk1);
  ws_store_f12(ws, i, f);
}

// item i: product of the k Miller values f[i*k .. i*k+k), final exponentiation, compare with one.
// status = first decode error among its pairs, else hash error (if use_hash), else 0 / 9.
KERNEL void k_final_exp(size_t n, size_t k, size_t item_stride, size_t pair_stride, Ws ws, int use_hash, uint8_t* gt_out, uint8_t* status_out,
                        int raw_only, size_t base, const uint32_t* map, const uint32_t* count) {
  size_t i = (size_t)blockIdx.x * BN_WAVE + threadIdx.x;
  if (i >= n) return;
  if (map) { if (i >= *count) return; i = map[i]; }   // see k_miller_verify
  // `base`: the factors of item i start at workspace index base + i*item_stride (the per-group values of the
  // randomised batch verification live behind the per-item region); outputs are indexed by i
  // factor j of item i sits at workspace index i*item_stride + j*pair_stride:
  //   pairing API  (k adjacent pairs per item): item_stride = k, pair_stride = 1
  //   split verify (f_A at i, f_B at half + i):  item_stride = 1, pair_stride = half
  // every factor slot carries its own decode status (slots of the B half hold 0)
  Fp12 f, g;
  ws_load_f12(ws, base + i * item_stride, f);
  uint8_t st = ws_byte(ws, BY_ST_DECODE, base + i * item_stride);
  for (size_t j = 1; j < k; ++j) {
    size_t idx = base + i * item_stride + j * pair_stride;
    ws_load_f12(ws, idx, g);
    fp12_mul(f, f, g);
    uint8_t sj = ws_byte(ws, BY_ST_DECODE, idx);
    if (st == ST_OK) st = sj;
  }
  if (st == ST_OK && use_hash) st = ws_byte(ws, BY_ST_HASH, i);
  __shared__ Fp12Slot lds_acc[BN_WAVE];
  if (!raw_only) {
    if (gt_out) final_exponentiation(f, f, lds_acc[threadIdx.x].v);        // canonical Gt: exact exponent
    else final_exponentiation_check(f, f, lds_acc[threadIdx.x].v);          // == one test only: shorter chain
  }
  if (gt_out) encode_fp12(gt_out + 384 * i, f);
  if (status_out) status_out[i] = st != ST_OK ? st : (fp12_is_one(f) ? (uint8_t)ST_OK : (uint8_t)ST_VERIFICATION_FAILED);
}

// ------------------------------------------------------------------------------------------
// Randomised batch verification: groups of 64 items = one wave.
//   group passes  <=>  prod_i e(r_i H(m_i), pk_i) * e(sum_i r_i sig_i, -G2) == 1   over its valid items
// N + N/64 Miller loops and N/64 final exponentiations instead of 2N and N.
//   k_rand_scale  : A_i = r_i H(m_i) (affine, HASH planes), S_g = sum_i r_i sig_i (wave reduction in LDS)
//   k_miller_rand : f_i = miller(A_i, pk_i), F_g = prod_i f_i (wave reduction in LDS)
//   k_rand_tail   : F_g * miller(S_g, -G2)  ->  k_final_exp  ->  one byte per group
//   k_rand_collect: statuses of passing groups; items of failing groups are queued for the exact kernels
// Per-group values live at workspace index gbase + g, behind the per-item region.
// ------------------------------------------------------------------------------------------
struct Seed { uint32_t w[8]; };
struct G1JacSlot { G1Jac v; int32_t pad; };   // 31 words: odd stride, no LDS bank conflicts

// mode: 0 = 128-bit scalar, 1 = 64-bit scalar, 2 = k1 + k2*lambda with 64-bit k1, k2 (BN254_FLAG_RAND_GLV)
KERNEL_SMALL void k_rand_scale(size_t n, Ws ws, Seed seed, int mode, size_t gbase) {
  const unsigned t = threadIdx.x;
  size_t i = (size_t)blockIdx.x * BN_WAVE + t;
  const bool live = i < n;                       // no early return: every lane reaches the barriers
  const size_t ii = live ? i : n - 1;
  if (blockIdx.x == 0 && t == 0) ws.h_cnt[0] = 0;   // queue length of k_rand_collect (hash rounds are done)
  G1Affine sig, h;
  ws_load_g1(ws, PL_P1X, BY_P1_INF, ii, sig);
  ws_load_g1(ws, PL_P2X, BY_P2_INF, ii, h);
  const bool valid = live && ws_byte(ws, BY_ST_DECODE, ii) == ST_OK && ws_byte(ws, BY_ST_HASH, ii) == ST_OK;
  uint32_t k[4];
  rand_scalar(k, seed.w, (uint64_t)ii, mode == 1);
  __shared__ G1JacSlot lds_s[BN_WAVE];                 // the accumulator of both scalar multiplications (in place, see k_krand_scale)
  G1Jac& sj = lds_s[t].v;
  G1Jac id;
  if (mode == 2) g1_mul_glv(sj, h, k, k + 2); else if (mode == 1) jac_mul_u64(sj, h, k); else jac_mul_u128(sj, h, k);   // wave-uniform
  G1Affine aa;
  jac_to_affine(aa, sj);
  aa.inf = aa.inf || !valid;
  if (live) ws_store_g1(ws, PL_HASHX, BY_A_INF, i, aa);
  if (mode == 2) g1_mul_glv(sj, sig, k, k + 2); else if (mode == 1) jac_mul_u64(sj, sig, k); else jac_mul_u128(sj, sig, k);
  jac_set_identity(id);
  jac_select(sj, !valid, id, sj);
  __syncthreads();
  for (unsigned stride = BN_WAVE / 2; stride >= 1; stride >>= 1) {
    if (t < stride) jac_add(lds_s[t].v, lds_s[t].v, lds_s[t + stride].v);
    __syncthreads();
  }
  if (t == 0) {
    G1Affine sa;
    jac_to_affine(sa, lds_s[0].v);
    ws_store_g1(ws, PL_P1X, BY_P1_INF, gbase + blockIdx.x, sa);
    ws_byte(ws, BY_ST_DECODE, gbase + blockIdx.x) = ST_OK;
  }
}
KERNEL void k_miller_rand(size_t n, Ws ws, size_t gbase) {
  const unsigned t = threadIdx.x;
  size_t i = (size_t)blockIdx.x * BN_WAVE + t;
  const bool live = i < n;
  const size_t ii = live ? i : n - 1;
  G1Affine a;
  G2Affine pk;
  ws_load_g1(ws, PL_HASHX, BY_A_INF, ii, a);
  if (!live) a.inf = true;
  ws_load_g2(ws, ii, pk);
  __shared__ Fp12Slot lds_f[BN_WAVE];
  Fp12& f = lds_f[t].v;
  miller_loop<true, false>(f, a, pk, a);
  __syncthreads();
  for (unsigned stride = BN_WAVE / 2; stride >= 1; stride >>= 1) {
    if (t < stride) fp12_mul(f, f, lds_f[t + stride].v);
    __syncthreads();
  }
  if (t == 0) ws_store_f12(ws, gbase + blockIdx.x, f);
}
// The same with TWO items per lane sharing f (one f^2 per loop step for both, merged line products): lanes
// [32h, 32h+32) of block b hold group 2b+h, lane t of a half the items 2t and 2t+1 of its group.  Used when the
// batch still fills the device at two items per lane.
KERNEL void k_miller_rand2(size_t n, size_t n_groups, Ws ws, size_t gbase) {
  const unsigned t = threadIdx.x, th = t & 31u;
  const size_t group = (size_t)blockIdx.x * 2 + (t >> 5);
  const size_t i0 = group * BN_WAVE + 2 * th, i1 = i0 + 1;
  G1Affine a0, a1;
  G2Affine pk0, pk1;
  const size_t j0 = i0 < n ? i0 : n - 1, j1 = i1 < n ? i1 : n - 1;
  ws_load_g1(ws, PL_HASHX, BY_A_INF, j0, a0);
  ws_load_g1(ws, PL_HASHX, BY_A_INF, j1, a1);
  if (i0 >= n) a0.inf = true;
  if (i1 >= n) a1.inf = true;
  ws_load_g2(ws, j0, pk0);
  ws_load_g2(ws, j1, pk1);
  __shared__ Fp12Slot lds_f[BN_WAVE];
  Fp12& f = lds_f[t].v;
  miller_loop_2var(f, a0, pk0, a1, pk1);
  __syncthreads();
  for (unsigned stride = 16; stride >= 1; stride >>= 1) {
    if (th < stride) fp12_mul(f, f, lds_f[t + stride].v);
    __syncthreads();
  }
  if (th == 0 && group < n_groups) ws_store_f12(ws, gbase + group, f);
}
KERNEL void k_rand_tail(size_t n_groups, Ws ws, size_t gbase) {
  size_t g = (size_t)blockIdx.x * BN_WAVE + threadIdx.x;
  if (g >= n_groups) return;
  G1Affine s, unused_g1;
  G2Affine unused_g2;
  ws_load_g1(ws, PL_P1X, BY_P1_INF, gbase + g, s);
  g1_set_generator(unused_g1);
  g2_set_generator(unused_g2);
  Fp12 fg;
  ws_load_f12(ws, gbase + g, fg);
  __shared__ Fp12Slot lds_f[BN_WAVE];
  Fp12& f = lds_f[threadIdx.x].v;
  miller_loop<false, true>(f, unused_g1, unused_g2, s);
  fp12_mul(f, f, fg);
  ws_store_f12(ws, gbase + g, f);
}
// group verdicts for batches that were verified exactly: 1 iff no item of the group failed the pairing check
KERNEL_SMALL void k_group_ok_from_status(size_t n_groups, size_t n, const uint8_t* status, uint8_t* group_ok_out) {
  size_t g = (size_t)blockIdx.x * BN_WAVE + threadIdx.x;
  if (g >= n_groups) return;
  uint8_t ok = 1;
  for (size_t i = g * BN_WAVE; i < (g + 1) * BN_WAVE && i < n; ++i) if (status[i] == ST_VERIFICATION_FAILED) ok = 0;
  group_ok_out[g] = ok;
}
KERNEL_SMALL void k_rand_collect(size_t n, Ws ws, const uint8_t* group_st, uint8_t* status_out, uint8_t* group_ok_out) {
  size_t i = (size_t)blockIdx.x * BN_WAVE + threadIdx.x;
  if (i >= n) return;
  uint8_t st = ws_byte(ws, BY_ST_DECODE, i);
  if (st == ST_OK) st = ws_byte(ws, BY_ST_HASH, i);
  const bool ok = group_st[i / BN_WAVE] == ST_OK;
  if (ok || st != ST_OK) {
    status_out[i] = st;
  } else {
    uint32_t pos = atomicAdd(&ws.h_cnt[0], 1u);
    ws.h_list[pos] = (uint32_t)i;
  }
  if (group_ok_out && threadIdx.x == 0) group_ok_out[i / BN_WAVE] = ok ? 1 : 0;
}

// ------------------------------------------------------------------------------------------
// Keyed randomised batch verification (opt-in like section 4c; for REGISTERED keys): items that share a key share the G2
// argument, so a whole group of them is ONE pairing product
//     e(sum_i r_i H(m_i), pk) * e(sum_i r_i sig_i, -G2) == 1
// — two table-driven Miller loops and one final exponentiation per 64 items, and per item only the two 128-bit scalar
// multiplications.  Items are grouped by key on the device (counting sort: k_krand_prepare / scan / scatter), every key's
// run padded to whole groups of 64; a group is a "virtual tuple" (H := sum r_i H(m_i), sig := sum r_i sig_i, key) at workspace
// index gbase + g and goes through the kernels of the exact keyed verify; the items of a failing group are re-checked exactly.
//   meta[0] = number of groups, meta[1] = number of slots of `perm` in use (both known on the device only)
// ------------------------------------------------------------------------------------------
#define KRAND_NONE 0xFFFFFFFFu
KERNEL_SMALL void k_krand_prepare(size_t n, Ws ws, const uint32_t* key_idx, KeyTable kt, uint32_t* cnt, uint8_t* status_out) {
  const size_t i = (size_t)blockIdx.x * BN_WAVE + threadIdx.x;
  if (i == 0) ws.h_cnt[0] = 0;                          // queue of the exact re-check (the hash rounds are done with it)
  if (i >= n) return;
  uint8_t st = ws_byte(ws, BY_ST_DECODE, i);
  const uint32_t key = key_idx[i];
  if (st == ST_OK) st = key >= kt.n_keys ? (uint8_t)ST_INDEX_OOB : kt.st[key];
  if (st == ST_OK) st = ws_byte(ws, BY_ST_HASH, i);
  ws_byte(ws, BY_ST_DECODE, i) = st;                    // the item's final status unless the pairing check has the last word
  if (st != ST_OK) status_out[i] = st;
  else atomicAdd(&cnt[key], 1u);
}
// one wave: start[k] = first slot of key k (runs padded to multiples of 64), gkey[g] = key of group g, cnt reset (the scatter's cursors)
KERNEL_SMALL void k_krand_scan(uint32_t n_keys, uint32_t* cnt, uint32_t* start, uint32_t* gkey, uint32_t* meta) {
  const unsigned t = threadIdx.x;
  uint32_t groups_before = 0;
  for (uint32_t base = 0; base < n_keys; base += BN_WAVE) {
    const uint32_t k = base + t;
    const uint32_t ng = k < n_keys ? (cnt[k] + BN_WAVE - 1) / BN_WAVE : 0u;
    uint32_t incl = ng;
    for (int off = 1; off < BN_WAVE; off <<= 1) {
      const uint32_t up = __shfl_up(incl, off, BN_WAVE);
      if ((int)t >= off) incl += up;
    }
    const uint32_t first = groups_before + incl - ng;
    if (k < n_keys) {
      start[k] = first * BN_WAVE;
      cnt[k] = 0;
      for (uint32_t j = 0; j < ng; ++j) gkey[first + j] = k;
    }
    groups_before += __shfl(incl, BN_WAVE - 1, BN_WAVE);
  }
  if (t == 0) { meta[0] = groups_before; meta[1] = groups_before * BN_WAVE; }
}
KERNEL_SMALL void k_krand_scatter(size_t n, Ws ws, const uint32_t* key_idx, const uint32_t* start, uint32_t* cursor, uint32_t* perm) {
  const size_t i = (size_t)blockIdx.x * BN_WAVE + threadIdx.x;
  if (i >= n || ws_byte(ws, BY_ST_DECODE, i) != ST_OK) return;
  const uint32_t key = key_idx[i];
  perm[start[key] + atomicAdd(&cursor[key], 1u)] = (uint32_t)i;
}
// group g = one wave: r_i H(m_i) and r_i sig_i of its items, both summed over the wave (LDS trees), as the tuple gbase + g
KERNEL_SMALL void k_krand_scale(const uint32_t* perm, const uint32_t* meta, Ws ws, Seed seed, int mode, size_t gbase) {
  const unsigned t = threadIdx.x;
  const size_t g = blockIdx.x;
  if (g >= meta[0]) return;                              // the whole block together
  const uint32_t item = perm[g * BN_WAVE + t];
  const bool valid = item != KRAND_NONE;
  const size_t ii = valid ? item : 0;
  G1Affine sig, h;
  ws_load_g1(ws, PL_P1X, BY_P1_INF, ii, sig);
  ws_load_g1(ws, PL_P2X, BY_P2_INF, ii, h);
  uint32_t k[4];
  rand_scalar(k, seed.w, (uint64_t)ii, mode == 1);
  // both products are accumulated IN their LDS slots (jac_mul_window works in place through the reference): the 4 doublings + 1
  // addition of every window stay out of the private segment
  __shared__ G1JacSlot lds_a[BN_WAVE], lds_s[BN_WAVE];
  G1Jac &a = lds_a[t].v, &sj = lds_s[t].v;
  G1Jac id;
  if (mode == 2) g1_mul_glv(a, h, k, k + 2); else if (mode == 1) jac_mul_u64(a, h, k); else jac_mul_u128(a, h, k);   // wave-uniform
  if (mode == 2) g1_mul_glv(sj, sig, k, k + 2); else if (mode == 1) jac_mul_u64(sj, sig, k); else jac_mul_u128(sj, sig, k);
  jac_set_identity(id);
  jac_select(a, !valid, id, a);
  jac_select(sj, !valid, id, sj);
  __syncthreads();
  for (unsigned stride = BN_WAVE / 2; stride >= 1; stride >>= 1) {
    if (t < stride) { jac_add(lds_a[t].v, lds_a[t].v, lds_a[t + stride].v); jac_add(lds_s[t].v, lds_s[t].v, lds_s[t + stride].v); }
    __syncthreads();
  }
  if (t == 0) {
    G1Affine aa, sa;
    jac_to_affine(aa, lds_a[0].v);
    jac_to_affine(sa, lds_s[0].v);
    ws_store_g1(ws, PL_P2X, BY_P2_INF, gbase + g, aa);
    ws_store_g1(ws, PL_P1X, BY_P1_INF, gbase + g, sa);
    ws_byte(ws, BY_ST_DECODE, gbase + g) = ST_OK;
    ws_byte(ws, BY_ST_HASH, gbase + g) = ST_OK;
  }
}
KERNEL_SMALL void k_krand_collect(size_t n_slots_max, const uint32_t* perm, const uint32_t* meta, const uint8_t* group_st, uint8_t* status_out, Ws ws) {
  const size_t j = (size_t)blockIdx.x * BN_WAVE + threadIdx.x;
  if (j >= n_slots_max || j >= meta[1]) return;
  const uint32_t item = perm[j];
  if (item == KRAND_NONE) return;
  if (group_st[j / BN_WAVE] == ST_OK) status_out[item] = ST_OK;
  else ws.h_list[atomicAdd(&ws.h_cnt[0], 1u)] = item;
}

// out[i] = a[i] + b[i]
KERNEL void k_g1_add(const uint8_t* a, const uint8_t* b, size_t n, uint8_t* out, uint8_t* status) {
  size_t i = (size_t)blockIdx.x * BN_WAVE + threadIdx.x;
  if (i >= n) return;
  G1Affine pa, pb, r;
  uint8_t st = decode_g1(pa, a + 64 * i, 0);
  uint8_t sb = decode_g1(pb, b + 64 * i, 0);
  if (st == ST_OK) st = sb;
  if (st != ST_OK) { g1_set_generator(pa); g1_set_generator(pb); }
  G1Jac ja, jb, jo;
  jac_from_affine(ja, pa); jac_from_affine(jb, pb);
  jac_add(jo, ja, jb);
  jac_to_affine(r, jo);
  if (st != ST_OK) r.inf = true;
  encode_g1(out + 64 * i, r);
  status[i] = st;
}
KERNEL void k_g2_add(const uint8_t* a, const uint8_t* b, size_t n, uint8_t* out, uint8_t* status) {
  size_t i = (size_t)blockIdx.x * BN_WAVE + threadIdx.x;
  if (i >= n) return;
  G2Affine pa, pb, r;
  uint8_t st = decode_g2(pa, a + 128 * i, 0);
  uint8_t sb = decode_g2(pb, b + 128 * i, 0);
  if (st == ST_OK) st = sb;
  if (st != ST_OK) { g2_set_generator(pa); g2_set_generator(pb); }
  G2Jac ja, jb, jo;
  jac_from_affine(ja, pa); jac_from_affine(jb, pb);
  jac_add(jo, ja, jb);
  jac_to_affine(r, jo);
  if (st != ST_OK) r.inf = true;
  encode_g2(out + 128 * i, r);
  status[i] = st;
}
// out[i] = scalar[i] * p[i]; p == nullptr: the point comes from the P1 planes (ECDSA::sign: H(m))
KERNEL void k_g1_mul(const uint8_t* p, const uint8_t* scalars, size_t n, int reduce, Ws ws, uint8_t* out, uint8_t* status) {
  size_t i = (size_t)blockIdx.x * BN_WAVE + threadIdx.x;
  if (i >= n) return;
  G1Affine pa, r;
  uint8_t st;
  if (p) {
    st = decode_g1(pa, p + 64 * i, 0);
  } else {
    ws_load_g1(ws, PL_P1X, BY_P1_INF, i, pa);
    st = ws_byte(ws, BY_ST_HASH, i);
  }
  if (st != ST_OK) g1_set_generator(pa);
  uint32_t k[8];
  scalar_from_be(k, scalars + 32 * i, reduce != 0);
  G1Jac jo;
  jac_mul(jo, pa, k);
  jac_to_affine(r, jo);
  if (st != ST_OK) r.inf = true;
  encode_g1(out + 64 * i, r);
  status[i] = st;
}
// p == nullptr: multiply the G2 generator (PublicKey::from_private_key)
KERNEL void k_g2_mul(const uint8_t* p, const uint8_t* scalars, size_t n, int reduce, uint8_t* out, uint8_t* status) {
  size_t i = (size_t)blockIdx.x * BN_WAVE + threadIdx.x;
  if (i >= n) return;
  G2Affine pa, r;
  uint8_t st = ST_OK;
  if (p) st = decode_g2(pa, p + 128 * i, 0); else g2_set_generator(pa);
  if (st != ST_OK) g2_set_generator(pa);
  uint32_t k[8];
  scalar_from_be(k, scalars + 32 * i, reduce != 0);
  G2Jac jo;
  jac_mul(jo, pa, k);
  jac_to_affine(r, jo);
  if (st != ST_OK) r.inf = true;
  encode_g2(out + 128 * i, r);
  status[i] = st;
}
// segmented sums (aggregation): out[i] = sum points[seg[i] .. seg[i+1])
KERNEL void k_g1_sum(const uint8_t* pts, const uint64_t* seg, size_t n, uint8_t* out, uint8_t* status) {
  size_t i = (size_t)blockIdx.x * BN_WAVE + threadIdx.x;
  if (i >= n) return;
  G1Jac acc;
  jac_set_identity(acc);
  uint8_t st = ST_OK;
  for (uint64_t j = seg[i]; j < seg[i + 1]; ++j) {
    G1Affine p;
    uint8_t s = decode_g1(p, pts + 64 * j, 0);
    if (s != ST_OK) { if (st == ST_OK) st = s; continue; }
    jac_accumulate(acc, p);
  }
  G1Affine r;
  jac_to_affine(r, acc);
  if (st != ST_OK) r.inf = true;
  encode_g1(out + 64 * i, r);
  status[i] = st;
}
KERNEL void k_g2_sum(const uint8_t* pts, const uint64_t* seg, size_t n, uint8_t* out, uint8_t* status) {
  size_t i = (size_t)blockIdx.x * BN_WAVE + threadIdx.x;
  if (i >= n) return;
  G2Jac acc;
  jac_set_identity(acc);
  uint8_t st = ST_OK;
  for (uint64_t j = seg[i]; j < seg[i + 1]; ++j) {
    G2Affine p;
    uint8_t s = decode_g2(p, pts + 128 * j, 0);
    if (s != ST_OK) { if (st == ST_OK) st = s; continue; }
    jac_accumulate(acc, p);
  }
  G2Affine r;
  jac_to_affine(r, acc);
  if (st != ST_OK) r.inf = true;
  encode_g2(out + 128 * i, r);
  status[i] = st;
}
// compressed -> uncompressed (Signature/PublicKeyG1::from_compressed, PublicKey::from_compressed)
KERNEL_SMALL void k_g1_decompress(const uint8_t* in, size_t n, uint8_t* out, uint8_t* status) {
  size_t i = (size_t)blockIdx.x * BN_WAVE + threadIdx.x;
  if (i >= n) return;
  G1Affine p;
  uint8_t st = decompress_g1(p, in + 33 * i);
  if (st != ST_OK) p.inf = true;
  encode_g1(out + 64 * i, p);
  status[i] = st;
}
KERNEL void k_g2_decompress(const uint8_t* in, size_t n, uint8_t* out, uint8_t* status) {
  size_t i = (size_t)blockIdx.x * BN_WAVE + threadIdx.x;
  if (i >= n) return;
  G2Affine p;
  uint8_t st = decompress_g2(p, in + 65 * i);
  if (st != ST_OK) g2_set_generator(p);
  bool in_sub = g2_in_subgroup(p);                 // wave-uniform ladder; AffineG2::new inside from_compressed
  if (st == ST_OK && !in_sub) st = ST_NOT_MEMBER;
  if (st != ST_OK) p.inf = true;
  encode_g2(out + 128 * i, p);
  status[i] = st;
}
// the same into the workspace planes of a verify: compressed signatures (33 B) -> P1 planes, compressed public keys
// (65 B, subgroup-checked as G2::from_compressed does) -> Q planes; status as in k_decode_g1 / k_decode_g2
KERNEL_SMALL void k_decompress_g1_ws(const uint8_t* in, size_t n, Ws ws) {
  size_t i = (size_t)blockIdx.x * BN_WAVE + threadIdx.x;
  if (i >= n) return;
  G1Affine p;
  uint8_t st = decompress_g1(p, in + 33 * i);
  if (st != ST_OK) g1_set_generator(p);
  ws_store_g1(ws, PL_P1X, BY_P1_INF, i, p);
  ws_byte(ws, BY_ST_DECODE, i) = st;
}
KERNEL void k_decompress_g2_ws(const uint8_t* in, size_t n, Ws ws) {
  size_t i = (size_t)blockIdx.x * BN_WAVE + threadIdx.x;
  if (i >= n) return;
  G2Affine q;
  uint8_t st = decompress_g2(q, in + 65 * i);
  if (st != ST_OK) g2_set_generator(q);
  bool in_sub = g2_in_subgroup(q);
  if (st == ST_OK && !in_sub) { st = ST_NOT_MEMBER; g2_set_generator(q); }
  ws_store_g2(ws, i, q);
  uint8_t prev = ws_byte(ws, BY_ST_DECODE, i);
  ws_byte(ws, BY_ST_DECODE, i) = prev != ST_OK ? prev : st;
}
// ---- aggregate verify (config 3): shared pools, per-tuple signer subsets -------------------------
KERNEL_SMALL void k_pool_decode_g1(const uint8_t* pts, size_t n, uint32_t flags, Pool pool) {
  size_t i = (size_t)blockIdx.x * BN_WAVE + threadIdx.x;
  if (i >= n) return;
  G1Affine p;
  uint8_t st = decode_g1(p, pts + 64 * i, flags);
  if (st != ST_OK) g1_set_generator(p);
  pool_store_fp(pool, 0, i, p.x); pool_store_fp(pool, 1, i, p.y);
  pool.st[i] = st | (p.inf ? 0x80 : 0);
}
KERNEL void k_pool_decode_g2(const uint8_t* pts, size_t n, uint32_t flags, Pool pool) {
  size_t i = (size_t)blockIdx.x * BN_WAVE + threadIdx.x;
  if (i >= n) return;
  G2Affine q;
  uint8_t st = decode_g2(q, pts + 128 * i, flags);
  if (st != ST_OK) g2_set_generator(q);
  if (flags & FLAG_G2_SUBGROUP_CHECK) {
    bool in = g2_in_subgroup(q);
    if (st == ST_OK && !in) { st = ST_INVALID_GROUP_POINT; g2_set_generator(q); }
  }
  pool_store_fp(pool, 0, i, q.x.c0); pool_store_fp(pool, 1, i, q.x.c1);
  pool_store_fp(pool, 2, i, q.y.c0); pool_store_fp(pool, 3, i, q.y.c1);
  pool.st[i] = st | (q.inf ? 0x80 : 0);
}
// Subset sums of the public-key pool ("four Russians"): every tuple of an aggregate verify adds up a subset of the SAME n_signers
// keys, so the sums of all 255 non-empty subsets of every group of 8 consecutive keys are tabulated once per call (n_signers / 8
// x 256 affine points, 4.7 MB for 1024 signers; ~4 additions + one inversion per entry) and a tuple adds ONE table entry per
// group — 128 additions instead of the ~512 of a dense list (k_aggregate_pair).  Entry j = group * 256 + mask; a pool entry
// that failed to decode counts as the identity here (the tuples that name it carry its status anyway).
KERNEL void k_pool_subsets_g2(Pool pk_pool, size_t n_signers, size_t n_groups, Pool sub) {
  const size_t j = (size_t)blockIdx.x * BN_WAVE + threadIdx.x;
  const bool live = j < n_groups * 256;
  const size_t g = (live ? j : 0) >> 8;
  const unsigned mask = (unsigned)(j & 255u);
  G2Jac acc;
  jac_set_identity(acc);
  for (int b = 0; b < 8; ++b) {                      // wave-uniform: jac_accumulate votes across the wave
    const size_t sgn = g * 8 + b;
    const size_t ss = sgn < n_signers ? sgn : 0;
    const uint8_t st = pk_pool.st[ss];
    G2Affine p;
    p.x.c0 = pool_load_fp(pk_pool, 0, ss); p.x.c1 = pool_load_fp(pk_pool, 1, ss);
    p.y.c0 = pool_load_fp(pk_pool, 2, ss); p.y.c1 = pool_load_fp(pk_pool, 3, ss);
    p.inf = !live || !((mask >> b) & 1u) || sgn >= n_signers || st != 0;       // st: 0x80 = identity entry, low bits = decode error
    jac_accumulate(acc, p);
  }
  G2Affine a;
  jac_to_affine(a, acc);
  if (!live) return;
  pool_store_fp(sub, 0, j, a.x.c0); pool_store_fp(sub, 1, j, a.x.c1);
  pool_store_fp(sub, 2, j, a.y.c0); pool_store_fp(sub, 3, j, a.y.c1);
  sub.st[j] = a.inf ? 0x80 : 0;
}
// The same for the signatures, per message: the sums of the 15 non-empty subsets of every group of 4 consecutive signers of
// message m (entry j = (m * groups4 + group) * 16 + mask; 302 MB for 1024 x 1024 — HBM is what this machine has), so that a
// tuple adds one table entry per group of 4 signers (256 instead of ~512 additions; 4 bits, not 8: an entry costs two additions
// and an inversion to build and is used by ~n / n_msgs tuples only).
KERNEL_SMALL void k_pool_subsets_g1(Pool sig_pool, size_t n_signers, size_t groups4, size_t n_msgs, Pool sub) {
  const size_t j = (size_t)blockIdx.x * BN_WAVE + threadIdx.x;
  const bool live = j < n_msgs * groups4 * 16;
  const size_t jj = live ? j : 0;
  const unsigned mask = (unsigned)(jj & 15u);
  const size_t g = (jj >> 4) % groups4, m = (jj >> 4) / groups4;
  G1Jac acc;
  jac_set_identity(acc);
  for (int b = 0; b < 4; ++b) {                      // wave-uniform
    const size_t sgn = g * 4 + b;
    const size_t sj = m * n_signers + (sgn < n_signers ? sgn : 0);
    const uint8_t st = sig_pool.st[sj];
    G1Affine p;
    p.x = pool_load_fp(sig_pool, 0, sj); p.y = pool_load_fp(sig_pool, 1, sj);
    p.inf = !live || !((mask >> b) & 1u) || sgn >= n_signers || st != 0;
    jac_accumulate(acc, p);
  }
  G1Affine a;
  jac_to_affine(a, acc);
  if (!live) return;
  pool_store_fp(sub, 0, j, a.x); pool_store_fp(sub, 1, j, a.y);
  sub.st[j] = a.inf ? 0x80 : 0;
}
// ---- WIDER subset tables for the largest aggregate batches (BN254_OPT_AGG_WIDE_MIN_TUPLES) ---------------------------------------------
// k_aggregate_pair adds one table entry per window of signers; twice the window is half the additions.  From a table of windows of w
// signers one of 2w signers is its "outer sum": T2w[hi * 2^w + lo] = Tw[group 2k][lo] + Tw[group 2k + 1][hi] — ONE affine addition per
// entry, 2^(2w) entries per doubled group: keys 8 -> 16 signers per entry (n_signers / 16 x 65 536 entries, 671 MB for 1 024 signers:
// HBM is what this machine has), signatures per message 4 -> 8.  An affine addition needs 1 / (x_B - x_A); a lane owns one `hi` and
// walks its `lo` values in batches of 8 whose denominators share ONE inversion (Montgomery's trick: prefix products up, the inverse
// peeled off on the way down; the B points are re-read from the source table, an L2 hit, instead of being kept in registers): 2
// products + 1 square for the chord, 3 products for the trick, an eighth of an inversion — ~16 products per entry where accumulate +
// jac_to_affine costs ~100.  Entries with an identity operand are copies; the rare lo with x_B = x_A (B = +-A: a pool that holds a
// point twice, or a point and its negative) takes the complete Jacobian formula and an inversion of its own.
__device__ __forceinline__ void pool_load_aff(const Pool& p, size_t j, G1Affine& q) { q.x = pool_load_fp(p, 0, j); q.y = pool_load_fp(p, 1, j); q.inf = (p.st[j] & 0x80) != 0; }
__device__ __forceinline__ void pool_load_aff(const Pool& p, size_t j, G2Affine& q) {
  q.x.c0 = pool_load_fp(p, 0, j); q.x.c1 = pool_load_fp(p, 1, j); q.y.c0 = pool_load_fp(p, 2, j); q.y.c1 = pool_load_fp(p, 3, j);
  q.inf = (p.st[j] & 0x80) != 0;
}
__device__ __forceinline__ void pool_store_aff(const Pool& p, size_t j, const G1Affine& q) { pool_store_fp(p, 0, j, q.x); pool_store_fp(p, 1, j, q.y); p.st[j] = q.inf ? 0x80 : 0; }
__device__ __forceinline__ void pool_store_aff(const Pool& p, size_t j, const G2Affine& q) {
  pool_store_fp(p, 0, j, q.x.c0); pool_store_fp(p, 1, j, q.x.c1); pool_store_fp(p, 2, j, q.y.c0); pool_store_fp(p, 3, j, q.y.c1);
  p.st[j] = q.inf ? 0x80 : 0;
}
// one lane: dst[dst0 + lo] = src[b0 + lo] + A for NLO consecutive lo (an entry of src may be the identity: the empty subset, or a sum that
// cancelled)
template <class F> __device__ __forceinline__ void aff_select(Affine<F>& r, bool c, const Affine<F>& a, const Affine<F>& b) {
  r.x = f_select(c, a.x, b.x); r.y = f_select(c, a.y, b.y); r.inf = c ? a.inf : b.inf;
}
template <class F, int NLO> __device__ __forceinline__ void pool_widen_lane(bool live, const Pool& src, size_t b0, Affine<F> A, const Pool& dst, size_t dst0) {
  constexpr int BATCH = 8;
  static_assert(NLO % BATCH == 0, "whole batches");
  for (int base = 0; base < NLO; base += BATCH) {
    F d[BATCH], pre[BATCH];
    bool exc[BATCH];
#pragma unroll
    for (int i = 0; i < BATCH; ++i) {
      Affine<F> B;
      pool_load_aff(src, b0 + base + i, B);
      d[i] = f_norm(f_sub(B.x, A.x));
      const bool zero = f_is_zero(d[i]);
      exc[i] = zero && !A.inf && !B.inf;              // B = +-A
      if (zero || A.inf || B.inf) f_set_one(d[i]);    // keeps the batch's product invertible; the chord of such an entry is not used
      pre[i] = i ? f_mul(pre[i - 1], d[i]) : d[i];
    }
    F inv = f_inv(pre[BATCH - 1]);
#pragma unroll
    for (int i = BATCH - 1; i >= 0; --i) {
      const F dinv = i ? f_mul(inv, pre[i - 1]) : inv;
      if (i) inv = f_mul(inv, d[i]);
      Affine<F> B, R;
      pool_load_aff(src, b0 + base + i, B);
      aff_add_given_inv(R, A, B, dinv);
      aff_select(R, B.inf, A, R);                     // identity operands: copies
      aff_select(R, A.inf, B, R);
      if (BN_WAVE_ANY(exc[i])) {                      // rare: the complete formula (and an inversion of its own) for the lanes that met B = +-A
        Jac<F> J;
        Affine<F> Bc = B, C;
        jac_from_affine(J, A);
        Bc.inf = !exc[i];                             // the other lanes add nothing here
        jac_madd(J, J, Bc);
        jac_to_affine(C, J);
        if (exc[i]) R = C;
      }
      if (live) pool_store_aff(dst, dst0 + base + i, R);
    }
  }
}
// keys: T16[k][hi * 256 + lo] = T8[2k][lo] + T8[2k + 1][hi]; lane = (k, hi, block of 32 lo values) — 2 048 waves for 1 024 signers.  A
// chunk whose second group does not exist (an odd number of groups) only ever sees hi = 0.
#define BN_WIDEN_G2_NLO 32
KERNEL void k_pool_widen_g2(Pool t8, size_t n_groups, size_t n_chunks, Pool t16) {
  const size_t lane = (size_t)blockIdx.x * BN_WAVE + threadIdx.x;
  constexpr size_t BLK = 256 / BN_WIDEN_G2_NLO;
  const bool live = lane < n_chunks * 256 * BLK;
  const size_t ll = live ? lane : 0, blk = ll % BLK, hi = (ll / BLK) & 255u, k = ll / (BLK * 256);
  const bool has_hi = 2 * k + 1 < n_groups;
  G2Affine A;
  pool_load_aff(t8, (has_hi ? 2 * k + 1 : 2 * k) * 256 + hi, A);
  A.inf = A.inf || !has_hi || hi == 0;
  if (A.inf) g2_set_generator_keep_inf(A);
  pool_widen_lane<Fp2, BN_WIDEN_G2_NLO>(live, t8, 2 * k * 256 + blk * BN_WIDEN_G2_NLO, A, t16, k * 65536 + hi * 256 + blk * BN_WIDEN_G2_NLO);
}
// signatures, per message: T8[m][g][hi * 16 + lo] = T4[m][2g][lo] + T4[m][2g + 1][hi]; lane = (m, g, hi)
KERNEL_SMALL void k_pool_widen_g1(Pool t4, size_t groups4, size_t n_groups, size_t n_msgs, Pool t8) {
  const size_t lane = (size_t)blockIdx.x * BN_WAVE + threadIdx.x;
  const bool live = lane < n_msgs * n_groups * 16;
  const size_t ll = live ? lane : 0, hi = ll & 15u, g = (ll >> 4) % n_groups, m = (ll >> 4) / n_groups;
  G1Affine A;
  pool_load_aff(t4, (m * groups4 + 2 * g + 1) * 16 + hi, A);
  A.inf = A.inf || hi == 0;
  if (A.inf) { A.x = fp_load_const(C_G1_GEN[0]); A.y = fp_load_const(C_G1_GEN[1]); }
  pool_widen_lane<Fp, 16>(live, t4, (m * groups4 + 2 * g) * 16, A, t8, (m * n_groups + g) * 256 + hi * 16);
}
// tuple i: agg_sig = sum_s sig_pool[msg_i * S + s], agg_pk = sum_s pk_pool[s] over its signer list
// (Add for Signature / PublicKey, types.rs:264-270, :126-132); results + H(msg_i) go to the verify planes.
// A wave walks its lanes' lists in lockstep until the longest is exhausted.
KERNEL void k_aggregate(const uint32_t* tuple_msg, const uint64_t* tuple_off, const uint32_t* signer_idx, size_t n, size_t n_signers, size_t n_msgs,
                        Pool pk_pool, Pool sig_pool, Pool h_pool, Ws ws) {
  size_t i = (size_t)blockIdx.x * BN_WAVE + threadIdx.x;
  if (i >= n) return;
  uint32_t m = tuple_msg[i];
  uint64_t lo = tuple_off[i], hi = tuple_off[i + 1];
  G1Jac acc1;
  G2Jac acc2;
  jac_set_identity(acc1);
  jac_set_identity(acc2);
  uint8_t st = ST_OK;
  // indices come from caller memory: a message index out of range is IndexOutOfBounds like a signer index, and a
  // decreasing offset pair is an empty list — never an out-of-range pool read
  if (m >= n_msgs) { st = ST_INDEX_OOB; m = 0; }
  if (hi < lo) { if (st == ST_OK) st = ST_INDEX_OOB; hi = lo; }
  uint64_t longest = hi - lo;
  for (int off = 32; off > 0; off >>= 1) {
    uint64_t other = __shfl_xor((unsigned long long)longest, off, BN_WAVE);
    longest = other > longest ? other : longest;
  }
  for (uint64_t t = 0; t < longest; ++t) {
    bool active = lo + t < hi;
    uint32_t sgn = active ? signer_idx[lo + t] : 0u;
    bool valid = active && sgn < n_signers;
    if (active && !valid && st == ST_OK) st = ST_INDEX_OOB;             // IndexOutOfBounds
    if (!valid) sgn = 0;
    G1Affine sp;
    G2Affine pp;
    size_t sj = (size_t)m * n_signers + sgn;
    sp.x = pool_load_fp(sig_pool, 0, sj); sp.y = pool_load_fp(sig_pool, 1, sj);
    uint8_t s1 = sig_pool.st[sj];
    pp.x.c0 = pool_load_fp(pk_pool, 0, sgn); pp.x.c1 = pool_load_fp(pk_pool, 1, sgn);
    pp.y.c0 = pool_load_fp(pk_pool, 2, sgn); pp.y.c1 = pool_load_fp(pk_pool, 3, sgn);
    uint8_t s2 = pk_pool.st[sgn];
    if (valid && st == ST_OK && (s1 & 0x7f)) st = s1 & 0x7f;
    if (valid && st == ST_OK && (s2 & 0x7f)) st = s2 & 0x7f;
    sp.inf = !valid || (s1 & 0x80);
    pp.inf = !valid || (s2 & 0x80);
    jac_accumulate(acc1, sp);
    jac_accumulate(acc2, pp);
  }
  G1Affine asig, h;
  G2Affine apk;
  jac_to_affine(asig, acc1);
  jac_to_affine(apk, acc2);
  h.x = pool_load_fp(h_pool, 0, m); h.y = pool_load_fp(h_pool, 1, m); h.inf = false;
  ws_store_g1(ws, PL_P1X, BY_P1_INF, i, asig);
  ws_store_g2(ws, i, apk);
  ws_store_g1(ws, PL_P2X, BY_P2_INF, i, h);
  ws_byte(ws, BY_ST_DECODE, i) = st;
  ws_byte(ws, BY_ST_HASH, i) = h_pool.st[m];
}
// copy the hash planes of the M messages into a pool
// Aggregate verify, large batches: the tuples are BUCKETED BY MESSAGE before the aggregation kernel (counting sort into an index
// map; results still land at the tuple's own index).  The kernel gathers signature sums from per-message subset tables (~0.3 MB
// each): with the caller's (random) order every lane pair of a workgroup reads another table and nothing stays in a cache; in
// bucket order a workgroup reads ONE message's table, and lanes that share a group index fetch from the same 16-entry block.
// Order inside a bucket depends on the atomics — irrelevant: every tuple is computed for itself.
KERNEL_SMALL void k_agg_sort_count(const uint32_t* tuple_msg, size_t n, uint32_t n_msgs, uint32_t* cnt) {
  const size_t i = (size_t)blockIdx.x * BN_WAVE + threadIdx.x;
  if (i >= n) return;
  const uint32_t m = tuple_msg[i];
  atomicAdd(&cnt[m < n_msgs ? m : n_msgs], 1u);          // out-of-range message indices share the last bucket
}
// one wave: exclusive prefix sums of cnt[0 .. n_buckets) into cursor[] (the scatter's running positions)
KERNEL_SMALL void k_agg_sort_scan(uint32_t n_buckets, const uint32_t* cnt, uint32_t* cursor) {
  const unsigned t = threadIdx.x;
  const uint32_t per = (n_buckets + BN_WAVE - 1) / BN_WAVE, lo = t * per, hi = lo + per < n_buckets ? lo + per : n_buckets;
  uint32_t sum = 0;
  for (uint32_t k = lo; k < hi; ++k) sum += cnt[k];
  uint32_t incl = sum;
  for (int off = 1; off < BN_WAVE; off <<= 1) {
    const uint32_t up = __shfl_up(incl, off, BN_WAVE);
    if ((int)t >= off) incl += up;
  }
  uint32_t run = incl - sum;
  for (uint32_t k = lo; k < hi; ++k) { const uint32_t c = cnt[k]; cursor[k] = run; run += c; }
}
KERNEL_SMALL void k_agg_sort_scatter(const uint32_t* tuple_msg, size_t n, uint32_t n_msgs, uint32_t* cursor, uint32_t* perm) {
  const size_t i = (size_t)blockIdx.x * BN_WAVE + threadIdx.x;
  if (i >= n) return;
  const uint32_t m = tuple_msg[i];
  perm[atomicAdd(&cursor[m < n_msgs ? m : n_msgs], 1u)] = (uint32_t)i;
}
KERNEL_SMALL void k_hash_to_pool(size_t n_msgs, Ws ws, Pool h_pool) {
  size_t i = (size_t)blockIdx.x * BN_WAVE + threadIdx.x;
  if (i >= n_msgs) return;
  pool_store_fp(h_pool, 0, i, ws_load_fp(ws, PL_P2X, i));
  pool_store_fp(h_pool, 1, i, ws_load_fp(ws, PL_P2X + 1, i));
  h_pool.st[i] = ws_byte(ws, BY_ST_HASH, i);
}
// encode the G1 planes (px, px+1) as uncompressed bytes
KERNEL_SMALL void k_encode_g1(size_t n, Ws ws, int px, int inf_plane, uint8_t* out, uint8_t* status_out) {
  size_t i = (size_t)blockIdx.x * BN_WAVE + threadIdx.x;
  if (i >= n) return;
  G1Affine p;
  ws_load_g1(ws, px, inf_plane, i, p);
  uint8_t st = ws_byte(ws, BY_ST_HASH, i);
  if (st != ST_OK) p.inf = true;
  encode_g1(out + 64 * i, p);
  if (status_out) status_out[i] = st;
}

// --- test hooks ---------------------------------------------------------------------------
KERNEL_SMALL void k_debug_fp_op(int op, const uint8_t* a, const uint8_t* b, size_t n, uint8_t* out, uint8_t* status) {
  size_t i = (size_t)blockIdx.x * BN_WAVE + threadIdx.x;
  if (i >= n) return;
  uint32_t any = 0;
  Fp x, y, r;
  bool ok = fp_from_be(x, a + 32 * i, any);
  if (b) ok = fp_from_be(y, b + 32 * i, any) && ok; else y = fp_zero();
  uint8_t st = ok ? ST_OK : ST_NOT_MEMBER;
  switch (op) {
    case 0: r = fp_mul(x, y); break;
    case 1: r = fp_add(x, y); break;
    case 2: r = fp_sub(x, y); break;
    case 3: r = fp_inv(x); break;
    case 4: r = fp_sqr(x); break;
    default: if (!fp_sqrt(r, x) && st == ST_OK) st = ST_NOT_MEMBER; break;
  }
  fp_to_be(out + 32 * i, r);
  status[i] = st;
}
// The try loop's treatment of ONE chosen digest value (32 B big-endian): range rules + mod_u256, the Jacobi filter of
// k_hash_round and the square root of k_hash_finish.  status 0 = yields the point written to out, 1 = next counter;
// bit 7 set = filter and square root disagree (never expected).
KERNEL_SMALL void k_debug_hash_candidate(const uint8_t* h, size_t n, uint8_t* out, uint8_t* status) {
  size_t i = (size_t)blockIdx.x * BN_WAVE + threadIdx.x;
  if (i >= n) return;
  U256 x;
  const uint32_t* w = (const uint32_t*)(h + 32 * i);
#pragma unroll
  for (int k = 0; k < 8; ++k) x.w[7 - k] = __builtin_bswap32(w[k]);
  bool cand = hash_reduce_candidate(x);
  bool filt = false, ok = false;
  G1Affine p;
  g1_set_generator(p);
  if (cand) {
    Fp xm, rhs;
    hash_curve_rhs(xm, rhs, x);
    filt = u256_is_square_mod_q(fp_to_u256(rhs));
    ok = hash_point_from_candidate(p, x);
  }
  if (!ok) p.inf = true;
  encode_g1(out + 64 * i, p);
  status[i] = (uint8_t)((ok ? 0 : 1) | (filt != ok ? 0x80 : 0));
}
__device__ __forceinline__ void decode_fp12(Fp12& f, const uint8_t* b) {
  uint32_t any = 0;
  Fp2* c[6] = {&f.c0.c0, &f.c0.c1, &f.c0.c2, &f.c1.c0, &f.c1.c1, &f.c1.c2};
  for (int k = 0; k < 6; ++k) { fp_from_be(c[k]->c0, b + 64 * k, any); fp_from_be(c[k]->c1, b + 64 * k + 32, any); }
}
KERNEL void k_debug_fp12_op(int op, const uint8_t* a, const uint8_t* b, size_t n, uint8_t* out) {
  size_t i = (size_t)blockIdx.x * BN_WAVE + threadIdx.x;
  if (i >= n) return;
  Fp12 x, y, r;
  decode_fp12(x, a + 384 * i);
  if (b) decode_fp12(y, b + 384 * i); else fp12_set_one(y);
  switch (op) {
    case 0: fp12_mul(r, x, y); break;
    case 1: fp12_sqr(r, x); break;
    case 2: fp12_inv(r, x); break;
    case 3: fp12_conj(r, x); break;
    case 4: fp12_frob(r, x, 1); break;
    case 5: fp12_frob(r, x, 2); break;
    case 6: fp12_frob(r, x, 3); break;
    case 7: fp12_cyclotomic_sqr(r, x); break;
    default: { Fp12 acc; final_exponentiation(r, x, acc); } break;
  }
  encode_fp12(out + 384 * i, r);
}

// test hook: LIMB vectors straight into the F planes of the workspace (12 coefficients x 9 int32 limbs per item, Gt order) — the input of a
// final exponentiation with non-canonical / extreme-digit representatives that no byte decoder would produce
KERNEL_SMALL void k_debug_load_f(const int32_t* limbs, size_t n, Ws ws) {
  size_t i = (size_t)blockIdx.x * BN_WAVE + threadIdx.x;
  if (i >= n) return;
  for (int e = 0; e < 12; ++e) {
    Fp x;
#pragma unroll
    for (int k = 0; k < BN_LIMBS; ++k) x.v[k] = limbs[(i * 12 + e) * BN_LIMBS + k];
    ws_store_fp(ws, PL_F0 + e, i, x);
  }
  ws_byte(ws, BY_ST_DECODE, i) = ST_OK;
  ws_byte(ws, BY_ST_HASH, i) = ST_OK;
}

// ---- in-process issue-rate probe (bench.py's roofline calibration) --------------------------------------------
// 16 independent chains of one instruction, 4096 trips, on every SIMD of the device with `waves_per_simd` waves each
// (256-thread workgroups = one wave per SIMD of a CU, like the pair kernels).  op 0: v_mad_u64_u32, 1: v_add_u32,
// 2: v_mul_lo_u32.  The standalone sweep over more instructions is bn254_amd/csrc/microbench/valu_rates.hip.
#define PROBE_ITERS 4096
#define PROBE_CHAINS 16
template <int OP>
__global__ void __launch_bounds__(256) k_issue_probe(uint32_t* out, uint32_t seed, unsigned long long* clk) {
  unsigned long long clk0 = 0, wall0 = 0;
  if (clk && threadIdx.x == 0) { clk0 = clock64(); wall0 = wall_clock64(); }
  uint32_t a = seed + threadIdx.x * 2654435761u, b = seed ^ (threadIdx.x * 40503u + 977u);
  uint64_t acc[PROBE_CHAINS];
#pragma unroll
  for (int j = 0; j < PROBE_CHAINS; ++j) acc[j] = a + j;
  for (int i = 0; i < PROBE_ITERS; ++i) {
#pragma unroll
    for (int j = 0; j < PROBE_CHAINS; ++j) {
      if (OP == 0) {
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b) : "vcc");
      } else if (OP == 1) {
        uint32_t lo = (uint32_t)acc[j];
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(lo) : "v"(a));
        acc[j] = lo;
      } else {
        uint32_t lo = (uint32_t)acc[j];
        asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(lo) : "v"(b));
        acc[j] = lo;
      }
    }
  }
  uint64_t sum = 0;
#pragma unroll
  for (int j = 0; j < PROBE_CHAINS; ++j) sum += acc[j];
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = (uint32_t)sum ^ (uint32_t)(sum >> 32);
  if (clk && threadIdx.x == 0 && blockIdx.x < BN_CLK_MAX_WG) {     // slot 2 of the clock probe (bn254_ws.h): this kernel's own clock
    unsigned long long* p = clk + ((size_t)2 * BN_CLK_MAX_WG + blockIdx.x) * 2;
    p[0] += clock64() - clk0; p[1] += wall_clock64() - wall0;
  }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
struct bn254_ctx {
  int device;
  hipStream_t stream;
  Ws ws;
  // staging buffers for the host-pointer entry points (device memory, grown on demand)
  uint8_t* stage[8];
  size_t stage_cap[8];
  int profiling;
  int split_miller;  // A/B knob: one pairing per lane (k_miller_verify_split) instead of the fused 2-pair loop
  Pool pool[7];       // aggregate verify: pk pool, sig pool, H(m) pool, subset sums of the pk pool and of the signature pool, and their widened
                      // forms (16 keys / 8 signatures per entry) for the largest batches (grown on demand)
  size_t pool_fp[7];  // coordinates per entry: 4, 2, 2, 4, 2, 4, 2
  int agg_wide_min_tuples;    // aggregate verify: the widened tables from this many tuples on (0 = never)
  int agg_subset_min_tuples;  // aggregate verify: tabulate subset sums of the pk pool for batches of at least this many tuples (0 = never)
  int agg_sort_by_msg;        // aggregate verify: bucket the tuples by message before the aggregation kernel (default 1; A/B and test knob)
  int pair_lanes;    // verify: Miller loop + final exponentiation on lane pairs (bn254_pair.hip); default on
  int rand_min_batch;      // randomised verify: batches below this size run the exact kernels (default RAND_MIN_BATCH_DEFAULT)
  int rand_items_per_lane; // randomised verify: 0 = by batch size, 1 or 2 forced (A/B and tests)
  int hash_max_tries; // test knob: counters tried before HashToPointError (0 = the reference's 255)
  int trio_wave_roles; // octet layout: the Miller loop's four lane pairs as the four waves of a workgroup (k_miller_verify_quad) instead of one wave
  int hash_direct_width; // small batches: counters tried at once with the square root itself (k_hash_direct); 0 = rounds only
  int trio_max_batch; // verify / check_public_keys batches up to this size run in the octet layout (bn254_trio.hip); 0 = never
  int nonet_max_batch; // ... and up to this size their final exponentiation runs on nine lane pairs per verify (bn254_nonet.hip); 0 = never
  hipEvent_t ev[5];
  int ev_valid;
  int ev_hash_first;   // the recorded intervals are hash, decode, ... (host-pointer verify) instead of decode, hash, ...
  hipStream_t copy_stream;   // host-pointer verify: signatures and keys cross PCIe here while the hash rounds run on `stream`
  uint8_t* pin;              // ... through this PINNED host buffer (hipHostMalloc, grown on demand): BN254_OPT_PINNED_STAGING
  size_t pin_cap;
  int pinned_staging;        // 0 = hipMemcpyAsync straight from the caller's (pageable) buffers
  hipEvent_t copy_done;
  uint64_t msgs_len_next;    // bn254_ctx_expect_msgs_len: size of the d_msgs buffer of the NEXT call that hashes messages
  int msgs_len_declared;
  uint64_t msgs_len_call;    // ... as taken by the entry point now running (MsgsLenScope); UINT64_MAX = not declared
  int entry_depth;           // the host-pointer entry points call their *_device forms: only the outermost one takes the declaration
  int32_t* key_lines;        // keyed verify: registered keys (bn254_ctx_register_keys), see KeyTable in bn254_ws.h
  int32_t* key_xy;           // ... and their affine coordinates (4 x 9 words per key) for the small-batch route
  uint8_t* key_st;
  uint8_t* key_inf;
  size_t n_keys, key_cap;
  bool fits_w8, fits_quad, fits_trio;   // the device can hold a workgroup of the small-batch kernels (LDS), asked at creation
};



struct ScopedEvents {
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipError_t create() {
    hipError_t e = hipEventCreate(&e0);
    return e == hipSuccess ? hipEventCreate(&e1) : e;
  }
  ~ScopedEvents() { if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); }
  ScopedEvents() = default;
  ScopedEvents(const ScopedEvents&) = delete;
  ScopedEvents& operator=(const ScopedEvents&) = delete;
};
static inline unsigned grid_for(size_t n) { return (unsigned)((n + BN_WAVE - 1) / BN_WAVE); }

static int ws_reserve(bn254_ctx* c, size_t n) {
  if (n <= c->ws.stride) return 0;
  size_t cap = (n + 255) & ~(size_t)255;
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipDeviceSynchronize());     // a *_device call may still be running on a caller's stream, not only on c->stream
  if (c->ws.planes) { HIP_TRY(hipFree(c->ws.planes)); c->ws.planes = nullptr; }
  if (c->ws.bytes) { HIP_TRY(hipFree(c->ws.bytes)); c->ws.bytes = nullptr; }
  if (c->ws.h_best) { HIP_TRY(hipFree(c->ws.h_best)); c->ws.h_best = nullptr; }
  if (c->ws.h_next) { HIP_TRY(hipFree(c->ws.h_next)); c->ws.h_next = nullptr; }
  if (c->ws.h_list) { HIP_TRY(hipFree(c->ws.h_list)); c->ws.h_list = nullptr; }
  c->ws.stride = 0;
  HIP_TRY(hipMalloc((void**)&c->ws.planes, (size_t)N_PLANES * BN_LIMBS * sizeof(int32_t) * cap));
  HIP_TRY(hipMalloc((void**)&c->ws.bytes, (size_t)N_BYTE_PLANES * cap));
  HIP_TRY(hipMalloc((void**)&c->ws.h_best, sizeof(uint32_t) * cap));
  HIP_TRY(hipMalloc((void**)&c->ws.h_next, cap));
  HIP_TRY(hipMalloc((void**)&c->ws.h_list, 2 * sizeof(uint32_t) * cap));
  if (!c->ws.h_cnt) HIP_TRY(hipMalloc((void**)&c->ws.h_cnt, sizeof(uint32_t) * (HASH_MAX_ROUNDS + 1)));
  c->ws.stride = cap;
  return 0;
}
static int stage_reserve(bn254_ctx* c, int slot, size_t bytes) {
  if (bytes <= c->stage_cap[slot]) return 0;
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipDeviceSynchronize());
  if (c->stage[slot]) { HIP_TRY(hipFree(c->stage[slot])); c->stage[slot] = nullptr; c->stage_cap[slot] = 0; }
  size_t cap = (bytes + 4095) & ~(size_t)4095;
  HIP_TRY(hipMalloc((void**)&c->stage[slot], cap));
  c->stage_cap[slot] = cap;
  return 0;
}
static int stage_in(bn254_ctx* c, int slot, const void* host, size_t bytes) {
  int rc = stage_reserve(c, slot, bytes ? bytes : 1);
  if (rc) return rc;
  if (bytes) HIP_TRY(hipMemcpyAsync(c->stage[slot], host, bytes, hipMemcpyHostToDevice, c->stream));
  return 0;
}
static int stage_out(bn254_ctx* c, int slot, void* host, size_t bytes) {
  if (bytes) HIP_TRY(hipMemcpyAsync(host, c->stage[slot], bytes, hipMemcpyDeviceToHost, c->stream));
  return 0;
}
static bool misaligned(const void* p) { return ((uintptr_t)p & 3u) != 0; }
// host-pointer entry points: an offsets array (n + 1 entries) must be non-decreasing — a kernel computes lengths as
// off[i+1] - off[i], and a wrapped length walks far outside the staged buffer.  O(n) on memory the host already has.
// (The *_device variants cannot look: there it is a documented precondition, include/bn254_hip.h.)
static bool offsets_ok(const uint64_t* off, size_t n) {
  for (size_t i = 0; i < n; ++i) if (off[i] > off[i + 1]) return false;
  return true;
}
static int pool_reserve(bn254_ctx* c, int which, size_t n_fp, size_t entries) {
  Pool& p = c->pool[which];
  if (entries <= p.stride && c->pool_fp[which] == n_fp) return 0;
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipDeviceSynchronize());
  if (p.planes) { HIP_TRY(hipFree(p.planes)); p.planes = nullptr; }
  if (p.st) { HIP_TRY(hipFree(p.st)); p.st = nullptr; }
  p.stride = 0;
  size_t cap = (entries + 255) & ~(size_t)255;
  p.g2 = n_fp == 4 ? 1u : 0u;                       // record layout: bn254_ws.h
  HIP_TRY(hipMalloc((void**)&p.planes, (n_fp / 2) * BN_POOL_HALF_WORDS * sizeof(int32_t) * cap));
  HIP_TRY(hipMalloc((void**)&p.st, cap));
  p.stride = cap;
  c->pool_fp[which] = n_fp;
  return 0;
}

// G2 decoding: with the subgroup test requested (one 63-bit ladder on the twist per point) it runs on lane pairs
static int launch_decode_g2(bn254_ctx* c, hipStream_t s, const uint8_t* d_pts, size_t n, uint32_t flags, int accumulate) {
  if (c->pair_lanes && (flags & FLAG_G2_SUBGROUP_CHECK)) return bn254_pair_decode_g2(d_pts, n, flags, c->ws, accumulate, s);
  k_decode_g2<<<grid_for(n), BN_WAVE, 0, s>>>(d_pts, n, flags, c->ws, accumulate);
  return 0;
}

// bn254_ctx_expect_msgs_len is consumed by the NEXT entry point that hashes messages — whatever that call goes on to do: every such
// entry point opens with a MsgsLenScope, which takes the declaration and clears it before any argument check, staging step or
// allocation can return early (a declaration left armed would bound-check an unrelated later call against the wrong length).
struct MsgsLenScope {
  bn254_ctx* c;
  explicit MsgsLenScope(bn254_ctx* ctx) : c(ctx) {
    if (c && c->entry_depth++ == 0) {
      c->msgs_len_call = c->msgs_len_declared ? c->msgs_len_next : UINT64_MAX;
      c->msgs_len_declared = 0;
    }
  }
  ~MsgsLenScope() { if (c) --c->entry_depth; }
  MsgsLenScope(const MsgsLenScope&) = delete;
  MsgsLenScope& operator=(const MsgsLenScope&) = delete;
};

// Enqueue the hash-to-G1 rounds for n messages; points land in planes (px, px+1), statuses in BY_ST_HASH.
// The schedule (widths, grid sizes) is fixed on the host from the EXPECTED survivor counts
// (p_fail = 0.5274 per try); the kernels read the actual counts from device memory and use grid-stride
// loops, so a wrong estimate costs time, never correctness.  No host synchronisation.
static int launch_hash_rounds(bn254_ctx* c, hipStream_t s, const uint8_t* d_msgs, const uint64_t* d_off, size_t n, int px, int inf_plane,
                              uint8_t* d_tries, int mark_finish = -1) {      // mark_finish: profiling event recorded in front of k_hash_finish
  const uint32_t max_ctr = c->hash_max_tries ? (uint32_t)c->hash_max_tries : 255u;
  const uint64_t msgs_len = c->msgs_len_call;   // bn254_ctx_expect_msgs_len, taken by the entry point's MsgsLenScope
  k_hash_init<<<grid_for(n > HASH_MAX_ROUNDS + 1 ? n : HASH_MAX_ROUNDS + 1), BN_WAVE, 0, s>>>(n, c->ws);
  if (n <= HASH_DIRECT_MAX_N && c->hash_direct_width > 0) {
    const uint32_t width = (uint32_t)c->hash_direct_width;
    k_hash_direct<<<grid_for(n * width), BN_WAVE, 0, s>>>(d_msgs, d_off, msgs_len, n, c->ws, width, max_ctr, px, inf_plane, d_tries);
    if (max_ctr > width)                 // the (rare) survivors: every remaining counter at once (grid-stride beyond 64 of them)
      k_hash_round<<<grid_for(64 * (max_ctr - width)), BN_WAVE, 0, s>>>(d_msgs, d_off, msgs_len, c->ws, 1, max_ctr - width, max_ctr);
    if (mark_finish >= 0 && c->profiling) HIP_TRY(hipEventRecord(c->ev[mark_finish], s));
    k_hash_finish<<<grid_for(n), BN_WAVE, 0, s>>>(d_msgs, d_off, msgs_len, n, c->ws, max_ctr, px, inf_plane, d_tries);
    HIP_TRY(hipGetLastError());
    return 0;
  }
  double expect = (double)n;
  uint32_t consumed = 0;
  for (int round = 0; round < HASH_MAX_ROUNDS && consumed < max_ctr; ++round) {
    bool last = round == HASH_MAX_ROUNDS - 1;
    double lanes_per_msg = (double)HASH_TARGET_LANES / (expect < 1.0 ? 1.0 : expect);
    uint32_t width = lanes_per_msg < 2.0 ? 1u : (uint32_t)lanes_per_msg;
    if (width > max_ctr - consumed || last) width = max_ctr - consumed;
    double bound = expect * 1.25 + 256.0;                 // generous estimate of the survivors
    if (bound > (double)n) bound = (double)n;
    size_t lanes = (size_t)(bound * width);
    if (lanes > HASH_MAX_GRID_LANES) lanes = HASH_MAX_GRID_LANES;   // grid-stride loops cover the rest
    k_hash_round<<<grid_for(lanes), BN_WAVE, 0, s>>>(d_msgs, d_off, msgs_len, c->ws, round, width, max_ctr);
    if ((size_t)bound > HASH_RESOLVE_TILES_MIN) {
      size_t tiles = ((size_t)bound + HASH_RESOLVE_WG - 1) / HASH_RESOLVE_WG;
      if (tiles > 16384) tiles = 16384;                     // grid-stride beyond
      k_hash_resolve_tiles<<<(unsigned)tiles, HASH_RESOLVE_WG, 0, s>>>(c->ws, round, width, max_ctr);
    } else {
      k_hash_resolve<<<grid_for((size_t)bound), BN_WAVE, 0, s>>>(c->ws, round, width, max_ctr);
    }
    consumed += width;
    double pf = 1.0;
    for (uint32_t t = 0; t < width && pf > 1e-12; ++t) pf *= 0.5274;
    expect *= pf;
  }
  if (mark_finish >= 0 && c->profiling) HIP_TRY(hipEventRecord(c->ev[mark_finish], s));
  k_hash_finish<<<grid_for(n), BN_WAVE, 0, s>>>(d_msgs, d_off, msgs_len, n, c->ws, max_ctr, px, inf_plane, d_tries);
  HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" {

const char* bn254_version(void) { return "bn254-mi355x 0.6 (gfx950; 9x29-bit balanced Montgomery limbs; verify on lane pairs, batches <= 16384 on lane octets with wave roles)"; }

int bn254_ctx_create(int hip_device, bn254_ctx** out) {
  if (!out) return BN254_E_BAD_ARGUMENT;
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) return BN254_E_NO_DEVICE;   // no CPU fallback, by design
  if (hip_device < 0 || hip_device >= count) return BN254_E_BAD_ARGUMENT;
  HIP_TRY(hipSetDevice(hip_device));
  bn254_ctx* c = new (std::nothrow) bn254_ctx();
  if (!c) return BN254_E_BAD_ARGUMENT;
  memset(c, 0, sizeof *c);
  c->pair_lanes = 1;
  c->rand_min_batch = RAND_MIN_BATCH_DEFAULT;
  c->trio_max_batch = TRIO_MAX_BATCH_DEFAULT;
  c->hash_direct_width = HASH_DIRECT_WIDTH_DEFAULT;
  c->trio_wave_roles = TRIO_WAVE_ROLES_DEFAULT;
  c->agg_subset_min_tuples = AGG_SUBSET_MIN_TUPLES_DEFAULT;
  c->agg_sort_by_msg = 1;
  c->agg_wide_min_tuples = AGG_WIDE_MIN_TUPLES_DEFAULT;
  c->pinned_staging = PINNED_STAGING_DEFAULT;
  // the small-batch kernels ask for up to 156 KB of dynamic LDS per workgroup: on a part that cannot hold one, step down
  // (eight wave roles -> four -> lane groups -> lane pairs only) instead of failing at the first launch
  c->fits_w8 = bn254_quad_fits_device(1); c->fits_quad = bn254_quad_fits_device(0); c->fits_trio = bn254_trio_fits_device();
  if (c->trio_wave_roles == 2 && !c->fits_w8) c->trio_wave_roles = 1;
  if (c->trio_wave_roles == 1 && !c->fits_quad) c->trio_wave_roles = 0;
  if (!c->fits_trio) c->trio_max_batch = 0;
  c->nonet_max_batch = bn254_nonet_fits_device() ? NONET_MAX_BATCH_DEFAULT : 0;
  c->device = hip_device;
  hipError_t err = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (err == hipSuccess) err = hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking);
  if (err == hipSuccess) err = hipEventCreateWithFlags(&c->copy_done, hipEventDisableTiming);
  for (int i = 0; i < 5 && err == hipSuccess; ++i) err = hipEventCreate(&c->ev[i]);
  if (err != hipSuccess) {
    for (int i = 0; i < 5; ++i) if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
    if (c->copy_done) (void)hipEventDestroy(c->copy_done);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return -(int)err;
  }
  *out = c;
  return 0;
}
void bn254_ctx_destroy(bn254_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  if (c->ws.planes) (void)hipFree(c->ws.planes);
  if (c->ws.bytes) (void)hipFree(c->ws.bytes);
  if (c->ws.h_best) (void)hipFree(c->ws.h_best);
  if (c->ws.h_next) (void)hipFree(c->ws.h_next);
  if (c->ws.h_list) (void)hipFree(c->ws.h_list);
  if (c->ws.h_cnt) (void)hipFree(c->ws.h_cnt);
  if (c->ws.clk) (void)hipFree(c->ws.clk);
  if (c->pin) (void)hipHostFree(c->pin);
  for (int i = 0; i < 7; ++i) { if (c->pool[i].planes) (void)hipFree(c->pool[i].planes); if (c->pool[i].st) (void)hipFree(c->pool[i].st); }
  for (int i = 0; i < 8; ++i) if (c->stage[i]) (void)hipFree(c->stage[i]);
  if (c->key_lines) (void)hipFree(c->key_lines);
  if (c->key_xy) (void)hipFree(c->key_xy);
  if (c->key_st) (void)hipFree(c->key_st);
  if (c->key_inf) (void)hipFree(c->key_inf);
  for (int i = 0; i < 5; ++i) (void)hipEventDestroy(c->ev[i]);
  (void)hipEventDestroy(c->copy_done);
  (void)hipStreamSynchronize(c->copy_stream);
  (void)hipStreamDestroy(c->copy_stream);
  (void)hipStreamDestroy(c->stream);
  delete c;
}
int bn254_ctx_reserve(bn254_ctx* c, size_t n) { return c ? ws_reserve(c, n) : BN254_E_BAD_ARGUMENT; }
int bn254_ctx_synchronize(bn254_ctx* c) {
  if (!c) return BN254_E_BAD_ARGUMENT;
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}
int bn254_ctx_expect_msgs_len(bn254_ctx* c, uint64_t msgs_len) {
  if (!c) return BN254_E_BAD_ARGUMENT;
  c->msgs_len_next = msgs_len;
  c->msgs_len_declared = 1;
  return 0;
}
int bn254_ctx_set_profiling(bn254_ctx* c, int enabled) {
  if (!c) return BN254_E_BAD_ARGUMENT;
  c->profiling = enabled;
  c->ev_valid = 0;
  return 0;
}
int bn254_ctx_set_option(bn254_ctx* c, int option, int value) {
  if (!c) return BN254_E_BAD_ARGUMENT;
  if (option == BN254_OPT_SPLIT_MILLER) { c->split_miller = value; return 0; }
  if (option == BN254_OPT_PAIR_LANES) { c->pair_lanes = value != 0; return 0; }
  if (option == BN254_OPT_RAND_MIN_BATCH) { if (value < 0) return BN254_E_BAD_ARGUMENT; c->rand_min_batch = value; return 0; }
  if (option == BN254_OPT_RAND_ITEMS_PER_LANE) { if (value < 0 || value > 2) return BN254_E_BAD_ARGUMENT; c->rand_items_per_lane = value; return 0; }
  if (option == BN254_OPT_AGG_SUBSET_MIN_TUPLES) { if (value < 0) return BN254_E_BAD_ARGUMENT; c->agg_subset_min_tuples = value; return 0; }
  if (option == BN254_OPT_TRIO_MAX_BATCH) {
    if (value < 0 || (value > 0 && !c->fits_trio)) return BN254_E_BAD_ARGUMENT;
    c->trio_max_batch = value;
    return 0;
  }
  if (option == BN254_OPT_TRIO_WAVE_ROLES) {
    if (value < 0 || value > 2 || (value == 2 && !c->fits_w8) || (value == 1 && !c->fits_quad)) return BN254_E_BAD_ARGUMENT;
    c->trio_wave_roles = value;
    return 0;
  }
  if (option == BN254_OPT_HASH_DIRECT_WIDTH) {
    if (value < 0 || value > 32 || (value & (value - 1))) return BN254_E_BAD_ARGUMENT;
    c->hash_direct_width = value;
    return 0;
  }
  if (option == BN254_OPT_NONET_MAX_BATCH) {
    if (value < 0 || (value > 0 && !bn254_nonet_fits_device())) return BN254_E_BAD_ARGUMENT;
    c->nonet_max_batch = value;
    return 0;
  }
  if (option == BN254_OPT_PINNED_STAGING) { if (value < 0 || value > 16) return BN254_E_BAD_ARGUMENT; c->pinned_staging = value; return 0; }
  if (option == BN254_OPT_AGG_SORT_BY_MSG) { c->agg_sort_by_msg = value != 0; return 0; }
  if (option == BN254_OPT_AGG_WIDE_MIN_TUPLES) { if (value < 0) return BN254_E_BAD_ARGUMENT; c->agg_wide_min_tuples = value; return 0; }
  if (option == BN254_OPT_CLOCK_PROBE) {
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipDeviceSynchronize());
    if (value && !c->ws.clk) {
      HIP_TRY(hipMalloc((void**)&c->ws.clk, sizeof(unsigned long long) * 3 * BN_CLK_MAX_WG * 2));
      HIP_TRY(hipMemset(c->ws.clk, 0, sizeof(unsigned long long) * 3 * BN_CLK_MAX_WG * 2));
    } else if (!value && c->ws.clk) {
      HIP_TRY(hipFree(c->ws.clk));
      c->ws.clk = nullptr;
    }
    return 0;
  }
  if (option == BN254_OPT_HASH_MAX_TRIES) { if (value < 0 || value > 255) return BN254_E_BAD_ARGUMENT; c->hash_max_tries = value; return 0; }
  return BN254_E_BAD_ARGUMENT;
}
// clock probe (BN254_OPT_CLOCK_PROBE): the clock the chip ran the last Miller kernel [0], final exponentiation [1] and issue probe [2]
// at, in MHz = shader-clock cycles / constant-rate ticks x the constant rate, summed over the workgroups that reported; 0 = none did
int bn254_ctx_last_clocks(bn254_ctx* c, double sclk_mhz[3]) {
  if (!c || !sclk_mhz || !c->ws.clk) return BN254_E_BAD_ARGUMENT;
  HIP_TRY(hipSetDevice(c->device));
  int wall_khz = 0;
  HIP_TRY(hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, c->device));
  const size_t words = (size_t)3 * BN_CLK_MAX_WG * 2;
  unsigned long long* h = (unsigned long long*)malloc(words * sizeof(unsigned long long));
  if (!h) return BN254_E_NO_MEMORY;
  hipError_t e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy(h, c->ws.clk, words * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemset(c->ws.clk, 0, words * sizeof(unsigned long long));     // read AND cleared: the next reading covers the launches from here on
  if (e != hipSuccess) { free(h); return -(int)e; }
  for (int k = 0; k < 3; ++k) {
    double cyc = 0, wall = 0;
    for (size_t w = 0; w < BN_CLK_MAX_WG; ++w) {
      const unsigned long long a = h[((size_t)k * BN_CLK_MAX_WG + w) * 2], b = h[((size_t)k * BN_CLK_MAX_WG + w) * 2 + 1];
      if (b) { cyc += (double)a; wall += (double)b; }
    }
    sclk_mhz[k] = wall > 0 ? cyc / wall * (double)wall_khz * 1e-3 : 0.0;
  }
  free(h);
  return 0;
}
int bn254_ctx_last_kernel_ms(bn254_ctx* c, float ms[4]) {
  if (!c || !ms || !c->ev_valid) return BN254_E_BAD_ARGUMENT;
  HIP_TRY(hipEventSynchronize(c->ev[4]));
  for (int i = 0; i < 4; ++i) HIP_TRY(hipEventElapsedTime(&ms[i], c->ev[i], c->ev[i + 1]));
  // the host-pointer verify hashes FIRST (the messages cross PCIe first) and decodes second: keep the documented slots
  // (ms[0] decode, ms[1] hash-to-G1 — there including the transfer of the messages)
  if (c->ev_hash_first) { float t = ms[0]; ms[0] = ms[1]; ms[1] = t; }
  return 0;
}

#define PROF_MARK(idx) do { if (c->profiling) HIP_TRY(hipEventRecord(c->ev[idx], s)); } while (0)

// Miller loop + final exponentiation of a verify-shaped batch on lane pairs, or — for batches that cannot fill the chip —
// in the octet layout (three lane pairs share the Fq6 products of every Fq12 operation: fewer instructions per lane,
// which is what latency is made of when a wave has its SIMD to itself).  Same status bytes either way.
static int launch_pair_or_trio(bn254_ctx* c, hipStream_t s, size_t n, int use_hash, uint8_t* d_status, int mode, bool mark) {
  int rc;
  if (c->trio_max_batch > 0 && n <= (size_t)c->trio_max_batch) {
    if ((rc = c->trio_wave_roles == 2 ? bn254_w8_miller_verify(n, c->ws, s, mode)
              : c->trio_wave_roles ? bn254_quad_miller_verify(n, c->ws, s, mode) : bn254_trio_miller_verify(n, c->ws, s, mode))) return rc;
    if (mark) PROF_MARK(3);
    // the smallest batches: nine lane pairs per verify (bn254_nonet.hip) — fewer instructions per lane again, while one pass of 3
    // verifies per wave still covers the batch
    if (c->nonet_max_batch > 0 && n <= (size_t)c->nonet_max_batch) return bn254_nonet_final_exp(n, c->ws, use_hash, d_status, s);
    return bn254_trio_final_exp(n, c->ws, use_hash, d_status, s);
  }
  if ((rc = bn254_pair_miller_verify(n, c->ws, nullptr, nullptr, s, mode))) return rc;
  if (mark) PROF_MARK(3);
  return bn254_pair_final_exp(n, c->ws, use_hash, d_status, nullptr, nullptr, s);
}

// decode kernels have filled the P1 / Q planes and BY_ST_DECODE: hash, Miller loop, final exponentiation
static int verify_after_decode(bn254_ctx* c, hipStream_t s, const uint8_t* d_msgs, const uint64_t* d_off, size_t n, uint8_t* d_status, bool split) {
  int rc;
  unsigned g = grid_for(n);
  PROF_MARK(1);
  if ((rc = launch_hash_rounds(c, s, d_msgs, d_off, n, PL_P2X, BY_P2_INF, nullptr))) return rc;
  PROF_MARK(2);
  if (split) {
    k_miller_verify_split<<<2 * g, BN_WAVE, 0, s>>>(n, c->ws.stride / 2, g, c->ws);
    PROF_MARK(3);
    k_final_exp<<<g, BN_WAVE, 0, s>>>(n, 2, 1, c->ws.stride / 2, c->ws, 1, nullptr, d_status, 0, 0, nullptr, nullptr);
  } else if (c->pair_lanes) {
    if ((rc = launch_pair_or_trio(c, s, n, 1, d_status, 0, true))) return rc;
  } else {
    k_miller_verify<<<g, BN_WAVE, 0, s>>>(n, c->ws, nullptr, nullptr);
    PROF_MARK(3);
    k_final_exp<<<g, BN_WAVE, 0, s>>>(n, 1, 1, 1, c->ws, 1, nullptr, d_status, 0, 0, nullptr, nullptr);
  }
  PROF_MARK(4);
  if (c->profiling) { c->ev_valid = 1; c->ev_hash_first = 0; }
  HIP_TRY(hipGetLastError());
  return 0;
}

int bn254_batch_verify_device(bn254_ctx* c, const uint8_t* d_msgs, const uint64_t* d_off, const uint8_t* d_sigs, const uint8_t* d_pks,
                              size_t n, uint32_t flags, uint8_t* d_status, void* stream) {
  MsgsLenScope msgs_len_scope(c);
  if (!c || (n && (!d_msgs || !d_off || !d_sigs || !d_pks || !d_status))) return BN254_E_BAD_ARGUMENT;
  if (n == 0) return 0;
  if (misaligned(d_sigs) || misaligned(d_pks) || ((uintptr_t)d_off & 7u)) return BN254_E_MISALIGNED;
  HIP_TRY(hipSetDevice(c->device));
  // default: verify on lane pairs (bn254_pair.hip).  BN254_OPT_PAIR_LANES = 0: one lane per verify, fused 2-pair
  // loop; BN254_OPT_SPLIT_MILLER additionally runs one pairing per lane (two waves per verify) — both kept for A/B
  // runs, see profiles/r01_c_ab_occupancy.log and DESIGN.md section 4
  bool split = c->split_miller && n <= BN_SPLIT_MAX_N;
  int rc = ws_reserve(c, split ? 2 * n : n);
  if (rc) return rc;
  hipStream_t s = stream ? (hipStream_t)stream : c->stream;
  PROF_MARK(0);
  k_decode_g1<<<grid_for(n), BN_WAVE, 0, s>>>(d_sigs, n, flags, c->ws, PL_P1X, BY_P1_INF, 0);
  if ((rc = launch_decode_g2(c, s, d_pks, n, flags, 1))) return rc;
  return verify_after_decode(c, s, d_msgs, d_off, n, d_status, split);
}

// the same from the COMPRESSED encodings callers store (serde, /root/reference/src/serde.rs:39, :54):
// signatures 33 B (src/utils.rs:84-104), public keys 65 B (src/utils.rs:130-158, subgroup-checked on decode)
int bn254_batch_verify_compressed_device(bn254_ctx* c, const uint8_t* d_msgs, const uint64_t* d_off, const uint8_t* d_sigs33,
                                         const uint8_t* d_pks65, size_t n, uint8_t* d_status, void* stream) {
  MsgsLenScope msgs_len_scope(c);
  if (!c || (n && (!d_msgs || !d_off || !d_sigs33 || !d_pks65 || !d_status))) return BN254_E_BAD_ARGUMENT;
  if (n == 0) return 0;
  if ((uintptr_t)d_off & 7u) return BN254_E_MISALIGNED;
  HIP_TRY(hipSetDevice(c->device));
  int rc = ws_reserve(c, n);
  if (rc) return rc;
  hipStream_t s = stream ? (hipStream_t)stream : c->stream;
  PROF_MARK(0);
  k_decompress_g1_ws<<<grid_for(n), BN_WAVE, 0, s>>>(d_sigs33, n, c->ws);
  if (c->pair_lanes) { if ((rc = bn254_pair_decompress_g2(d_pks65, n, c->ws, s))) return rc; }
  else k_decompress_g2_ws<<<grid_for(n), BN_WAVE, 0, s>>>(d_pks65, n, c->ws);
  return verify_after_decode(c, s, d_msgs, d_off, n, d_status, false);
}
int bn254_batch_verify_compressed(bn254_ctx* c, const uint8_t* msgs, const uint64_t* off, const uint8_t* sigs33, const uint8_t* pks65, size_t n,
                                  uint8_t* status) {
  MsgsLenScope msgs_len_scope(c);
  if (!c || (n && (!off || !sigs33 || !pks65 || !status))) return BN254_E_BAD_ARGUMENT;
  if (n == 0) return 0;
  HIP_TRY(hipSetDevice(c->device));
  int rc;
  if (!offsets_ok(off, n)) return BN254_E_BAD_ARGUMENT;
  if ((rc = stage_in(c, 0, msgs, (size_t)off[n]))) return rc;
  if ((rc = stage_in(c, 1, off, (n + 1) * sizeof(uint64_t)))) return rc;
  if ((rc = stage_in(c, 2, sigs33, n * 33))) return rc;
  if ((rc = stage_in(c, 3, pks65, n * 65))) return rc;
  if ((rc = stage_reserve(c, 4, n))) return rc;
  if ((rc = bn254_batch_verify_compressed_device(c, c->stage[0], (const uint64_t*)c->stage[1], c->stage[2], c->stage[3], n, c->stage[4], nullptr))) return rc;
  if ((rc = stage_out(c, 4, status, n))) return rc;
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

// Pinned staging (BN254_OPT_PINNED_STAGING = T > 0): the caller's buffers are pageable, and a hipMemcpyAsync from pageable memory is
// staged by the runtime on the calling thread.  With the option the bytes go through the context's own pinned buffer instead: T
// threads (the caller's + T - 1 helpers, started per call) each copy a contiguous share of a buffer into it in 1 MB pieces and
// enqueue the DMA of every piece as soon as it is in place, so page copies and DMA overlap and the DMA runs at the link's rate.
static int pin_reserve(bn254_ctx* c, size_t bytes) {
  if (bytes <= c->pin_cap) return 0;
  HIP_TRY(hipDeviceSynchronize());
  if (c->pin) { HIP_TRY(hipHostFree(c->pin)); c->pin = nullptr; c->pin_cap = 0; }
  const size_t cap = (bytes + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
  HIP_TRY(hipHostMalloc((void**)&c->pin, cap, hipHostMallocDefault));
  c->pin_cap = cap;
  return 0;
}
static int pinned_copy_in(bn254_ctx* c, uint8_t* d_dst, uint8_t* pin, const uint8_t* src, size_t bytes, hipStream_t stream, int threads) {
  if (!bytes) return 0;
  const size_t piece = (size_t)1 << 20;
  if (threads > 1 && bytes < 4 * piece) threads = 1;
  std::vector<int> rcs((size_t)threads, 0);
  auto work = [&](int t) {
    if (t && hipSetDevice(c->device) != hipSuccess) { rcs[t] = -1; return; }
    const size_t share = ((bytes + threads - 1) / threads + 255) & ~(size_t)255, lo = (size_t)t * share, hi = lo + share < bytes ? lo + share : bytes;
    for (size_t o = lo; o < hi; o += piece) {
      const size_t len = o + piece < hi ? piece : hi - o;
      memcpy(pin + o, src + o, len);
      const hipError_t e = hipMemcpyAsync(d_dst + o, pin + o, len, hipMemcpyHostToDevice, stream);
      if (e != hipSuccess) { rcs[t] = -(int)e; return; }
    }
  };
  std::vector<std::thread> helpers;
  for (int t = 1; t < threads; ++t) helpers.emplace_back(work, t);
  work(0);
  for (auto& h : helpers) h.join();
  for (int r : rcs) if (r) return r;
  return 0;
}
static int verify_host_overlapped(bn254_ctx* c, const uint8_t* msgs, const uint64_t* off, const uint8_t* sigs, const uint8_t* pks, size_t n,
                                  uint32_t flags, uint8_t* status, size_t msg_bytes) {
  int rc;
  if ((rc = ws_reserve(c, n))) return rc;
  for (int slot = 0; slot < 5; ++slot) {
    const size_t need[5] = {msg_bytes ? msg_bytes : 1, (n + 1) * sizeof(uint64_t), n * 64, n * 128, n};
    if ((rc = stage_reserve(c, slot, need[slot]))) return rc;
  }
  hipStream_t s = c->stream;
  const bool pinned = c->pinned_staging > 0 && n >= PINNED_STAGING_MIN_N;
  auto up = [](size_t x) { return (x + 4095) & ~(size_t)4095; };
  const size_t o_off = up(msg_bytes), o_sig = o_off + up((n + 1) * sizeof(uint64_t)), o_pk = o_sig + up(n * 64), pin_bytes = o_pk + up(n * 128);
  if (pinned && (rc = pin_reserve(c, pin_bytes))) return rc;
  PROF_MARK(0);
  if (pinned) {
    if ((rc = pinned_copy_in(c, c->stage[0], c->pin, msgs, msg_bytes, s, c->pinned_staging))) return rc;
    if ((rc = pinned_copy_in(c, c->stage[1], c->pin + o_off, (const uint8_t*)off, (n + 1) * sizeof(uint64_t), s, 1))) return rc;
  } else {
    if (msg_bytes) HIP_TRY(hipMemcpyAsync(c->stage[0], msgs, msg_bytes, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(c->stage[1], off, (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, s));
  }
  if ((rc = launch_hash_rounds(c, s, c->stage[0], (const uint64_t*)c->stage[1], n, PL_P2X, BY_P2_INF, nullptr))) return rc;
  if (pinned) {
    if ((rc = pinned_copy_in(c, c->stage[2], c->pin + o_sig, sigs, n * 64, c->copy_stream, c->pinned_staging))) return rc;
    if ((rc = pinned_copy_in(c, c->stage[3], c->pin + o_pk, pks, n * 128, c->copy_stream, c->pinned_staging))) return rc;
  } else {
    HIP_TRY(hipMemcpyAsync(c->stage[2], sigs, n * 64, hipMemcpyHostToDevice, c->copy_stream));
    HIP_TRY(hipMemcpyAsync(c->stage[3], pks, n * 128, hipMemcpyHostToDevice, c->copy_stream));
  }
  HIP_TRY(hipEventRecord(c->copy_done, c->copy_stream));
  HIP_TRY(hipStreamWaitEvent(s, c->copy_done, 0));
  PROF_MARK(1);
  k_decode_g1<<<grid_for(n), BN_WAVE, 0, s>>>(c->stage[2], n, flags, c->ws, PL_P1X, BY_P1_INF, 0);
  if ((rc = launch_decode_g2(c, s, c->stage[3], n, flags, 1))) return rc;
  PROF_MARK(2);
  if (c->pair_lanes) {
    if ((rc = launch_pair_or_trio(c, s, n, 1, c->stage[4], 0, true))) return rc;
  } else {
    k_miller_verify<<<grid_for(n), BN_WAVE, 0, s>>>(n, c->ws, nullptr, nullptr);
    PROF_MARK(3);
    k_final_exp<<<grid_for(n), BN_WAVE, 0, s>>>(n, 1, 1, 1, c->ws, 1, nullptr, c->stage[4], 0, 0, nullptr, nullptr);
  }
  PROF_MARK(4);
  if (c->profiling) { c->ev_valid = 1; c->ev_hash_first = 1; }   // intervals: transfer + hash, decode, Miller, final exp.
  HIP_TRY(hipGetLastError());
  if ((rc = stage_out(c, 4, status, n))) return rc;
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

// Host-buffer entry point.  The hash-to-G1 rounds need only the messages, so those cross PCIe first and the hash
// kernels start at once; signatures and public keys (5/6 of the bytes) follow on a second stream while the hash runs,
// and the decode kernels wait for them on an event.  What is left exposed of the transfer is the message copy and the
// status bytes coming back.
int bn254_batch_verify(bn254_ctx* c, const uint8_t* msgs, const uint64_t* off, const uint8_t* sigs, const uint8_t* pks, size_t n,
                       uint32_t flags, uint8_t* status) {
  MsgsLenScope msgs_len_scope(c);
  if (!c || (n && (!off || !sigs || !pks || !status))) return BN254_E_BAD_ARGUMENT;
  if (n == 0) return 0;
  HIP_TRY(hipSetDevice(c->device));
  if (!offsets_ok(off, n)) return BN254_E_BAD_ARGUMENT;
  size_t msg_bytes = (size_t)off[n];
  if (msg_bytes && !msgs) return BN254_E_BAD_ARGUMENT;
  int rc;
  bool split = c->split_miller && n <= BN_SPLIT_MAX_N;
  if (split) {                                       // A/B layout: plain staging, then the device entry point
    if ((rc = stage_in(c, 0, msgs, msg_bytes))) return rc;
    if ((rc = stage_in(c, 1, off, (n + 1) * sizeof(uint64_t)))) return rc;
    if ((rc = stage_in(c, 2, sigs, n * 64))) return rc;
    if ((rc = stage_in(c, 3, pks, n * 128))) return rc;
    if ((rc = stage_reserve(c, 4, n))) return rc;
    if ((rc = bn254_batch_verify_device(c, c->stage[0], (const uint64_t*)c->stage[1], c->stage[2], c->stage[3], n, flags, c->stage[4], nullptr))) return rc;
    if ((rc = stage_out(c, 4, status, n))) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
  }
  rc = verify_host_overlapped(c, msgs, off, sigs, pks, n, flags, status, msg_bytes);
  if (rc) {
    // a failure after the first asynchronous enqueue: the copies and kernels already in flight still read the caller's
    // buffers — wait for both streams before handing them back
    (void)hipStreamSynchronize(c->copy_stream);
    (void)hipStreamSynchronize(c->stream);
  }
  return rc;
}

// ---- keyed verify (include/bn254_hip.h) ---------------------------------------------------------------------------------
int bn254_ctx_register_keys(bn254_ctx* c, const uint8_t* pks, size_t n_keys, uint32_t flags, uint8_t* key_status) {
  if (!c || (n_keys && !pks) || n_keys > 0xFFFFFFFFu) return BN254_E_BAD_ARGUMENT;
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipDeviceSynchronize());                    // no keyed verify — on c->stream or on a caller's stream — may still be reading the previous tables
  c->n_keys = 0;
  if (n_keys == 0) return 0;
  if (n_keys > c->key_cap) {
    if (c->key_lines) { HIP_TRY(hipFree(c->key_lines)); c->key_lines = nullptr; }
    if (c->key_xy) { HIP_TRY(hipFree(c->key_xy)); c->key_xy = nullptr; }
    if (c->key_st) { HIP_TRY(hipFree(c->key_st)); c->key_st = nullptr; }
    if (c->key_inf) { HIP_TRY(hipFree(c->key_inf)); c->key_inf = nullptr; }
    c->key_cap = 0;
    HIP_TRY(hipMalloc((void**)&c->key_lines, n_keys * (size_t)BN_N_FIXED_LINES * BN_KEY_LINE_WORDS * sizeof(int32_t)));
    HIP_TRY(hipMalloc((void**)&c->key_xy, n_keys * 4 * BN_LIMBS * sizeof(int32_t)));
    HIP_TRY(hipMalloc((void**)&c->key_st, n_keys));
    HIP_TRY(hipMalloc((void**)&c->key_inf, n_keys));
    c->key_cap = n_keys;
  }
  int rc;
  if ((rc = stage_in(c, 3, pks, n_keys * 128))) return rc;
  k_register_keys<<<grid_for(n_keys), BN_WAVE, 0, c->stream>>>(c->stage[3], n_keys, flags & FLAG_REJECT_IDENTITY, c->key_lines, c->key_st, c->key_inf, c->key_xy);
  HIP_TRY(hipGetLastError());
  if (key_status) HIP_TRY(hipMemcpyAsync(key_status, c->key_st, n_keys, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->n_keys = n_keys;
  return 0;
}
int bn254_batch_verify_keyed_device(bn254_ctx* c, const uint8_t* d_msgs, const uint64_t* d_off, const uint8_t* d_sigs, const uint32_t* d_key_idx,
                                    size_t n, uint32_t flags, uint8_t* d_status, void* stream) {
  MsgsLenScope msgs_len_scope(c);
  if (!c || (n && (!d_msgs || !d_off || !d_sigs || !d_key_idx || !d_status))) return BN254_E_BAD_ARGUMENT;
  if (n == 0) return 0;
  if (misaligned(d_sigs) || misaligned(d_key_idx) || ((uintptr_t)d_off & 7u)) return BN254_E_MISALIGNED;
  HIP_TRY(hipSetDevice(c->device));
  int rc = ws_reserve(c, n);
  if (rc) return rc;
  hipStream_t s = stream ? (hipStream_t)stream : c->stream;
  KeyTable kt = {c->key_lines, c->key_st, c->key_inf, (uint32_t)c->n_keys};
  PROF_MARK(0);
  k_decode_g1<<<grid_for(n), BN_WAVE, 0, s>>>(d_sigs, n, flags, c->ws, PL_P1X, BY_P1_INF, 0);
  if (c->n_keys == 0 || !c->key_lines) {             // nothing registered: no table to read — every item is out of range
    k_keyed_no_keys<<<grid_for(n), BN_WAVE, 0, s>>>(n, c->ws, d_status);
    HIP_TRY(hipGetLastError());
    return 0;
  }
  PROF_MARK(1);
  if ((rc = launch_hash_rounds(c, s, d_msgs, d_off, n, PL_P2X, BY_P2_INF, nullptr))) return rc;
  PROF_MARK(2);
  if (c->pair_lanes && c->trio_max_batch > 0 && n <= (size_t)c->trio_max_batch) {
    // a batch that cannot fill the chip: latency counts — expand the keys and take the small-batch kernels (2.3 ms instead of the
    // 6 ms of the lane-pair layout; the line tables pay off only where throughput binds)
    k_keyed_expand<<<grid_for(n), BN_WAVE, 0, s>>>(n, c->ws, d_key_idx, kt, c->key_xy);
    if ((rc = launch_pair_or_trio(c, s, n, 1, d_status, 0, true))) return rc;
    PROF_MARK(4);
    if (c->profiling) { c->ev_valid = 1; c->ev_hash_first = 0; }
    HIP_TRY(hipGetLastError());
    return 0;
  }
  if ((rc = bn254_pair_miller_verify_keyed(n, c->ws, d_key_idx, kt, s))) return rc;
  PROF_MARK(3);
  if ((rc = bn254_pair_final_exp(n, c->ws, 1, d_status, nullptr, nullptr, s))) return rc;
  PROF_MARK(4);
  if (c->profiling) { c->ev_valid = 1; c->ev_hash_first = 0; }
  HIP_TRY(hipGetLastError());
  return 0;
}
int bn254_batch_verify_keyed(bn254_ctx* c, const uint8_t* msgs, const uint64_t* off, const uint8_t* sigs, const uint32_t* key_idx, size_t n,
                             uint32_t flags, uint8_t* status) {
  MsgsLenScope msgs_len_scope(c);
  if (!c || (n && (!off || !sigs || !key_idx || !status))) return BN254_E_BAD_ARGUMENT;
  if (n == 0) return 0;
  HIP_TRY(hipSetDevice(c->device));
  if (!offsets_ok(off, n)) return BN254_E_BAD_ARGUMENT;
  if (off[n] && !msgs) return BN254_E_BAD_ARGUMENT;
  int rc;
  if ((rc = stage_in(c, 0, msgs, (size_t)off[n]))) return rc;
  if ((rc = stage_in(c, 1, off, (n + 1) * sizeof(uint64_t)))) return rc;
  if ((rc = stage_in(c, 2, sigs, n * 64))) return rc;
  if ((rc = stage_in(c, 3, key_idx, n * sizeof(uint32_t)))) return rc;
  if ((rc = stage_reserve(c, 4, n))) return rc;
  rc = bn254_batch_verify_keyed_device(c, c->stage[0], (const uint64_t*)c->stage[1], c->stage[2], (const uint32_t*)c->stage[3], n, flags, c->stage[4], nullptr);
  if (!rc) rc = stage_out(c, 4, status, n);
  hipError_t e = hipStreamSynchronize(c->stream);     // also on failure: the staged copies read the caller's buffers
  return rc ? rc : -(int)e;
}

int bn254_batch_verify_keyed_randomized_device(bn254_ctx* c, const uint8_t* d_msgs, const uint64_t* d_off, const uint8_t* d_sigs,
                                               const uint32_t* d_key_idx, size_t n, uint32_t flags, const uint8_t* seed32, uint8_t* d_status,
                                               void* stream) {
  MsgsLenScope msgs_len_scope(c);
  if (!c || !seed32 || (n && (!d_msgs || !d_off || !d_sigs || !d_key_idx || !d_status))) return BN254_E_BAD_ARGUMENT;
  if (n == 0) return 0;
  if (n > 0xFFFFFFF0u) return BN254_E_BAD_ARGUMENT;
  if (misaligned(d_sigs) || misaligned(d_key_idx) || ((uintptr_t)d_off & 7u)) return BN254_E_MISALIGNED;
  const uint32_t dflags = flags & (BN254_FLAG_G2_SUBGROUP_CHECK | BN254_FLAG_REJECT_IDENTITY);
  if (c->n_keys == 0 || !c->key_lines || n < (size_t)c->rand_min_batch)      // nothing to group by / too small to pay off: the exact keyed path
    return bn254_batch_verify_keyed_device(c, d_msgs, d_off, d_sigs, d_key_idx, n, dflags, d_status, stream);
  HIP_TRY(hipSetDevice(c->device));
  const size_t K = c->n_keys;
  const size_t groups_max = n / BN_WAVE + (K < n ? K : n) + 1, slots_max = groups_max * BN_WAVE;
  const size_t gbase = (n + 255) & ~(size_t)255;
  int rc = ws_reserve(c, gbase + groups_max);
  if (rc) return rc;
  // scratch of this mode (device memory, grown on demand): [cnt K | start K | meta 2 | gkey groups_max | perm slots_max] words, group statuses
  const size_t words = 2 * K + 2 + groups_max + slots_max;
  if ((rc = stage_reserve(c, 5, words * sizeof(uint32_t)))) return rc;
  if ((rc = stage_reserve(c, 7, groups_max))) return rc;
  uint32_t* cnt = (uint32_t*)c->stage[5];
  uint32_t *start = cnt + K, *meta = start + K, *gkey = meta + 2, *perm = gkey + groups_max;
  uint8_t* d_group_st = c->stage[7];
  hipStream_t s = stream ? (hipStream_t)stream : c->stream;
  Seed seed;
  for (int j = 0; j < 8; ++j)
    seed.w[j] = ((uint32_t)seed32[4 * j] << 24) | ((uint32_t)seed32[4 * j + 1] << 16) | ((uint32_t)seed32[4 * j + 2] << 8) | seed32[4 * j + 3];
  KeyTable kt = {c->key_lines, c->key_st, c->key_inf, (uint32_t)c->n_keys};
  PROF_MARK(0);
  k_decode_g1<<<grid_for(n), BN_WAVE, 0, s>>>(d_sigs, n, dflags, c->ws, PL_P1X, BY_P1_INF, 0);
  PROF_MARK(1);
  if ((rc = launch_hash_rounds(c, s, d_msgs, d_off, n, PL_P2X, BY_P2_INF, nullptr))) return rc;
  PROF_MARK(2);
  HIP_TRY(hipMemsetAsync(cnt, 0, K * sizeof(uint32_t), s));
  HIP_TRY(hipMemsetAsync(perm, 0xFF, slots_max * sizeof(uint32_t), s));
  k_krand_prepare<<<grid_for(n), BN_WAVE, 0, s>>>(n, c->ws, d_key_idx, kt, cnt, d_status);
  k_krand_scan<<<1, BN_WAVE, 0, s>>>((uint32_t)K, cnt, start, gkey, meta);
  k_krand_scatter<<<grid_for(n), BN_WAVE, 0, s>>>(n, c->ws, d_key_idx, start, cnt, perm);
  k_krand_scale<<<(unsigned)groups_max, BN_WAVE, 0, s>>>(perm, meta, c->ws, seed, (flags & BN254_FLAG_RAND64) ? 1 : (flags & BN254_FLAG_RAND_GLV) ? 2 : 0, gbase);
  PROF_MARK(3);                                        // ms[2] = grouping + scalar multiplications, ms[3] = group checks + exact re-checks
  if ((rc = bn254_pair_miller_verify_keyed(groups_max, c->ws, gkey, kt, s, gbase, nullptr, meta))) return rc;
  if ((rc = bn254_pair_final_exp(groups_max, c->ws, 0, d_group_st, nullptr, meta, s, gbase))) return rc;
  k_krand_collect<<<grid_for(slots_max), BN_WAVE, 0, s>>>(slots_max, perm, meta, d_group_st, d_status, c->ws);
  // exact re-check of the items of failed groups (none queued: both kernels leave at once)
  if ((rc = bn254_pair_miller_verify_keyed(n, c->ws, d_key_idx, kt, s, 0, c->ws.h_list, c->ws.h_cnt))) return rc;
  if ((rc = bn254_pair_final_exp(n, c->ws, 1, d_status, c->ws.h_list, c->ws.h_cnt, s))) return rc;
  PROF_MARK(4);
  if (c->profiling) { c->ev_valid = 1; c->ev_hash_first = 0; }
  HIP_TRY(hipGetLastError());
  return 0;
}
int bn254_batch_verify_keyed_randomized(bn254_ctx* c, const uint8_t* msgs, const uint64_t* off, const uint8_t* sigs, const uint32_t* key_idx, size_t n,
                                        uint32_t flags, const uint8_t* seed32, uint8_t* status) {
  MsgsLenScope msgs_len_scope(c);
  if (!c || !seed32 || (n && (!off || !sigs || !key_idx || !status))) return BN254_E_BAD_ARGUMENT;
  if (n == 0) return 0;
  HIP_TRY(hipSetDevice(c->device));
  if (!offsets_ok(off, n)) return BN254_E_BAD_ARGUMENT;
  if (off[n] && !msgs) return BN254_E_BAD_ARGUMENT;
  int rc;
  if ((rc = stage_in(c, 0, msgs, (size_t)off[n]))) return rc;
  if ((rc = stage_in(c, 1, off, (n + 1) * sizeof(uint64_t)))) return rc;
  if ((rc = stage_in(c, 2, sigs, n * 64))) return rc;
  if ((rc = stage_in(c, 3, key_idx, n * sizeof(uint32_t)))) return rc;
  if ((rc = stage_reserve(c, 4, n))) return rc;
  rc = bn254_batch_verify_keyed_randomized_device(c, c->stage[0], (const uint64_t*)c->stage[1], c->stage[2], (const uint32_t*)c->stage[3], n, flags, seed32,
                                                  c->stage[4], nullptr);
  if (!rc) rc = stage_out(c, 4, status, n);
  hipError_t e = hipStreamSynchronize(c->stream);
  return rc ? rc : -(int)e;
}

int bn254_batch_verify_randomized_device(bn254_ctx* c, const uint8_t* d_msgs, const uint64_t* d_off, const uint8_t* d_sigs,
                                         const uint8_t* d_pks, size_t n, uint32_t flags, const uint8_t* seed32, uint8_t* d_status,
                                         uint8_t* d_group_ok, void* stream) {
  MsgsLenScope msgs_len_scope(c);
  if (!c || !seed32 || (n && (!d_msgs || !d_off || !d_sigs || !d_pks || !d_status))) return BN254_E_BAD_ARGUMENT;
  if (n == 0) return 0;
  if (n > 0xFFFFFFFFu) return BN254_E_BAD_ARGUMENT;
  if (misaligned(d_sigs) || misaligned(d_pks) || ((uintptr_t)d_off & 7u)) return BN254_E_MISALIGNED;
  HIP_TRY(hipSetDevice(c->device));
  const size_t n_groups = (n + BN_WAVE - 1) / BN_WAVE;
  if (n < (size_t)c->rand_min_batch) {
    // too small for the combined check to pay off (its per-group tail has the latency of a whole Miller loop + final
    // exponentiation): the exact kernels give the same statuses, faster
    int rc0 = bn254_batch_verify_device(c, d_msgs, d_off, d_sigs, d_pks, n, flags & (BN254_FLAG_G2_SUBGROUP_CHECK | BN254_FLAG_REJECT_IDENTITY),
                                        d_status, stream);
    if (rc0) return rc0;
    if (d_group_ok) {
      hipStream_t s0 = stream ? (hipStream_t)stream : c->stream;
      k_group_ok_from_status<<<grid_for(n_groups), BN_WAVE, 0, s0>>>(n_groups, n, d_status, d_group_ok);
      HIP_TRY(hipGetLastError());
    }
    return 0;
  }
  const size_t gbase = (n + 255) & ~(size_t)255;
  int rc = ws_reserve(c, gbase + n_groups);
  if (rc) return rc;
  hipStream_t s = stream ? (hipStream_t)stream : c->stream;
  Seed seed;
  for (int j = 0; j < 8; ++j)
    seed.w[j] = ((uint32_t)seed32[4 * j] << 24) | ((uint32_t)seed32[4 * j + 1] << 16) | ((uint32_t)seed32[4 * j + 2] << 8) | seed32[4 * j + 3];
  const unsigned g = grid_for(n), gg = grid_for(n_groups);
  const uint32_t dflags = flags & (BN254_FLAG_G2_SUBGROUP_CHECK | BN254_FLAG_REJECT_IDENTITY);
  uint8_t* d_group_st = c->ws.h_next;            // free once the hash rounds are done; n_groups <= stride
  PROF_MARK(0);
  k_decode_g1<<<g, BN_WAVE, 0, s>>>(d_sigs, n, dflags, c->ws, PL_P1X, BY_P1_INF, 0);
  if ((rc = launch_decode_g2(c, s, d_pks, n, dflags, 1))) return rc;
  PROF_MARK(1);
  if ((rc = launch_hash_rounds(c, s, d_msgs, d_off, n, PL_P2X, BY_P2_INF, nullptr))) return rc;
  PROF_MARK(2);
  k_rand_scale<<<g, BN_WAVE, 0, s>>>(n, c->ws, seed, (flags & BN254_FLAG_RAND64) ? 1 : (flags & BN254_FLAG_RAND_GLV) ? 2 : 0, gbase);
  const bool two = c->rand_items_per_lane ? c->rand_items_per_lane == 2 : n >= RAND_TWO_PER_LANE_MIN_N;
  if (c->pair_lanes) {
    if ((rc = bn254_pair_miller_rand(n, n_groups, two ? 2 : 1, c->ws, gbase, s))) return rc;
    PROF_MARK(3);
    if ((rc = bn254_pair_rand_tail(n_groups, c->ws, gbase, s))) return rc;
    if ((rc = bn254_pair_final_exp(n_groups, c->ws, 0, d_group_st, nullptr, nullptr, s, gbase))) return rc;
  } else {
    if (two) k_miller_rand2<<<(unsigned)((n_groups + 1) / 2), BN_WAVE, 0, s>>>(n, n_groups, c->ws, gbase);
    else k_miller_rand<<<g, BN_WAVE, 0, s>>>(n, c->ws, gbase);
    PROF_MARK(3);
    k_rand_tail<<<gg, BN_WAVE, 0, s>>>(n_groups, c->ws, gbase);
    k_final_exp<<<gg, BN_WAVE, 0, s>>>(n_groups, 1, 1, 1, c->ws, 0, nullptr, d_group_st, 0, gbase, nullptr, nullptr);
  }
  k_rand_collect<<<g, BN_WAVE, 0, s>>>(n, c->ws, d_group_st, d_status, d_group_ok);
  // exact per-item check of the items of failed groups (none queued: both kernels leave at once)
  if (c->pair_lanes) {
    if ((rc = bn254_pair_miller_verify(n, c->ws, c->ws.h_list, c->ws.h_cnt, s))) return rc;
    if ((rc = bn254_pair_final_exp(n, c->ws, 1, d_status, c->ws.h_list, c->ws.h_cnt, s))) return rc;
  } else {
    k_miller_verify<<<g, BN_WAVE, 0, s>>>(n, c->ws, c->ws.h_list, c->ws.h_cnt);
    k_final_exp<<<g, BN_WAVE, 0, s>>>(n, 1, 1, 1, c->ws, 1, nullptr, d_status, 0, 0, c->ws.h_list, c->ws.h_cnt);
  }
  PROF_MARK(4);
  if (c->profiling) { c->ev_valid = 1; c->ev_hash_first = 0; }
  HIP_TRY(hipGetLastError());
  return 0;
}

int bn254_batch_verify_randomized(bn254_ctx* c, const uint8_t* msgs, const uint64_t* off, const uint8_t* sigs, const uint8_t* pks, size_t n,
                                  uint32_t flags, const uint8_t* seed32, uint8_t* status, uint8_t* group_ok) {
  MsgsLenScope msgs_len_scope(c);
  if (!c || !seed32 || (n && (!off || !sigs || !pks || !status))) return BN254_E_BAD_ARGUMENT;
  if (n == 0) return 0;
  HIP_TRY(hipSetDevice(c->device));
  if (!offsets_ok(off, n)) return BN254_E_BAD_ARGUMENT;
  size_t msg_bytes = (size_t)off[n], n_groups = (n + BN_WAVE - 1) / BN_WAVE;
  int rc;
  if ((rc = stage_in(c, 0, msgs, msg_bytes))) return rc;
  if ((rc = stage_in(c, 1, off, (n + 1) * sizeof(uint64_t)))) return rc;
  if ((rc = stage_in(c, 2, sigs, n * 64))) return rc;
  if ((rc = stage_in(c, 3, pks, n * 128))) return rc;
  if ((rc = stage_reserve(c, 4, n))) return rc;
  if ((rc = stage_reserve(c, 5, n_groups))) return rc;
  if ((rc = bn254_batch_verify_randomized_device(c, c->stage[0], (const uint64_t*)c->stage[1], c->stage[2], c->stage[3], n, flags, seed32,
                                                 c->stage[4], c->stage[5], nullptr))) return rc;
  if ((rc = stage_out(c, 4, status, n))) return rc;
  if (group_ok && (rc = stage_out(c, 5, group_ok, n_groups))) return rc;
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

int bn254_batch_hash_to_g1_device(bn254_ctx* c, const uint8_t* d_msgs, const uint64_t* d_off, size_t n, uint8_t* d_points, uint8_t* d_status,
                                  uint8_t* d_tries, void* stream) {
  MsgsLenScope msgs_len_scope(c);
  if (!c || (n && (!d_msgs || !d_off || !d_points || !d_status))) return BN254_E_BAD_ARGUMENT;
  if (n == 0) return 0;
  if (misaligned(d_points) || ((uintptr_t)d_off & 7u)) return BN254_E_MISALIGNED;
  HIP_TRY(hipSetDevice(c->device));
  int rc = ws_reserve(c, n);
  if (rc) return rc;
  hipStream_t s = stream ? (hipStream_t)stream : c->stream;
  unsigned g = grid_for(n);
  PROF_MARK(0);                                        // ms[0] = the filter rounds (init / round / resolve), ms[1] = k_hash_finish (the square roots),
  if ((rc = launch_hash_rounds(c, s, d_msgs, d_off, n, PL_P1X, BY_P1_INF, d_tries, 1))) return rc;     // ms[2] = encoding the points, ms[3] = 0
  PROF_MARK(2);
  k_encode_g1<<<g, BN_WAVE, 0, s>>>(n, c->ws, PL_P1X, BY_P1_INF, d_points, d_status);
  PROF_MARK(3);
  PROF_MARK(4);
  if (c->profiling) { c->ev_valid = 1; c->ev_hash_first = 0; }
  HIP_TRY(hipGetLastError());
  return 0;
}
int bn254_batch_hash_to_g1(bn254_ctx* c, const uint8_t* msgs, const uint64_t* off, size_t n, uint8_t* points, uint8_t* status, uint8_t* tries) {
  MsgsLenScope msgs_len_scope(c);
  if (!c || (n && (!off || !points || !status))) return BN254_E_BAD_ARGUMENT;
  if (n == 0) return 0;
  HIP_TRY(hipSetDevice(c->device));
  int rc;
  if (!offsets_ok(off, n)) return BN254_E_BAD_ARGUMENT;
  if ((rc = stage_in(c, 0, msgs, (size_t)off[n]))) return rc;
  if ((rc = stage_in(c, 1, off, (n + 1) * sizeof(uint64_t)))) return rc;
  if ((rc = stage_reserve(c, 2, n * 64))) return rc;
  if ((rc = stage_reserve(c, 3, n))) return rc;
  if ((rc = stage_reserve(c, 4, n))) return rc;
  if ((rc = bn254_batch_hash_to_g1_device(c, c->stage[0], (const uint64_t*)c->stage[1], n, c->stage[2], c->stage[3], c->stage[4], nullptr))) return rc;
  if ((rc = stage_out(c, 2, points, n * 64))) return rc;
  if ((rc = stage_out(c, 3, status, n))) return rc;
  if (tries && (rc = stage_out(c, 4, tries, n))) return rc;
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

// shared by pairing / pairing_check: mode 0 = reduced Gt + status, 1 = raw Miller value (debug)
static int pairing_device(bn254_ctx* c, const uint8_t* d_g1, const uint8_t* d_g2, size_t n, size_t k, uint32_t flags, uint8_t* d_gt,
                          uint8_t* d_status, void* stream, int raw_only) {
  if (!c || k == 0 || (n && (!d_g1 || !d_g2))) return BN254_E_BAD_ARGUMENT;
  if (n == 0) return 0;
  if (misaligned(d_g1) || misaligned(d_g2) || (d_gt && misaligned(d_gt))) return BN254_E_MISALIGNED;
  HIP_TRY(hipSetDevice(c->device));
  size_t lanes = n * k;
  int rc = ws_reserve(c, lanes);
  if (rc) return rc;
  hipStream_t s = stream ? (hipStream_t)stream : c->stream;
  PROF_MARK(0);
  k_decode_g1<<<grid_for(lanes), BN_WAVE, 0, s>>>(d_g1, lanes, flags, c->ws, PL_P1X, BY_P1_INF, 0);
  if ((rc = launch_decode_g2(c, s, d_g2, lanes, flags, 1))) return rc;
  PROF_MARK(1);
  PROF_MARK(2);                                      // no hash in a pairing: ms[1] = 0
  if (c->pair_lanes) {
    if ((rc = bn254_pair_miller_var(lanes, c->ws, s))) return rc;
    PROF_MARK(3);
    if ((rc = bn254_pair_final_exp_product(n, k, c->ws, d_gt, d_status, raw_only, s))) return rc;
  } else {
    k_miller_var<<<grid_for(lanes), BN_WAVE, 0, s>>>(lanes, c->ws);
    PROF_MARK(3);
    k_final_exp<<<grid_for(n), BN_WAVE, 0, s>>>(n, k, k, 1, c->ws, 0, d_gt, d_status, raw_only, 0, nullptr, nullptr);
  }
  PROF_MARK(4);
  if (c->profiling) { c->ev_valid = 1; c->ev_hash_first = 0; }
  HIP_TRY(hipGetLastError());
  return 0;
}
int bn254_batch_pairing_device(bn254_ctx* c, const uint8_t* d_g1, const uint8_t* d_g2, size_t n, size_t k, uint32_t flags, uint8_t* d_gt,
                               uint8_t* d_status, void* stream) {
  return pairing_device(c, d_g1, d_g2, n, k, flags, d_gt, d_status, stream, 0);
}
static int pairing_host(bn254_ctx* c, const uint8_t* g1, const uint8_t* g2, size_t n, size_t k, uint32_t flags, uint8_t* gt, uint8_t* status,
                        int raw_only) {
  if (!c || k == 0 || (n && (!g1 || !g2))) return BN254_E_BAD_ARGUMENT;
  if (n == 0) return 0;
  HIP_TRY(hipSetDevice(c->device));
  int rc;
  if ((rc = stage_in(c, 0, g1, n * k * 64))) return rc;
  if ((rc = stage_in(c, 1, g2, n * k * 128))) return rc;
  if ((rc = stage_reserve(c, 2, n * 384))) return rc;
  if ((rc = stage_reserve(c, 3, n))) return rc;
  if ((rc = pairing_device(c, c->stage[0], c->stage[1], n, k, flags, gt ? c->stage[2] : nullptr, c->stage[3], nullptr, raw_only))) return rc;
  if (gt && (rc = stage_out(c, 2, gt, n * 384))) return rc;
  if (status && (rc = stage_out(c, 3, status, n))) return rc;
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}
int bn254_batch_pairing_check(bn254_ctx* c, const uint8_t* g1, const uint8_t* g2, size_t n, size_t k, uint32_t flags, uint8_t* status) {
  if (!status && n) return BN254_E_BAD_ARGUMENT;
  return pairing_host(c, g1, g2, n, k, flags, nullptr, status, 0);
}
int bn254_batch_pairing(bn254_ctx* c, const uint8_t* g1, const uint8_t* g2, size_t n, size_t k, uint32_t flags, uint8_t* gt, uint8_t* status) {
  if (!gt && n) return BN254_E_BAD_ARGUMENT;
  return pairing_host(c, g1, g2, n, k, flags, gt, status, 0);
}
int bn254_debug_miller_loop(bn254_ctx* c, const uint8_t* g1, const uint8_t* g2, size_t n, uint8_t* f) {
  if (!f && n) return BN254_E_BAD_ARGUMENT;
  return pairing_host(c, g1, g2, n, 1, 0, f, nullptr, 1);
}

int bn254_batch_check_public_keys(bn254_ctx* c, const uint8_t* pk_g2, const uint8_t* pk_g1, size_t n, uint32_t flags, uint8_t* status) {
  if (!c || (n && (!pk_g2 || !pk_g1 || !status))) return BN254_E_BAD_ARGUMENT;
  if (n == 0) return 0;
  HIP_TRY(hipSetDevice(c->device));
  int rc;
  if ((rc = ws_reserve(c, n))) return rc;
  if ((rc = stage_in(c, 0, pk_g2, n * 128))) return rc;
  if ((rc = stage_in(c, 1, pk_g1, n * 64))) return rc;
  if ((rc = stage_reserve(c, 2, n))) return rc;
  hipStream_t s = c->stream;
  unsigned g = grid_for(n);
  if ((rc = launch_decode_g2(c, s, c->stage[0], n, flags, 0))) return rc;       // ecdsa.rs:82: pk_g2 first
  k_decode_g1<<<g, BN_WAVE, 0, s>>>(c->stage[1], n, flags, c->ws, PL_P1X, BY_P1_INF, 1);
  if (c->pair_lanes) {
    if ((rc = launch_pair_or_trio(c, s, n, 0, c->stage[2], 1, false))) return rc;
  } else {
    k_miller_cpk<<<g, BN_WAVE, 0, s>>>(n, c->ws);
    k_final_exp<<<g, BN_WAVE, 0, s>>>(n, 1, 1, 1, c->ws, 0, nullptr, c->stage[2], 0, 0, nullptr, nullptr);
  }
  HIP_TRY(hipGetLastError());
  if ((rc = stage_out(c, 2, status, n))) return rc;
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

// ---- group operations --------------------------------------------------------------------
static int binop_host(bn254_ctx* c, int g2, const uint8_t* a, const uint8_t* b, size_t n, uint8_t* out, uint8_t* status) {
  if (!c || (n && (!a || !b || !out || !status))) return BN254_E_BAD_ARGUMENT;
  if (n == 0) return 0;
  HIP_TRY(hipSetDevice(c->device));
  size_t sz = g2 ? 128 : 64;
  int rc;
  if ((rc = stage_in(c, 0, a, n * sz))) return rc;
  if ((rc = stage_in(c, 1, b, n * sz))) return rc;
  if ((rc = stage_reserve(c, 2, n * sz))) return rc;
  if ((rc = stage_reserve(c, 3, n))) return rc;
  if (g2) k_g2_add<<<grid_for(n), BN_WAVE, 0, c->stream>>>(c->stage[0], c->stage[1], n, c->stage[2], c->stage[3]);
  else k_g1_add<<<grid_for(n), BN_WAVE, 0, c->stream>>>(c->stage[0], c->stage[1], n, c->stage[2], c->stage[3]);
  HIP_TRY(hipGetLastError());
  if ((rc = stage_out(c, 2, out, n * sz))) return rc;
  if ((rc = stage_out(c, 3, status, n))) return rc;
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}
int bn254_batch_g1_add(bn254_ctx* c, const uint8_t* a, const uint8_t* b, size_t n, uint8_t* out, uint8_t* status) { return binop_host(c, 0, a, b, n, out, status); }
int bn254_batch_g2_add(bn254_ctx* c, const uint8_t* a, const uint8_t* b, size_t n, uint8_t* out, uint8_t* status) { return binop_host(c, 1, a, b, n, out, status); }

int bn254_batch_g1_mul_device(bn254_ctx* c, const uint8_t* d_p, const uint8_t* d_k, size_t n, int reduce, uint8_t* d_out, uint8_t* d_status, void* stream) {
  if (!c || (n && (!d_p || !d_k || !d_out || !d_status))) return BN254_E_BAD_ARGUMENT;
  if (n == 0) return 0;
  if (misaligned(d_p) || misaligned(d_k) || misaligned(d_out)) return BN254_E_MISALIGNED;
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t s = stream ? (hipStream_t)stream : c->stream;
  k_g1_mul<<<grid_for(n), BN_WAVE, 0, s>>>(d_p, d_k, n, reduce, c->ws, d_out, d_status);
  HIP_TRY(hipGetLastError());
  return 0;
}
int bn254_batch_g2_mul_device(bn254_ctx* c, const uint8_t* d_p, const uint8_t* d_k, size_t n, int reduce, uint8_t* d_out, uint8_t* d_status, void* stream) {
  if (!c || (n && (!d_k || !d_out || !d_status))) return BN254_E_BAD_ARGUMENT;   // d_p == NULL: generator
  if (n == 0) return 0;
  if ((d_p && misaligned(d_p)) || misaligned(d_k) || misaligned(d_out)) return BN254_E_MISALIGNED;
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t s = stream ? (hipStream_t)stream : c->stream;
  k_g2_mul<<<grid_for(n), BN_WAVE, 0, s>>>(d_p, d_k, n, reduce, d_out, d_status);
  HIP_TRY(hipGetLastError());
  return 0;
}
static int mul_host(bn254_ctx* c, int g2, const uint8_t* p, const uint8_t* k, size_t n, int reduce, uint8_t* out, uint8_t* status) {
  if (!c || (n && (!k || !out || !status)) || (!g2 && n && !p)) return BN254_E_BAD_ARGUMENT;
  if (n == 0) return 0;
  HIP_TRY(hipSetDevice(c->device));
  size_t sz = g2 ? 128 : 64;
  int rc;
  if (p && (rc = stage_in(c, 0, p, n * sz))) return rc;
  if ((rc = stage_in(c, 1, k, n * 32))) return rc;
  if ((rc = stage_reserve(c, 2, n * sz))) return rc;
  if ((rc = stage_reserve(c, 3, n))) return rc;
  rc = g2 ? bn254_batch_g2_mul_device(c, p ? c->stage[0] : nullptr, c->stage[1], n, reduce, c->stage[2], c->stage[3], nullptr)
          : bn254_batch_g1_mul_device(c, c->stage[0], c->stage[1], n, reduce, c->stage[2], c->stage[3], nullptr);
  if (rc) return rc;
  if ((rc = stage_out(c, 2, out, n * sz))) return rc;
  if ((rc = stage_out(c, 3, status, n))) return rc;
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}
int bn254_batch_g1_mul(bn254_ctx* c, const uint8_t* p, const uint8_t* k, size_t n, int reduce, uint8_t* out, uint8_t* status) { return mul_host(c, 0, p, k, n, reduce, out, status); }
int bn254_batch_g2_mul(bn254_ctx* c, const uint8_t* p, const uint8_t* k, size_t n, int reduce, uint8_t* out, uint8_t* status) { return mul_host(c, 1, p, k, n, reduce, out, status); }

int bn254_batch_sign_device(bn254_ctx* c, const uint8_t* d_msgs, const uint64_t* d_off, const uint8_t* d_sks, size_t n, uint8_t* d_sigs,
                            uint8_t* d_status, void* stream) {
  MsgsLenScope msgs_len_scope(c);
  if (!c || (n && (!d_msgs || !d_off || !d_sks || !d_sigs || !d_status))) return BN254_E_BAD_ARGUMENT;
  if (n == 0) return 0;
  if (misaligned(d_sks) || misaligned(d_sigs) || ((uintptr_t)d_off & 7u)) return BN254_E_MISALIGNED;
  HIP_TRY(hipSetDevice(c->device));
  int rc = ws_reserve(c, n);
  if (rc) return rc;
  hipStream_t s = stream ? (hipStream_t)stream : c->stream;
  if ((rc = launch_hash_rounds(c, s, d_msgs, d_off, n, PL_P1X, BY_P1_INF, nullptr))) return rc;             // ecdsa.rs:28
  k_g1_mul<<<grid_for(n), BN_WAVE, 0, s>>>(nullptr, d_sks, n, 1, c->ws, d_sigs, d_status);              // ecdsa.rs:31
  HIP_TRY(hipGetLastError());
  return 0;
}
int bn254_batch_sign(bn254_ctx* c, const uint8_t* msgs, const uint64_t* off, const uint8_t* sks, size_t n, uint8_t* sigs, uint8_t* status) {
  MsgsLenScope msgs_len_scope(c);
  if (!c || (n && (!off || !sks || !sigs || !status))) return BN254_E_BAD_ARGUMENT;
  if (n == 0) return 0;
  HIP_TRY(hipSetDevice(c->device));
  int rc;
  if (!offsets_ok(off, n)) return BN254_E_BAD_ARGUMENT;
  if ((rc = stage_in(c, 0, msgs, (size_t)off[n]))) return rc;
  if ((rc = stage_in(c, 1, off, (n + 1) * sizeof(uint64_t)))) return rc;
  if ((rc = stage_in(c, 2, sks, n * 32))) return rc;
  if ((rc = stage_reserve(c, 3, n * 64))) return rc;
  if ((rc = stage_reserve(c, 4, n))) return rc;
  if ((rc = bn254_batch_sign_device(c, c->stage[0], (const uint64_t*)c->stage[1], c->stage[2], n, c->stage[3], c->stage[4], nullptr))) return rc;
  if ((rc = stage_out(c, 3, sigs, n * 64))) return rc;
  if ((rc = stage_out(c, 4, status, n))) return rc;
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

static int sum_host(bn254_ctx* c, int g2, const uint8_t* pts, const uint64_t* seg, size_t n, uint8_t* out, uint8_t* status) {
  if (!c || (n && (!seg || !out || !status))) return BN254_E_BAD_ARGUMENT;
  if (n == 0) return 0;
  HIP_TRY(hipSetDevice(c->device));
  size_t sz = g2 ? 128 : 64;
  if (!offsets_ok(seg, n)) return BN254_E_BAD_ARGUMENT;
  size_t total = (size_t)seg[n];
  int rc;
  if ((rc = stage_in(c, 0, pts, total * sz))) return rc;
  if ((rc = stage_in(c, 1, seg, (n + 1) * sizeof(uint64_t)))) return rc;
  if ((rc = stage_reserve(c, 2, n * sz))) return rc;
  if ((rc = stage_reserve(c, 3, n))) return rc;
  if (g2) k_g2_sum<<<grid_for(n), BN_WAVE, 0, c->stream>>>(c->stage[0], (const uint64_t*)c->stage[1], n, c->stage[2], c->stage[3]);
  else k_g1_sum<<<grid_for(n), BN_WAVE, 0, c->stream>>>(c->stage[0], (const uint64_t*)c->stage[1], n, c->stage[2], c->stage[3]);
  HIP_TRY(hipGetLastError());
  if ((rc = stage_out(c, 2, out, n * sz))) return rc;
  if ((rc = stage_out(c, 3, status, n))) return rc;
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}
int bn254_batch_g1_sum(bn254_ctx* c, const uint8_t* pts, const uint64_t* seg, size_t n, uint8_t* out, uint8_t* status) { return sum_host(c, 0, pts, seg, n, out, status); }
int bn254_batch_g2_sum(bn254_ctx* c, const uint8_t* pts, const uint64_t* seg, size_t n, uint8_t* out, uint8_t* status) { return sum_host(c, 1, pts, seg, n, out, status); }

int bn254_batch_aggregate_verify_device(bn254_ctx* c, const uint8_t* d_msgs, const uint64_t* d_msg_off, size_t n_msgs, const uint8_t* d_pk_pool,
                                        size_t n_signers, const uint8_t* d_sig_pool, const uint32_t* d_tuple_msg, const uint64_t* d_tuple_off,
                                        const uint32_t* d_signer_idx, size_t n, uint32_t flags, uint8_t* d_status, void* stream) {
  MsgsLenScope msgs_len_scope(c);
  if (!c || !n_msgs || !n_signers || (n && (!d_msgs || !d_msg_off || !d_pk_pool || !d_sig_pool || !d_tuple_msg || !d_tuple_off || !d_signer_idx || !d_status)))
    return BN254_E_BAD_ARGUMENT;
  if (n == 0) return 0;
  if (misaligned(d_pk_pool) || misaligned(d_sig_pool) || misaligned(d_tuple_msg) || misaligned(d_signer_idx) || ((uintptr_t)d_msg_off & 7u) ||
      ((uintptr_t)d_tuple_off & 7u))
    return BN254_E_MISALIGNED;
  HIP_TRY(hipSetDevice(c->device));
  int rc;
  if ((rc = ws_reserve(c, n > n_msgs ? n : n_msgs))) return rc;
  if ((rc = pool_reserve(c, 0, 4, n_signers))) return rc;
  if ((rc = pool_reserve(c, 1, 2, n_msgs * n_signers))) return rc;
  if ((rc = pool_reserve(c, 2, 2, n_msgs))) return rc;
  hipStream_t s = stream ? (hipStream_t)stream : c->stream;
  PROF_MARK(0);                                        // ms[0] = pools (decode, hash of the messages, subset-sum table), ms[1] = the aggregation kernel
  k_pool_decode_g2<<<grid_for(n_signers), BN_WAVE, 0, s>>>(d_pk_pool, n_signers, flags, c->pool[0]);
  k_pool_decode_g1<<<grid_for(n_msgs * n_signers), BN_WAVE, 0, s>>>(d_sig_pool, n_msgs * n_signers, flags, c->pool[1]);
  if ((rc = launch_hash_rounds(c, s, d_msgs, d_msg_off, n_msgs, PL_P2X, BY_P2_INF, nullptr))) return rc;
  k_hash_to_pool<<<grid_for(n_msgs), BN_WAVE, 0, s>>>(n_msgs, c->ws, c->pool[2]);
  if (c->pair_lanes) {
    // subset sums of the key pool for batches large enough to repay the table (n_groups x 256 entries of ~4 additions + an
    // inversion each); the kernel uses it for the waves whose longest signer list has more entries than there are groups
    size_t n_groups = 0, groups4 = 0;
    if (c->agg_subset_min_tuples > 0 && n >= (size_t)c->agg_subset_min_tuples && n_signers <= AGG_SUBSET_MAX_SIGNERS) {
      n_groups = (n_signers + 7) / 8;
      if ((rc = pool_reserve(c, 3, 4, n_groups * 256))) return rc;
      k_pool_subsets_g2<<<grid_for(n_groups * 256), BN_WAVE, 0, s>>>(c->pool[0], n_signers, n_groups, c->pool[3]);
      // the signature tables are per message: worth it when a message's table (4 n_signers entries of ~2 additions + an inversion)
      // is shared by enough tuples, and only while it fits a budget of HBM
      const size_t entries = n_msgs * 2 * n_groups * 16;
      // priced at what pool_reserve allocates per entry (a record of BN_POOL_HALF_WORDS words + its status byte, entries rounded up to 256)
      const size_t table_bytes = ((entries + 255) & ~(size_t)255) * (BN_POOL_HALF_WORDS * sizeof(int32_t) + 1);
      if (n >= AGG_SUBSET_G1_TUPLES_PER_MSG * n_msgs && table_bytes <= AGG_SUBSET_G1_MAX_BYTES) {
        if (pool_reserve(c, 4, 2, entries) == 0) {
          groups4 = 2 * n_groups;
          k_pool_subsets_g1<<<grid_for(entries), BN_WAVE, 0, s>>>(c->pool[1], n_signers, groups4, n_msgs, c->pool[4]);
        } else {
          (void)hipGetLastError();     // no HBM for the table: the signatures are added one by one (groups4 = 0), same statuses
        }
      }
    }
    // with the per-message signature tables in use: bucket the tuples by message (see k_agg_sort_count).  The hash rounds of the
    // messages are done with ws.h_list (2 x stride words): its first n words take the index map, the counters sit behind.
    const uint32_t* perm = nullptr;
    if (groups4 != 0 && c->agg_sort_by_msg && n >= 4 * n_msgs && n <= 0xFFFFFFFFull && n_msgs < 0xFFFFFFFFull && c->ws.stride >= 2 * (n_msgs + 1)) {
      uint32_t* map = c->ws.h_list;
      uint32_t* cnt = c->ws.h_list + c->ws.stride;
      uint32_t* cursor = cnt + (n_msgs + 1);
      HIP_TRY(hipMemsetAsync(cnt, 0, sizeof(uint32_t) * (n_msgs + 1), s));
      k_agg_sort_count<<<grid_for(n), BN_WAVE, 0, s>>>(d_tuple_msg, n, (uint32_t)n_msgs, cnt);
      k_agg_sort_scan<<<1, BN_WAVE, 0, s>>>((uint32_t)n_msgs + 1, cnt, cursor);
      k_agg_sort_scatter<<<grid_for(n), BN_WAVE, 0, s>>>(d_tuple_msg, n, (uint32_t)n_msgs, cursor, map);
      perm = map;
    }
    // the largest batches: tables of twice the window, built from the ones above by one batched affine addition per entry
    // (k_pool_widen_*): half the additions per tuple.  A table that does not fit its budget (or HBM) is simply not used.
    const Pool* wide2 = nullptr;
    const Pool* wide1 = nullptr;
    if (n_groups != 0 && c->agg_wide_min_tuples > 0 && n >= (size_t)c->agg_wide_min_tuples) {
      const size_t n_chunks = (n_groups + 1) / 2;
      const size_t e2 = n_chunks * 65536, bytes2 = e2 * (2 * BN_POOL_HALF_WORDS * sizeof(int32_t) + 1);
      if (bytes2 <= AGG_WIDE_G2_MAX_BYTES) {
        if (pool_reserve(c, 5, 4, e2) == 0) {
          k_pool_widen_g2<<<grid_for(n_chunks * 256 * (256 / BN_WIDEN_G2_NLO)), BN_WAVE, 0, s>>>(c->pool[3], n_groups, n_chunks, c->pool[5]);
          wide2 = &c->pool[5];
        } else {
          (void)hipGetLastError();
        }
      }
      const size_t e1 = n_msgs * n_groups * 256, bytes1 = ((e1 + 255) & ~(size_t)255) * (BN_POOL_HALF_WORDS * sizeof(int32_t) + 1);
      if (groups4 != 0 && n >= AGG_WIDE_G1_TUPLES_PER_MSG * n_msgs && bytes1 <= AGG_SUBSET_G1_MAX_BYTES) {
        if (pool_reserve(c, 6, 2, e1) == 0) {
          k_pool_widen_g1<<<grid_for(n_msgs * n_groups * 16), BN_WAVE, 0, s>>>(c->pool[4], groups4, n_groups, n_msgs, c->pool[6]);
          wide1 = &c->pool[6];
        } else {
          (void)hipGetLastError();
        }
      }
    }
    PROF_MARK(1);
    if ((rc = bn254_pair_aggregate(d_tuple_msg, d_tuple_off, d_signer_idx, n, n_signers, n_msgs, c->pool[0], c->pool[1], c->pool[2], c->pool[3], n_groups,
                                   c->pool[4], groups4, c->ws, s, perm, wide2, wide1))) return rc;
  } else {
    PROF_MARK(1);
    k_aggregate<<<grid_for(n), BN_WAVE, 0, s>>>(d_tuple_msg, d_tuple_off, d_signer_idx, n, n_signers, n_msgs, c->pool[0], c->pool[1], c->pool[2], c->ws);
  }
  PROF_MARK(2);
  if (c->pair_lanes) {
    if ((rc = bn254_pair_miller_verify(n, c->ws, nullptr, nullptr, s))) return rc;
    PROF_MARK(3);
    if ((rc = bn254_pair_final_exp(n, c->ws, 1, d_status, nullptr, nullptr, s))) return rc;
  } else {
    k_miller_verify<<<grid_for(n), BN_WAVE, 0, s>>>(n, c->ws, nullptr, nullptr);
    PROF_MARK(3);
    k_final_exp<<<grid_for(n), BN_WAVE, 0, s>>>(n, 1, 1, 1, c->ws, 1, nullptr, d_status, 0, 0, nullptr, nullptr);
  }
  PROF_MARK(4);
  if (c->profiling) { c->ev_valid = 1; c->ev_hash_first = 0; }
  HIP_TRY(hipGetLastError());
  return 0;
}
int bn254_batch_aggregate_verify(bn254_ctx* c, const uint8_t* msgs, const uint64_t* msg_off, size_t n_msgs, const uint8_t* pk_pool, size_t n_signers,
                                 const uint8_t* sig_pool, const uint32_t* tuple_msg, const uint64_t* tuple_off, const uint32_t* signer_idx, size_t n,
                                 uint32_t flags, uint8_t* status) {
  MsgsLenScope msgs_len_scope(c);
  if (!c || !n_msgs || !n_signers || (n && (!msg_off || !pk_pool || !sig_pool || !tuple_msg || !tuple_off || !signer_idx || !status))) return BN254_E_BAD_ARGUMENT;
  if (n == 0) return 0;
  if (!offsets_ok(tuple_off, n) || !offsets_ok(msg_off, n_msgs)) return BN254_E_BAD_ARGUMENT;
  HIP_TRY(hipSetDevice(c->device));
  int rc;
  if ((rc = stage_in(c, 0, msgs, (size_t)msg_off[n_msgs]))) return rc;
  if ((rc = stage_in(c, 1, msg_off, (n_msgs + 1) * sizeof(uint64_t)))) return rc;
  if ((rc = stage_in(c, 2, pk_pool, n_signers * 128))) return rc;
  if ((rc = stage_in(c, 3, sig_pool, n_msgs * n_signers * 64))) return rc;
  if ((rc = stage_in(c, 4, tuple_msg, n * sizeof(uint32_t)))) return rc;
  if ((rc = stage_in(c, 5, tuple_off, (n + 1) * sizeof(uint64_t)))) return rc;
  if ((rc = stage_in(c, 6, signer_idx, (size_t)tuple_off[n] * sizeof(uint32_t)))) return rc;
  if ((rc = stage_reserve(c, 7, n))) return rc;
  if ((rc = bn254_batch_aggregate_verify_device(c, c->stage[0], (const uint64_t*)c->stage[1], n_msgs, c->stage[2], n_signers, c->stage[3],
                                                (const uint32_t*)c->stage[4], (const uint64_t*)c->stage[5], (const uint32_t*)c->stage[6], n, flags,
                                                c->stage[7], nullptr))) return rc;
  if ((rc = stage_out(c, 7, status, n))) return rc;
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

static int decompress_host(bn254_ctx* c, int g2, const uint8_t* in, size_t n, uint8_t* out, uint8_t* status) {
  if (!c || (n && (!in || !out || !status))) return BN254_E_BAD_ARGUMENT;
  if (n == 0) return 0;
  HIP_TRY(hipSetDevice(c->device));
  size_t isz = g2 ? 65 : 33, osz = g2 ? 128 : 64;
  int rc;
  if ((rc = stage_in(c, 0, in, n * isz))) return rc;
  if ((rc = stage_reserve(c, 2, n * osz))) return rc;
  if ((rc = stage_reserve(c, 3, n))) return rc;
  if (g2) k_g2_decompress<<<grid_for(n), BN_WAVE, 0, c->stream>>>(c->stage[0], n, c->stage[2], c->stage[3]);
  else k_g1_decompress<<<grid_for(n), BN_WAVE, 0, c->stream>>>(c->stage[0], n, c->stage[2], c->stage[3]);
  HIP_TRY(hipGetLastError());
  if ((rc = stage_out(c, 2, out, n * osz))) return rc;
  if ((rc = stage_out(c, 3, status, n))) return rc;
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}
int bn254_batch_g1_decompress(bn254_ctx* c, const uint8_t* in, size_t n, uint8_t* out, uint8_t* status) { return decompress_host(c, 0, in, n, out, status); }
int bn254_batch_g2_decompress(bn254_ctx* c, const uint8_t* in, size_t n, uint8_t* out, uint8_t* status) { return decompress_host(c, 1, in, n, out, status); }

// issue-rate probe: wave-instructions per second of `op` with `waves_per_simd` waves on every SIMD, timed with HIP events
int bn254_probe_issue_rate(bn254_ctx* c, int op, int waves_per_simd, double* wave_inst_per_s, int* n_simd) {
  if (!c || !wave_inst_per_s || op < 0 || op > 2 || waves_per_simd < 1 || waves_per_simd > 8) return BN254_E_BAD_ARGUMENT;
  HIP_TRY(hipSetDevice(c->device));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, c->device));
  const int n_cu = prop.multiProcessorCount, blocks = n_cu * waves_per_simd;
  int rc;
  if ((rc = stage_reserve(c, 0, sizeof(uint32_t) * 256 * (size_t)blocks))) return rc;
  uint32_t* out = (uint32_t*)c->stage[0];
  ScopedEvents ev;                                   // destroyed on every path out, the early error returns included
  HIP_TRY(ev.create());
  hipEvent_t e0 = ev.e0, e1 = ev.e1;
  float best = 0;
  for (int rep = 0; rep < 3; ++rep) {       // first repetition warms up; keep the fastest
    HIP_TRY(hipEventRecord(e0, c->stream));
    if (op == 0) k_issue_probe<0><<<blocks, 256, 0, c->stream>>>(out, 12345u, c->ws.clk);
    else if (op == 1) k_issue_probe<1><<<blocks, 256, 0, c->stream>>>(out, 12345u, c->ws.clk);
    else k_issue_probe<2><<<blocks, 256, 0, c->stream>>>(out, 12345u, c->ws.clk);
    HIP_TRY(hipEventRecord(e1, c->stream));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    if (rep > 0 && (best == 0 || ms < best)) best = ms;
  }
  *wave_inst_per_s = (double)PROBE_ITERS * PROBE_CHAINS * 4.0 * blocks / (best * 1e-3);
  if (n_simd) *n_simd = n_cu * 4;
  return 0;
}

// Measurement: the product leaves of one verify's Miller loop alone (k_leaf_floor_pair, bn254_pair.hip) on the planes the last verify
// left in the workspace (n <= the size of that batch); ms = the kernel's duration (HIP events), best of 3 after a warm-up launch.
int bn254_probe_leaf_floor(bn254_ctx* c, size_t n, int mode, float* ms) {
  if (!c || !ms || n == 0 || n > c->ws.stride || mode < 0 || mode > 7) return BN254_E_BAD_ARGUMENT;
  HIP_TRY(hipSetDevice(c->device));
  ScopedEvents ev;                                   // destroyed on every path out, the early error returns included
  HIP_TRY(ev.create());
  hipEvent_t e0 = ev.e0, e1 = ev.e1;
  float best = 0;
  int rc = 0;
  for (int rep = 0; rep < 4 && rc == 0; ++rep) {
    HIP_TRY(hipEventRecord(e0, c->stream));
    rc = bn254_pair_leaf_floor(n, c->ws, c->stream, mode);
    HIP_TRY(hipEventRecord(e1, c->stream));
    HIP_TRY(hipEventSynchronize(e1));
    float t = 0;
    HIP_TRY(hipEventElapsedTime(&t, e0, e1));
    if (rep > 0 && (best == 0 || t < best)) best = t;
  }
  *ms = best;
  return rc;
}

// Measurement: the final exponentiation's accumulator machine on a caller-supplied program (pairs of bytes (opcode, argument), ended by
// (0, 0); opcodes 1 LOAD s, 2 STORE s, 3 CSQR, 4 MUL s, 5 CONJ, 6 FROB k, 7 INV — bn254_pairing.h) for n lane pairs, on whatever the
// F planes of the workspace hold (run a verify first).  ms = the kernel's duration, best of 3 after a warm-up launch.  The values are
// meaningless (a cyclotomic squaring of a non-cyclotomic element): this times the routines in place, it does not check them.
int bn254_probe_fe_program(bn254_ctx* c, size_t n, const uint8_t* prog, size_t n_steps, float* ms) {
  if (!c || !ms || !prog || n == 0 || n > c->ws.stride || n_steps == 0 || n_steps > 4096) return BN254_E_BAD_ARGUMENT;
  for (size_t k = 0; k < n_steps; ++k) {
    const uint8_t op = prog[2 * k], arg = prog[2 * k + 1];
    if (op == 0 || op > 7) return BN254_E_BAD_ARGUMENT;
    if ((op == 1 || op == 2 || op == 4) && arg >= (BN_FE_EXACT_SLOTS > BN_FE_CHECK_SLOTS ? BN_FE_EXACT_SLOTS : BN_FE_CHECK_SLOTS)) return BN254_E_BAD_ARGUMENT;
    if (op == 6 && (arg < 1 || arg > 3)) return BN254_E_BAD_ARGUMENT;
  }
  HIP_TRY(hipSetDevice(c->device));
  int rc;
  if ((rc = stage_reserve(c, 7, 2 * n_steps + 2))) return rc;
  std::vector<uint8_t> buf(prog, prog + 2 * n_steps);
  buf.push_back(0); buf.push_back(0);
  HIP_TRY(hipMemcpy(c->stage[7], buf.data(), buf.size(), hipMemcpyHostToDevice));
  ScopedEvents ev;                                   // destroyed on every path out, the early error returns included
  HIP_TRY(ev.create());
  hipEvent_t e0 = ev.e0, e1 = ev.e1;
  float best = 0;
  for (int rep = 0; rep < 4 && rc == 0; ++rep) {
    HIP_TRY(hipEventRecord(e0, c->stream));
    rc = bn254_pair_fe_program(n, c->ws, c->stage[7], c->stream);
    HIP_TRY(hipEventRecord(e1, c->stream));
    HIP_TRY(hipEventSynchronize(e1));
    float t = 0;
    HIP_TRY(hipEventElapsedTime(&t, e0, e1));
    if (rep > 0 && (best == 0 || t < best)) best = t;
  }
  *ms = best;
  return rc;
}

// ---- test hooks --------------------------------------------------------------------------
int bn254_debug_fp_op(bn254_ctx* c, int op, const uint8_t* a, const uint8_t* b, size_t n, uint8_t* out, uint8_t* status) {
  if (!c || (n && (!a || !out || !status))) return BN254_E_BAD_ARGUMENT;
  if (n == 0) return 0;
  HIP_TRY(hipSetDevice(c->device));
  int rc;
  if ((rc = stage_in(c, 0, a, n * 32))) return rc;
  if (b && (rc = stage_in(c, 1, b, n * 32))) return rc;
  if ((rc = stage_reserve(c, 2, n * 32))) return rc;
  if ((rc = stage_reserve(c, 3, n))) return rc;
  k_debug_fp_op<<<grid_for(n), BN_WAVE, 0, c->stream>>>(op, c->stage[0], b ? c->stage[1] : nullptr, n, c->stage[2], c->stage[3]);
  HIP_TRY(hipGetLastError());
  if ((rc = stage_out(c, 2, out, n * 32))) return rc;
  if ((rc = stage_out(c, 3, status, n))) return rc;
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}
int bn254_debug_hash_candidate(bn254_ctx* c, const uint8_t* h, size_t n, uint8_t* out, uint8_t* status) {
  if (!c || (n && (!h || !out || !status))) return BN254_E_BAD_ARGUMENT;
  if (n == 0) return 0;
  HIP_TRY(hipSetDevice(c->device));
  int rc;
  if ((rc = stage_in(c, 0, h, n * 32))) return rc;
  if ((rc = stage_reserve(c, 2, n * 64))) return rc;
  if ((rc = stage_reserve(c, 3, n))) return rc;
  k_debug_hash_candidate<<<grid_for(n), BN_WAVE, 0, c->stream>>>(c->stage[0], n, c->stage[2], c->stage[3]);
  HIP_TRY(hipGetLastError());
  if ((rc = stage_out(c, 2, out, n * 64))) return rc;
  if ((rc = stage_out(c, 3, status, n))) return rc;
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}
// layout: 0 one lane per item, exact chain (Gt out) | 1 lane pairs, program C_FE_EXACT (Gt out) | 2 lane pairs, program C_FE_CHECK |
// 3 octet (straight-line chains below 128 items, accumulator machine from 128 on) | 4 nonet | 5 one lane per item, check chain
int bn254_debug_final_exp_limbs(bn254_ctx* c, int layout, const int32_t* limbs, size_t n, uint8_t* gt, uint8_t* status) {
  if (!c || layout < 0 || layout > 5 || (n && (!limbs || !status)) || (gt && layout > 1)) return BN254_E_BAD_ARGUMENT;
  if (n == 0) return 0;
  if ((layout == 3 && !c->fits_trio) || (layout == 4 && !bn254_nonet_fits_device())) return BN254_E_BAD_ARGUMENT;
  HIP_TRY(hipSetDevice(c->device));
  int rc;
  if ((rc = ws_reserve(c, n))) return rc;
  if ((rc = stage_in(c, 0, limbs, n * 12 * BN_LIMBS * sizeof(int32_t)))) return rc;
  if ((rc = stage_reserve(c, 1, n * 384))) return rc;
  if ((rc = stage_reserve(c, 2, n))) return rc;
  hipStream_t s = c->stream;
  uint8_t* d_gt = gt ? c->stage[1] : nullptr;
  k_debug_load_f<<<grid_for(n), BN_WAVE, 0, s>>>((const int32_t*)c->stage[0], n, c->ws);
  switch (layout) {
    case 0: k_final_exp<<<grid_for(n), BN_WAVE, 0, s>>>(n, 1, 1, 1, c->ws, 0, d_gt ? d_gt : c->stage[1], c->stage[2], 0, 0, nullptr, nullptr); break;
    case 1: rc = bn254_pair_final_exp_product(n, 1, c->ws, d_gt ? d_gt : c->stage[1], c->stage[2], 0, s); break;
    case 2: rc = bn254_pair_final_exp(n, c->ws, 0, c->stage[2], nullptr, nullptr, s); break;
    case 3: rc = bn254_trio_final_exp(n, c->ws, 0, c->stage[2], s); break;
    case 4: rc = bn254_nonet_final_exp(n, c->ws, 0, c->stage[2], s); break;
    default: k_final_exp<<<grid_for(n), BN_WAVE, 0, s>>>(n, 1, 1, 1, c->ws, 0, nullptr, c->stage[2], 0, 0, nullptr, nullptr); break;
  }
  if (rc) return rc;
  HIP_TRY(hipGetLastError());
  if (gt && (rc = stage_out(c, 1, gt, n * 384))) return rc;
  if ((rc = stage_out(c, 2, status, n))) return rc;
  HIP_TRY(hipStreamSynchronize(s));
  return 0;
}
int bn254_debug_fp12_op(bn254_ctx* c, int op, const uint8_t* a, const uint8_t* b, size_t n, uint8_t* out) {
  if (!c || (n && (!a || !out))) return BN254_E_BAD_ARGUMENT;
  if (n == 0) return 0;
  HIP_TRY(hipSetDevice(c->device));
  int rc;
  if ((rc = stage_in(c, 0, a, n * 384))) return rc;
  if (b && (rc = stage_in(c, 1, b, n * 384))) return rc;
  if ((rc = stage_reserve(c, 2, n * 384))) return rc;
  k_debug_fp12_op<<<grid_for(n), BN_WAVE, 0, c->stream>>>(op, c->stage[0], b ? c->stage[1] : nullptr, n, c->stage[2]);
  HIP_TRY(hipGetLastError());
  if ((rc = stage_out(c, 2, out, n * 384))) return rc;
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

}  // extern "C"
