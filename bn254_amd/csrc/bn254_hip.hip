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
#include "bn254_lane.h"
#include "bn254_host.h"

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


// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------

int ctx_quiesce(bn254_ctx* c) {
  HIP_TRY(hipStreamSynchronize(c->stream));
  HIP_TRY(hipStreamSynchronize(c->copy_stream));
  if (c->last_done_armed) {                         // the last *_device call ran on a caller's stream: wait for the event recorded behind it
    c->last_done_armed = false;
    HIP_TRY(hipEventSynchronize(c->last_done));
  }
  return 0;
}
int ws_reserve(bn254_ctx* c, size_t n) {
  if (n <= c->ws.stride) return 0;
  size_t cap = (n + 255) & ~(size_t)255;
  HIP_TRY(hipSetDevice(c->device));
  { int rc_ = ctx_quiesce(c); if (rc_) return rc_; }     // a *_device call may still be running on a caller's stream, not only on c->stream
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
#define WS_BYTES_PER_ITEM ((size_t)N_PLANES * BN_LIMBS * sizeof(int32_t) + N_BYTE_PLANES + sizeof(uint32_t) + 1 + 2 * sizeof(uint32_t))
size_t ws_chunk_for(bn254_ctx* c, size_t n) {
  if (c->max_chunk > 0) return n > (size_t)c->max_chunk ? (size_t)c->max_chunk : 0;
  if (n <= c->ws.stride) return 0;                   // already reserved
  size_t avail;
  if (c->assume_free_mb > 0) avail = (size_t)c->assume_free_mb << 20;
  else {
    size_t fr = 0, total = 0;
    if (hipSetDevice(c->device) != hipSuccess || hipMemGetInfo(&fr, &total) != hipSuccess) { (void)hipGetLastError(); return 0; }
    avail = fr + c->ws.stride * WS_BYTES_PER_ITEM;   // growing frees the present workspace first
  }
  const size_t need = ((n + 255) & ~(size_t)255) * WS_BYTES_PER_ITEM;
  if (need <= avail / 10 * 9) return 0;
  size_t chunk = (avail / 10 * 8) / WS_BYTES_PER_ITEM;
  chunk &= ~(size_t)65535;
  if (chunk == 0) chunk = (avail / 10 * 8) / WS_BYTES_PER_ITEM & ~(size_t)255;   // a very small device share: whatever fits (a failure to allocate is then reported as before)
  return chunk && chunk < n ? chunk : 0;
}
int stage_reserve(bn254_ctx* c, int slot, size_t bytes) {
  if (bytes <= c->stage_cap[slot]) return 0;
  HIP_TRY(hipSetDevice(c->device));
  { int rc_ = ctx_quiesce(c); if (rc_) return rc_; }
  if (c->stage[slot]) { HIP_TRY(hipFree(c->stage[slot])); c->stage[slot] = nullptr; c->stage_cap[slot] = 0; }
  size_t cap = (bytes + 4095) & ~(size_t)4095;
  HIP_TRY(hipMalloc((void**)&c->stage[slot], cap));
  c->stage_cap[slot] = cap;
  return 0;
}
int stage_in(bn254_ctx* c, int slot, const void* host, size_t bytes) {
  int rc = stage_reserve(c, slot, bytes ? bytes : 1);
  if (rc) return rc;
  if (bytes) HIP_TRY(hipMemcpyAsync(c->stage[slot], host, bytes, hipMemcpyHostToDevice, c->stream));
  return 0;
}
int stage_out(bn254_ctx* c, int slot, void* host, size_t bytes) {
  if (bytes) HIP_TRY(hipMemcpyAsync(host, c->stage[slot], bytes, hipMemcpyDeviceToHost, c->stream));
  return 0;
}
// a pool outside the numbered slots (the comb table of the G2 generator): allocated once, never resized in practice
int pool_reserve_one(bn254_ctx* c, Pool* p, size_t n_fp, size_t entries) {
  if (entries <= p->stride) return 0;
  HIP_TRY(hipSetDevice(c->device));
  { int rc_ = ctx_quiesce(c); if (rc_) return rc_; }
  if (p->planes) { HIP_TRY(hipFree(p->planes)); p->planes = nullptr; }
  if (p->st) { HIP_TRY(hipFree(p->st)); p->st = nullptr; }
  p->stride = 0;
  const size_t cap = (entries + 255) & ~(size_t)255;
  p->g2 = n_fp == 4 ? 1u : 0u;
  HIP_TRY(hipMalloc((void**)&p->planes, (n_fp / 2) * BN_POOL_HALF_WORDS * sizeof(int32_t) * cap));
  HIP_TRY(hipMalloc((void**)&p->st, cap));
  p->stride = cap;
  return 0;
}
int pool_reserve(bn254_ctx* c, int which, size_t n_fp, size_t entries) {
  Pool& p = c->pool[which];
  if (entries <= p.stride && c->pool_fp[which] == n_fp) return 0;
  HIP_TRY(hipSetDevice(c->device));
  { int rc_ = ctx_quiesce(c); if (rc_) return rc_; }
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
int launch_decode_g2(bn254_ctx* c, hipStream_t s, const uint8_t* d_pts, size_t n, uint32_t flags, int accumulate) {
  if (c->pair_lanes && (flags & FLAG_G2_SUBGROUP_CHECK)) {
    if (route_lane_machine_helpers(c, n)) {
      // the smallest batches: decode without the test, then the test with its ladder in the lane machine's level tables
      k_decode_g2<<<grid_for(n), BN_WAVE, 0, s>>>(d_pts, n, flags & ~(uint32_t)FLAG_G2_SUBGROUP_CHECK, c->ws, accumulate);
      return bn254_lm_g2_subgroup(n, c->ws, s);
    }
    return bn254_pair_decode_g2(d_pts, n, flags, c->ws, accumulate, s);
  }
  k_decode_g2<<<grid_for(n), BN_WAVE, 0, s>>>(d_pts, n, flags, c->ws, accumulate);
  return 0;
}

// The schedule (widths, grid sizes) is fixed on the host from the EXPECTED survivor counts
// (p_fail = 0.5274 per try); the kernels read the actual counts from device memory and use grid-stride
// loops, so a wrong estimate costs time, never correctness.  No host synchronisation.
int launch_hash_rounds(bn254_ctx* c, hipStream_t s, const uint8_t* d_msgs, const uint64_t* d_off, size_t n, int px, int inf_plane,
                              uint8_t* d_tries, int mark_finish) {           // mark_finish: profiling event recorded in front of k_hash_finish
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

// launchers for the other translation units (bn254_host.h)
int launch_decode_g1(bn254_ctx* c, hipStream_t s, const uint8_t* d_pts, size_t n, uint32_t flags, int px, int inf_plane, int accumulate) {
  k_decode_g1<<<grid_for(n), BN_WAVE, 0, s>>>(d_pts, n, flags, c->ws, px, inf_plane, accumulate);
  return 0;
}
int launch_miller_verify_lane(bn254_ctx* c, hipStream_t s, size_t n, const uint32_t* map, const uint32_t* count) {
  k_miller_verify<<<grid_for(n), BN_WAVE, 0, s>>>(n, c->ws, map, count);
  return 0;
}
int launch_final_exp_lane(bn254_ctx* c, hipStream_t s, size_t n, size_t k, size_t item_stride, size_t pair_stride, int use_hash, uint8_t* gt_out,
                          uint8_t* status_out, int raw_only, size_t base, const uint32_t* map, const uint32_t* count) {
  k_final_exp<<<grid_for(n), BN_WAVE, 0, s>>>(n, k, item_stride, pair_stride, c->ws, use_hash, gt_out, status_out, raw_only, base, map, count);
  return 0;
}
int launch_encode_g1(bn254_ctx* c, hipStream_t s, size_t n, int px, int inf_plane, uint8_t* out, uint8_t* status_out) {
  k_encode_g1<<<grid_for(n), BN_WAVE, 0, s>>>(n, c->ws, px, inf_plane, out, status_out);
  return 0;
}

extern "C" {

const char* bn254_version(void) { return "bn254-mi355x 0.7 (gfx950; 9x29-bit balanced Montgomery limbs; verify on lane pairs, batches <= 16384 on lane octets with wave roles; bn254_mgpu_*: all the GPUs of a node behind one handle)"; }

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
  c->lm_max_batch = bn254_lm_fits_device() ? LM_MAX_BATCH_DEFAULT : 0;
  c->nonet_wide = 1;
  c->g2_fixed_base = 1;
  c->device = hip_device;
  hipError_t err = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (err == hipSuccess) err = hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking);
  if (err == hipSuccess) err = hipEventCreateWithFlags(&c->copy_done, hipEventDisableTiming);
  if (err == hipSuccess) err = hipEventCreateWithFlags(&c->last_done, hipEventDisableTiming);
  for (int i = 0; i < 5 && err == hipSuccess; ++i) err = hipEventCreate(&c->ev[i]);
  if (err != hipSuccess) {
    if (c->last_done) (void)hipEventDestroy(c->last_done);
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
  (void)ctx_quiesce(c);                               // own streams + the end of the last call on a caller's stream
  if (c->ws.planes) (void)hipFree(c->ws.planes);
  if (c->ws.bytes) (void)hipFree(c->ws.bytes);
  if (c->ws.h_best) (void)hipFree(c->ws.h_best);
  if (c->ws.h_next) (void)hipFree(c->ws.h_next);
  if (c->ws.h_list) (void)hipFree(c->ws.h_list);
  if (c->ws.h_cnt) (void)hipFree(c->ws.h_cnt);
  if (c->ws.clk) (void)hipFree(c->ws.clk);
  if (c->pin) (void)hipHostFree(c->pin);
  for (int i = 0; i < 8; ++i) { if (c->pool[i].planes) (void)hipFree(c->pool[i].planes); if (c->pool[i].st) (void)hipFree(c->pool[i].st); }
  if (c->g2_comb.planes) (void)hipFree(c->g2_comb.planes);
  if (c->g2_comb.st) (void)hipFree(c->g2_comb.st);
  if (c->g1_comb.planes) (void)hipFree(c->g1_comb.planes);
  if (c->g1_comb.st) (void)hipFree(c->g1_comb.st);
  for (int i = 0; i < 8; ++i) if (c->stage[i]) (void)hipFree(c->stage[i]);
  if (c->key_lines) (void)hipFree(c->key_lines);
  if (c->key_xy) (void)hipFree(c->key_xy);
  if (c->key_st) (void)hipFree(c->key_st);
  if (c->key_inf) (void)hipFree(c->key_inf);
  for (int i = 0; i < 5; ++i) (void)hipEventDestroy(c->ev[i]);
  (void)hipEventDestroy(c->copy_done);
  (void)hipEventDestroy(c->last_done);
  (void)hipStreamSynchronize(c->copy_stream);
  (void)hipStreamDestroy(c->copy_stream);
  (void)hipStreamDestroy(c->stream);
  delete c;
}
int bn254_ctx_reserve(bn254_ctx* c, size_t n) { return c ? ws_reserve(c, n) : BN254_E_BAD_ARGUMENT; }
int bn254_ctx_reserve_host(bn254_ctx* c, size_t n, size_t msg_bytes) {
  if (!c) return BN254_E_BAD_ARGUMENT;
  int rc = ws_reserve(c, n);
  const size_t need[5] = {msg_bytes ? msg_bytes : 1, (n + 1) * sizeof(uint64_t), n * 64, n * 128, n ? n : 1};
  for (int slot = 0; slot < 5 && !rc; ++slot) rc = stage_reserve(c, slot, need[slot]);
  return rc;
}
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
    HIP_TRY(hipSetDevice(c->device));                 // the fits query asks the CURRENT device
    if (value < 0 || (value > 0 && !bn254_nonet_fits_device())) return BN254_E_BAD_ARGUMENT;
    c->nonet_max_batch = value;
    return 0;
  }
  if (option == BN254_OPT_NONET_WIDE) { c->nonet_wide = value != 0; return 0; }
  if (option == BN254_OPT_LM_MAX_BATCH) {
    HIP_TRY(hipSetDevice(c->device));                 // the fits query asks the CURRENT device
    if (value < 0 || (value > 0 && !bn254_lm_fits_device())) return BN254_E_BAD_ARGUMENT;
    c->lm_max_batch = value;
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
  if (option == BN254_OPT_G2_FIXED_BASE) { c->g2_fixed_base = value != 0; return 0; }
  if (option == BN254_OPT_MAX_CHUNK) { if (value < 0) return BN254_E_BAD_ARGUMENT; c->max_chunk = value; return 0; }
  if (option == BN254_OPT_ASSUME_FREE_MB) { if (value < 0) return BN254_E_BAD_ARGUMENT; c->assume_free_mb = value; return 0; }
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


}  // extern "C"
// Miller loop + final exponentiation of a verify-shaped batch on lane pairs, or — for batches that cannot fill the chip —
// in the octet layout (three lane pairs share the Fq6 products of every Fq12 operation: fewer instructions per lane,
// which is what latency is made of when a wave has its SIMD to itself).  Same status bytes either way.
// final exponentiation of a small batch: the smallest on nine lane pairs per verify (bn254_nonet.hip; eighteen while one verify per wave
// still covers the batch) — fewer instructions per lane again —, the others in the octet layout
// (the layouts come from the routing table, bn254_ws.h: bn_route; a caller that has ALREADY chosen a small-batch Miller kernel — the keyed lane
// machine — asks for the small-batch final exponentiation of the same row)
static int launch_final_exp_layout(bn254_ctx* c, hipStream_t s, size_t n, int use_hash, uint8_t* d_status, int fe) {
  switch (fe) {
    case BN_FE_NONET_WIDE: return bn254_nonet_final_exp(n, c->ws, use_hash, d_status, s, 1);
    case BN_FE_NONET: return bn254_nonet_final_exp(n, c->ws, use_hash, d_status, s, 0);
    case BN_FE_OCTET: return bn254_trio_final_exp(n, c->ws, use_hash, d_status, s);
    default: return bn254_pair_final_exp(n, c->ws, use_hash, d_status, nullptr, nullptr, s);
  }
}
int launch_small_final_exp(bn254_ctx* c, hipStream_t s, size_t n, int use_hash, uint8_t* d_status) {
  const BnRoute r = route_for(c, n);
  return launch_final_exp_layout(c, s, n, use_hash, d_status, r.fe == BN_FE_LANE_PAIRS ? BN_FE_OCTET : r.fe);
}
int launch_pair_or_trio(bn254_ctx* c, hipStream_t s, size_t n, int use_hash, uint8_t* d_status, int mode, bool mark) {
  const BnRoute r = route_for(c, n);
  int rc;
  switch (r.miller) {
    case BN_ML_LANE_MACHINE: rc = bn254_lm_miller_verify(n, c->ws, s, mode); break;
    case BN_ML_WAVE_ROLES:         // BN254_OPT_TRIO_WAVE_ROLES (developer knob): eight waves (default), four, or the lane groups of one wave
      rc = c->trio_wave_roles == 2 ? bn254_w8_miller_verify(n, c->ws, s, mode)
           : c->trio_wave_roles ? bn254_quad_miller_verify(n, c->ws, s, mode) : bn254_trio_miller_verify(n, c->ws, s, mode);
      break;
    default: rc = bn254_pair_miller_verify(n, c->ws, nullptr, nullptr, s, mode); break;
  }
  if (rc) return rc;
  if (mark) PROF_MARK(3);
  return launch_final_exp_layout(c, s, n, use_hash, d_status, r.fe);
}

extern "C" {
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
  if (const size_t chunk = ws_chunk_for(c, n)) {
    // an oversized batch: slices of `chunk` items through this same entry point, one after the other on the caller's stream — the offsets are
    // absolute into d_msgs, so a slice is the same arrays further in; statuses land at the items' own positions (profiling: the last slice's)
    for (size_t lo = 0; lo < n; lo += chunk) {
      const size_t len = n - lo < chunk ? n - lo : chunk;
      const int rc_ = bn254_batch_verify_device(c, d_msgs, d_off + lo, d_sigs + 64 * lo, d_pks + 128 * lo, len, flags, d_status + lo, stream);
      if (rc_) return rc_;
    }
    return 0;
  }
  int rc = ws_reserve(c, split ? 2 * n : n);
  if (rc) return rc;
  hipStream_t s = stream ? (hipStream_t)stream : c->stream;
  CallDone call_done(c, s);
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
  if (const size_t chunk = ws_chunk_for(c, n)) {       // an oversized batch in slices (see bn254_batch_verify_device)
    for (size_t lo = 0; lo < n; lo += chunk) {
      const size_t len = n - lo < chunk ? n - lo : chunk;
      const int rc_ = bn254_batch_verify_compressed_device(c, d_msgs, d_off + lo, d_sigs33 + 33 * lo, d_pks65 + 65 * lo, len, d_status + lo, stream);
      if (rc_) return rc_;
    }
    return 0;
  }
  int rc = ws_reserve(c, n);
  if (rc) return rc;
  hipStream_t s = stream ? (hipStream_t)stream : c->stream;
  CallDone call_done(c, s);
  PROF_MARK(0);
  k_decompress_g1_ws<<<grid_for(n), BN_WAVE, 0, s>>>(d_sigs33, n, c->ws);
  if (route_lane_machine_helpers(c, n)) {
    // the smallest batches: the subgroup test of the decompressed keys with its ladder in the lane machine's level tables
    if ((rc = bn254_pair_decompress_g2(d_pks65, n, c->ws, s, 1))) return rc;
    if ((rc = bn254_lm_g2_subgroup(n, c->ws, s, ST_NOT_MEMBER))) return rc;
  } else if (c->pair_lanes) { if ((rc = bn254_pair_decompress_g2(d_pks65, n, c->ws, s))) return rc; }
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
  if (const size_t chunk = ws_chunk_for(c, n)) {       // an oversized batch in slices, offsets rebased per slice (see bn254_batch_verify)
    uint64_t* tmp = (uint64_t*)malloc((chunk + 1) * sizeof(uint64_t));
    if (!tmp) return BN254_E_NO_MEMORY;
    rc = 0;
    for (size_t lo = 0; lo < n && !rc; lo += chunk) {
      const size_t len = n - lo < chunk ? n - lo : chunk;
      for (size_t i = 0; i <= len; ++i) tmp[i] = off[lo + i] - off[lo];
      rc = bn254_batch_verify_compressed(c, msgs ? msgs + off[lo] : nullptr, tmp, sigs33 + 33 * lo, pks65 + 65 * lo, len, status + lo);
    }
    free(tmp);
    return rc;
  }
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
  { int rc_ = ctx_quiesce(c); if (rc_) return rc_; }
  if (c->pin) { HIP_TRY(hipHostFree(c->pin)); c->pin = nullptr; c->pin_cap = 0; }
  const size_t cap = (bytes + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
  HIP_TRY(hipHostMalloc((void**)&c->pin, cap, hipHostMallocDefault));
  c->pin_cap = cap;
  return 0;
}
static int pinned_copy_in(bn254_ctx* c, uint8_t* d_dst, uint8_t* pin, const uint8_t* src, size_t bytes, hipStream_t stream, int threads) {
  if (!bytes) return 0;
  const size_t piece = (size_t)1 << 20;
  if (threads > 16) threads = 16;
  if (threads > 1 && bytes < 4 * piece) threads = 1;
  int rcs[16] = {0};
  auto work = [&](int t, int n_threads) {
    if (t && hipSetDevice(c->device) != hipSuccess) { rcs[t] = -1; return; }
    const size_t share = ((bytes + n_threads - 1) / n_threads + 255) & ~(size_t)255, lo = (size_t)t * share, hi = lo + share < bytes ? lo + share : bytes;
    for (size_t o = lo; o < hi; o += piece) {
      const size_t len = o + piece < hi ? piece : hi - o;
      memcpy(pin + o, src + o, len);
      const hipError_t e = hipMemcpyAsync(d_dst + o, pin + o, len, hipMemcpyHostToDevice, stream);
      if (e != hipSuccess) { rcs[t] = -(int)e; return; }
    }
  };
  // helper threads: nothing may unwind through the C ABI — if a thread cannot be created the caller's thread copies everything itself
  std::thread helpers[15];
  int started = 0;
  bool fallback = false;
  try {
    for (int t = 1; t < threads; ++t) { helpers[started] = std::thread(work, t, threads); ++started; }
  } catch (...) {
    fallback = true;
  }
  if (!fallback) work(0, threads);
  for (int t = 0; t < started; ++t) helpers[t].join();
  if (fallback) {                                   // the shares of the helpers that did start are done (and harmlessly redone here)
    for (int t = 0; t < 16; ++t) rcs[t] = 0;
    work(0, 1);
  }
  for (int t = 0; t < 16; ++t) if (rcs[t]) return rcs[t];
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
  if (const size_t chunk = ws_chunk_for(c, n)) {
    // an oversized batch: slices through this same entry point, each with its offsets rebased to its own first message byte
    uint64_t* tmp = (uint64_t*)malloc((chunk + 1) * sizeof(uint64_t));
    if (!tmp) return BN254_E_NO_MEMORY;
    rc = 0;
    for (size_t lo = 0; lo < n && !rc; lo += chunk) {
      const size_t len = n - lo < chunk ? n - lo : chunk;
      for (size_t i = 0; i <= len; ++i) tmp[i] = off[lo + i] - off[lo];
      rc = bn254_batch_verify(c, msgs ? msgs + off[lo] : nullptr, tmp, sigs + 64 * lo, pks + 128 * lo, len, flags, status + lo);
    }
    free(tmp);
    return rc;
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
  CallDone call_done(c, s);
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
  CallDone call_done(c, s);
  PROF_MARK(0);
  k_decode_g1<<<grid_for(lanes), BN_WAVE, 0, s>>>(d_g1, lanes, flags, c->ws, PL_P1X, BY_P1_INF, 0);
  if ((rc = launch_decode_g2(c, s, d_g2, lanes, flags, 1))) return rc;
  PROF_MARK(1);
  PROF_MARK(2);                                      // no hash in a pairing: ms[1] = 0
  if (!raw_only && route_lane_machine_helpers(c, lanes) && route_for(c, n).fe == BN_FE_NONET_WIDE) {
    // a batch that cannot fill the chip: the small-batch kernels of a verify — the lane machine with the fixed
    // pair skipped, the final exponentiation (exact program, Gt bytes) on eighteen lane pairs per item: 5.7 -> 1.3 ms for one pairing
    if ((rc = bn254_lm_miller_verify(lanes, c->ws, s, 2))) return rc;
    PROF_MARK(3);
    if ((rc = bn254_nonet_final_exp_product(n, k, c->ws, d_gt, d_status, s))) return rc;
  } else if (c->pair_lanes) {
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

}  // extern "C"
