// Translation unit of libbn254hip.so: the REGISTERED-KEY and RANDOMISED modes of the verify path, one lane per item (the lane-pair kernels
// they hand over to live in bn254_pair.hip) — kernels and the host side of their entry points (include/bn254_hip.h):
//   bn254_ctx_register_keys, bn254_batch_verify_keyed[_device], bn254_batch_verify_keyed_randomized[_device],
//   bn254_batch_verify_randomized[_device].
// Per-tuple semantics: /root/reference/src/ecdsa.rs:49-64; key validation: /root/reference/src/types.rs:96-99.
#include <hip/hip_runtime.h>

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


extern "C" {

// ---- keyed verify (include/bn254_hip.h) ---------------------------------------------------------------------------------
int bn254_ctx_register_keys(bn254_ctx* c, const uint8_t* pks, size_t n_keys, uint32_t flags, uint8_t* key_status) {
  if (!c || (n_keys && !pks) || n_keys > 0xFFFFFFFFu) return BN254_E_BAD_ARGUMENT;
  HIP_TRY(hipSetDevice(c->device));
  { int rc_ = ctx_quiesce(c); if (rc_) return rc_; }   // no keyed verify — on c->stream or on the caller's stream of the last call — may still be reading the previous tables
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
  if (const size_t chunk = ws_chunk_for(c, n)) {       // an oversized batch in slices (see bn254_batch_verify_device)
    for (size_t lo = 0; lo < n; lo += chunk) {
      const size_t len = n - lo < chunk ? n - lo : chunk;
      const int rc_ = bn254_batch_verify_keyed_device(c, d_msgs, d_off + lo, d_sigs + 64 * lo, d_key_idx + lo, len, flags, d_status + lo, stream);
      if (rc_) return rc_;
    }
    return 0;
  }
  int rc = ws_reserve(c, n);
  if (rc) return rc;
  hipStream_t s = stream ? (hipStream_t)stream : c->stream;
  CallDone call_done(c, s);
  KeyTable kt = {c->key_lines, c->key_st, c->key_inf, (uint32_t)c->n_keys};
  PROF_MARK(0);
  { int rc_ = launch_decode_g1(c, s, d_sigs, n, flags, PL_P1X, BY_P1_INF, 0); if (rc_) return rc_; }
  if (c->n_keys == 0 || !c->key_lines) {             // nothing registered: no table to read — every item is out of range
    k_keyed_no_keys<<<grid_for(n), BN_WAVE, 0, s>>>(n, c->ws, d_status);
    HIP_TRY(hipGetLastError());
    return 0;
  }
  PROF_MARK(1);
  if ((rc = launch_hash_rounds(c, s, d_msgs, d_off, n, PL_P2X, BY_P2_INF, nullptr))) return rc;
  PROF_MARK(2);
  if (route_for(c, n).miller != BN_ML_LANE_PAIRS) {
    // a batch that cannot fill the chip: latency counts — expand the keys and take the small-batch kernels (2.3 ms instead of the
    // 6 ms of the lane-pair layout; the line tables pay off only where throughput binds)
    if (route_for(c, n).miller == BN_ML_LANE_MACHINE) {
      // the smallest: the lane machine's keyed form on the line tables themselves (no twist point to walk: 0.32 ms against 0.43)
      if ((rc = bn254_lm_miller_verify_keyed(n, c->ws, d_key_idx, kt, s))) return rc;
      PROF_MARK(3);
      if ((rc = launch_small_final_exp(c, s, n, 1, d_status))) return rc;
    } else {
      k_keyed_expand<<<grid_for(n), BN_WAVE, 0, s>>>(n, c->ws, d_key_idx, kt, c->key_xy);
      if ((rc = launch_pair_or_trio(c, s, n, 1, d_status, 0, true))) return rc;
    }
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
  CallDone call_done(c, s);
  Seed seed;
  for (int j = 0; j < 8; ++j)
    seed.w[j] = ((uint32_t)seed32[4 * j] << 24) | ((uint32_t)seed32[4 * j + 1] << 16) | ((uint32_t)seed32[4 * j + 2] << 8) | seed32[4 * j + 3];
  KeyTable kt = {c->key_lines, c->key_st, c->key_inf, (uint32_t)c->n_keys};
  PROF_MARK(0);
  { int rc_ = launch_decode_g1(c, s, d_sigs, n, dflags, PL_P1X, BY_P1_INF, 0); if (rc_) return rc_; }
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
  CallDone call_done(c, s);
  Seed seed;
  for (int j = 0; j < 8; ++j)
    seed.w[j] = ((uint32_t)seed32[4 * j] << 24) | ((uint32_t)seed32[4 * j + 1] << 16) | ((uint32_t)seed32[4 * j + 2] << 8) | seed32[4 * j + 3];
  const unsigned g = grid_for(n), gg = grid_for(n_groups);
  const uint32_t dflags = flags & (BN254_FLAG_G2_SUBGROUP_CHECK | BN254_FLAG_REJECT_IDENTITY);
  uint8_t* d_group_st = c->ws.h_next;            // free once the hash rounds are done; n_groups <= stride
  PROF_MARK(0);
  { int rc_ = launch_decode_g1(c, s, d_sigs, n, dflags, PL_P1X, BY_P1_INF, 0); if (rc_) return rc_; }
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
    { int rc_ = launch_final_exp_lane(c, s, n_groups, 1, 1, 1, 0, nullptr, d_group_st, 0, gbase, nullptr, nullptr); if (rc_) return rc_; }
  }
  k_rand_collect<<<g, BN_WAVE, 0, s>>>(n, c->ws, d_group_st, d_status, d_group_ok);
  // exact per-item check of the items of failed groups (none queued: both kernels leave at once)
  if (c->pair_lanes) {
    if ((rc = bn254_pair_miller_verify(n, c->ws, c->ws.h_list, c->ws.h_cnt, s))) return rc;
    if ((rc = bn254_pair_final_exp(n, c->ws, 1, d_status, c->ws.h_list, c->ws.h_cnt, s))) return rc;
  } else {
    { int rc_ = launch_miller_verify_lane(c, s, n, c->ws.h_list, c->ws.h_cnt); if (rc_) return rc_; }
    { int rc_ = launch_final_exp_lane(c, s, n, 1, 1, 1, 1, nullptr, d_status, 0, 0, c->ws.h_list, c->ws.h_cnt); if (rc_) return rc_; }
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

}  // extern "C"
