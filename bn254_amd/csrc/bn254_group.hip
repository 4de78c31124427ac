// Translation unit of libbn254hip.so: GROUP OPERATIONS and the aggregate verify of BASELINE configs[2], one lane per item — point addition
// and sums (`Add for Signature / PublicKey`, /root/reference/src/types.rs:126-132, :264-270), scalar multiplication (sign / key derivation,
// src/ecdsa.rs:26-35, src/types.rs:85-87, :155-157), the compressed decoders (src/types.rs:91-93, :233-237), the pools and subset-sum
// tables of bn254_batch_aggregate_verify — kernels and the host side of their entry points (include/bn254_hip.h).
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
  g1_mul_glv_full(jo, pa, k);                       // round 6: the joint 128-step ladder over the endomorphism (bn254_curve.h); a raw scalar acts mod r
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
template <class F, int NLO, int BATCH = 8> __device__ __forceinline__ void pool_widen_lane(bool live, const Pool& src, size_t b0, Affine<F> A, const Pool& dst, size_t dst0) {
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
// The 4-signer signature tables themselves are built the same way, in two steps (round 5: 4.6 -> ~1.1 ms per 1 024 x 1 024 pool):
//   k_pool_pairs_g1: T2[m][pair][mask] = {O, s0, s1, s0 + s1} for every pair of consecutive signers (one complete addition and one
//                    inversion per PAIR), then  k_pool_quads_g1: T4[hi * 4 + lo] = T2[pair 2g][lo] + T2[pair 2g + 1][hi], batches of 4.
// k_pool_subsets_g1 above (four accumulations and an inversion per ENTRY) stays as the fallback when the pair table cannot be allocated.
KERNEL_SMALL void k_pool_pairs_g1(Pool sig_pool, size_t n_signers, size_t groups2, size_t n_msgs, Pool t2) {
  const size_t lane = (size_t)blockIdx.x * BN_WAVE + threadIdx.x;
  const bool live = lane < n_msgs * groups2;
  const size_t ll = live ? lane : 0, g = ll % groups2, m = ll / groups2;
  G1Affine s[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const size_t sgn = 2 * g + b, sj = m * n_signers + (sgn < n_signers ? sgn : 0);
    s[b].x = pool_load_fp(sig_pool, 0, sj); s[b].y = pool_load_fp(sig_pool, 1, sj);
    s[b].inf = sgn >= n_signers || sig_pool.st[sj] != 0;          // st: 0x80 = identity entry, low bits = decode error (counts as the identity here)
    if (s[b].inf) { s[b].x = fp_load_const(C_G1_GEN[0]); s[b].y = fp_load_const(C_G1_GEN[1]); }
  }
  G1Jac J;
  G1Affine sum, none;
  jac_from_affine(J, s[0]);
  jac_madd(J, J, s[1]);
  jac_to_affine(sum, J);
  none.x = fp_zero(); none.y = fp_zero(); none.inf = true;
  if (!live) return;
  pool_store_aff(t2, 4 * lane + 0, none);
  pool_store_aff(t2, 4 * lane + 1, s[0]);
  pool_store_aff(t2, 4 * lane + 2, s[1]);
  pool_store_aff(t2, 4 * lane + 3, sum);
}
KERNEL_SMALL void k_pool_quads_g1(Pool t2, size_t groups2, size_t groups4, size_t n_msgs, Pool t4) {
  const size_t lane = (size_t)blockIdx.x * BN_WAVE + threadIdx.x;
  const bool live = lane < n_msgs * groups4 * 4;
  const size_t ll = live ? lane : 0, hi = ll & 3u, g = (ll >> 2) % groups4, m = (ll >> 2) / groups4;
  G1Affine A;
  pool_load_aff(t2, (m * groups2 + 2 * g + 1) * 4 + hi, A);
  if (A.inf) { A.x = fp_load_const(C_G1_GEN[0]); A.y = fp_load_const(C_G1_GEN[1]); }
  pool_widen_lane<Fp, 4, 4>(live, t2, (m * groups2 + 2 * g) * 4, A, t4, (m * groups4 + g) * 16 + hi * 4);
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

extern "C" {

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

static int comb_build(bn254_ctx* c, int g2);
KERNEL void k_g1_gen_mul_reduce(const uint8_t* scalars, size_t n, int reduce, uint8_t* out, uint8_t* status);
int bn254_batch_g1_mul_device(bn254_ctx* c, const uint8_t* d_p, const uint8_t* d_k, size_t n, int reduce, uint8_t* d_out, uint8_t* d_status, void* stream) {
  if (!c || (n && (!d_k || !d_out || !d_status))) return BN254_E_BAD_ARGUMENT;    // d_p == NULL: the generator (PublicKeyG1::from_private_key)
  if (n == 0) return 0;
  if ((d_p && misaligned(d_p)) || misaligned(d_k) || misaligned(d_out)) return BN254_E_MISALIGNED;
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t s = stream ? (hipStream_t)stream : c->stream;
  if (!d_p) {
    if (c->pair_lanes && c->g2_fixed_base) {           // the fixed generator: 65 table additions split over the two lanes of a pair
      int rc = comb_build(c, 0);
      if (rc) return rc;
      CallDone call_done(c, s);
      return bn254_pair_g1_mul_fixed(d_k, n, reduce, c->g1_comb, d_out, d_status, s);
    }
    CallDone call_done(c, s);
    k_g1_gen_mul_reduce<<<grid_for(n), BN_WAVE, 0, s>>>(d_k, n, reduce, d_out, d_status);
    HIP_TRY(hipGetLastError());
    return 0;
  }
  CallDone call_done(c, s);
  k_g1_mul<<<grid_for(n), BN_WAVE, 0, s>>>(d_p, d_k, n, reduce, c->ws, d_out, d_status);
  HIP_TRY(hipGetLastError());
  return 0;
}
// The comb table of the generator for key derivation (k_g2_mul_fixed_pair, bn254_pair.hip): records j * 8 + (d - 1) = d 16^j G2::one() for
// j = 0 .. 64, d = 1 .. 8, and record 520 = the blinding point.  Built once per context with the library's own ladder (k_g2_mul on the 521
// scalars, reduced mod r on the host) and decoder; ~6 ms, on the context's own stream, waited for before the first use.
// sk * G1::one() by the general ladder: the fallback of bn254_batch_g1_mul(points = NULL) when the comb table is switched off
KERNEL void k_g1_gen_mul_reduce(const uint8_t* scalars, size_t n, int reduce, uint8_t* out, uint8_t* status) {
  size_t i = (size_t)blockIdx.x * BN_WAVE + threadIdx.x;
  if (i >= n) return;
  G1Affine g, r;
  g1_set_generator(g);
  uint32_t k[8];
  scalar_from_be(k, scalars + 32 * i, reduce != 0);
  G1Jac jo;
  g1_mul_glv_full(jo, g, k);
  jac_to_affine(r, jo);
  encode_g1(out + 64 * i, r);
  status[i] = ST_OK;
}
// ... and for the builder of the comb table
KERNEL void k_g1_gen_mul(const uint8_t* scalars, size_t n, uint8_t* out) {
  size_t i = (size_t)blockIdx.x * BN_WAVE + threadIdx.x;
  if (i >= n) return;
  G1Affine g, r;
  g1_set_generator(g);
  uint32_t k[8];
  scalar_from_be(k, scalars + 32 * i, false);
  G1Jac jo;
  jac_mul(jo, g, k);
  jac_to_affine(r, jo);
  encode_g1(out + 64 * i, r);
}
static int comb_build(bn254_ctx* c, int g2) {
  if (g2 ? c->g2_comb_ready : c->g1_comb_ready) return 0;
  const size_t N = g2 ? BN_G2_COMB_RECORDS : BN_G1_COMB_RECORDS;
  const size_t sz = g2 ? 128 : 64;
  uint8_t* h = (uint8_t*)malloc(N * 32);
  if (!h) return BN254_E_NO_MEMORY;
  // 256-bit arithmetic mod r on little-endian words: x -> 2x mod r, x + y mod r
  const uint32_t* R = nullptr;
  static const uint32_t r_words[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
  R = r_words;
  auto geq = [&](const uint32_t* a) { for (int i = 7; i >= 0; --i) { if (a[i] != R[i]) return a[i] > R[i]; } return true; };
  auto sub_r = [&](uint32_t* a) { uint64_t bw = 0; for (int i = 0; i < 8; ++i) { uint64_t d = (uint64_t)a[i] - R[i] - bw; a[i] = (uint32_t)d; bw = (d >> 63) & 1; } };
  auto add = [&](uint32_t* a, const uint32_t* b) {         // a, b < r < 2^254: the sum fits 256 bits
    uint64_t cy = 0; for (int i = 0; i < 8; ++i) { uint64_t t = (uint64_t)a[i] + b[i] + cy; a[i] = (uint32_t)t; cy = t >> 32; }
    if (geq(a)) sub_r(a);
  };
  auto put = [&](size_t idx, const uint32_t* a) { for (int i = 0; i < 8; ++i) { uint32_t w = a[7 - i]; h[32 * idx + 4 * i] = (uint8_t)(w >> 24); h[32 * idx + 4 * i + 1] = (uint8_t)(w >> 16); h[32 * idx + 4 * i + 2] = (uint8_t)(w >> 8); h[32 * idx + 4 * i + 3] = (uint8_t)w; } };
  uint32_t base[8] = {1, 0, 0, 0, 0, 0, 0, 0};              // 16^j mod r
  for (int j = 0; j < 65; ++j) {
    uint32_t acc[8] = {0};
    for (int d = 1; d <= 8; ++d) { add(acc, base); put((size_t)j * 8 + (d - 1), acc); }
    for (int t = 0; t < 4; ++t) { uint32_t dbl[8]; memcpy(dbl, base, sizeof dbl); add(base, dbl); }
  }
  // the blinding point: a fixed scalar nobody's key is related to (SHA-256("bn254-mi355x g2 comb blinding point"), reduced once)
  static const uint32_t blind[8] = {0x6b2f1c9du, 0x0f3a7e55u, 0x9c4d21a7u, 0x5be0cd19u, 0x1f83d9abu, 0x3c6ef372u, 0xa54ff53au, 0x2b67ae85u};
  {
    uint32_t b0[8];
    memcpy(b0, blind, sizeof b0);
    if (geq(b0)) sub_r(b0);
    put(65 * 8, b0);
    if (!g2) {                                           // G1: one blinding point per lane of the pair, and minus their sum
      uint32_t b1[8], sum[8], neg[8];
      memcpy(b1, b0, sizeof b1); add(b1, b0); add(b1, b0);                 // B1 = 3 b0 (any fixed scalar unrelated to a key will do)
      put(65 * 8 + 1, b1);
      memcpy(sum, b0, sizeof sum); add(sum, b1);
      uint64_t bw = 0;                                  // neg = r - sum
      for (int i = 0; i < 8; ++i) { uint64_t d = (uint64_t)R[i] - sum[i] - bw; neg[i] = (uint32_t)d; bw = (d >> 63) & 1; }
      put(65 * 8 + 2, neg);
    }
  }
  int rc;
  if ((rc = pool_reserve_one(c, g2 ? &c->g2_comb : &c->g1_comb, g2 ? 4 : 2, N))) { free(h); return rc; }
  // staging slots 5 .. 7: a host-pointer caller (mul_host) has ITS scalars in slots 0 .. 3 when this runs
  if ((rc = stage_in(c, 5, h, N * 32))) { free(h); return rc; }
  if ((rc = stage_reserve(c, 6, N * sz))) { free(h); return rc; }
  if ((rc = stage_reserve(c, 7, N))) { free(h); return rc; }
  if (g2) {
    k_g2_mul<<<grid_for(N), BN_WAVE, 0, c->stream>>>(nullptr, c->stage[5], N, 0, c->stage[6], c->stage[7]);
    k_pool_decode_g2<<<grid_for(N), BN_WAVE, 0, c->stream>>>(c->stage[6], N, 0, c->g2_comb);
  } else {
    k_g1_gen_mul<<<grid_for(N), BN_WAVE, 0, c->stream>>>(c->stage[5], N, c->stage[6]);
    k_pool_decode_g1<<<grid_for(N), BN_WAVE, 0, c->stream>>>(c->stage[6], N, 0, c->g1_comb);
  }
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);      // the staged scalars are read from `h`; the table is complete before any stream uses it
  free(h);
  if (e != hipSuccess) return -(int)e;
  if (g2) c->g2_comb_ready = 1; else c->g1_comb_ready = 1;
  return 0;
}
int bn254_batch_g2_mul_device(bn254_ctx* c, const uint8_t* d_p, const uint8_t* d_k, size_t n, int reduce, uint8_t* d_out, uint8_t* d_status, void* stream) {
  if (!c || (n && (!d_k || !d_out || !d_status))) return BN254_E_BAD_ARGUMENT;   // d_p == NULL: generator
  if (n == 0) return 0;
  if ((d_p && misaligned(d_p)) || misaligned(d_k) || misaligned(d_out)) return BN254_E_MISALIGNED;
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t s = stream ? (hipStream_t)stream : c->stream;
  if (!d_p && c->pair_lanes && c->g2_fixed_base) {
    // key derivation multiplies the FIXED generator: 65 table additions on a lane pair instead of the 256-step ladder on one lane
    int rc = comb_build(c, 1);
    if (rc) return rc;
    CallDone call_done(c, s);
    return bn254_pair_g2_mul_fixed(d_k, n, reduce, c->g2_comb, d_out, d_status, s);
  }
  CallDone call_done(c, s);
  k_g2_mul<<<grid_for(n), BN_WAVE, 0, s>>>(d_p, d_k, n, reduce, d_out, d_status);
  HIP_TRY(hipGetLastError());
  return 0;
}
static int mul_host(bn254_ctx* c, int g2, const uint8_t* p, const uint8_t* k, size_t n, int reduce, uint8_t* out, uint8_t* status) {
  if (!c || (n && (!k || !out || !status))) return BN254_E_BAD_ARGUMENT;             // p == NULL: the generator of the group
  if (n == 0) return 0;
  HIP_TRY(hipSetDevice(c->device));
  size_t sz = g2 ? 128 : 64;
  int rc;
  if (p && (rc = stage_in(c, 0, p, n * sz))) return rc;
  if ((rc = stage_in(c, 1, k, n * 32))) return rc;
  if ((rc = stage_reserve(c, 2, n * sz))) return rc;
  if ((rc = stage_reserve(c, 3, n))) return rc;
  rc = g2 ? bn254_batch_g2_mul_device(c, p ? c->stage[0] : nullptr, c->stage[1], n, reduce, c->stage[2], c->stage[3], nullptr)
          : bn254_batch_g1_mul_device(c, p ? c->stage[0] : nullptr, c->stage[1], n, reduce, c->stage[2], c->stage[3], nullptr);
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
  CallDone call_done(c, s);
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

// ---- aggregate verify: the pools' side (decode, H(m), subset-sum tables) and the tuples' side, apart -------------------------------------
// The tables depend on the POOLS and on how many tuples they are built for (thresholds of bn254_ws.h) — not on the tuples themselves.  A call
// with raw pools builds them and runs; bn254_ctx_register_pools builds them ONCE for a caller whose pools are fixed (a validator set signing
// a stream of messages' worth of tuples), and bn254_batch_aggregate_verify_registered only walks tuples — what bn254_ctx_register_keys is to
// the verify path (/root/reference/src/types.rs:126-132, :264-270: the sums; src/ecdsa.rs:49-64: the check).
static int agg_build_tables(bn254_ctx* c, hipStream_t s, const uint8_t* d_msgs, const uint64_t* d_msg_off, size_t n_msgs, const uint8_t* d_pk_pool,
                            size_t n_signers, const uint8_t* d_sig_pool, uint32_t flags, size_t n, AggTables* t) {
  int rc;
  t->valid = 0; t->n_msgs = n_msgs; t->n_signers = n_signers; t->n_groups = 0; t->groups4 = 0; t->wide2 = 0; t->wide1 = 0; t->built_for = n;
  if ((rc = ws_reserve(c, n > n_msgs ? n : n_msgs))) return rc;
  if ((rc = pool_reserve(c, 0, 4, n_signers))) return rc;
  if ((rc = pool_reserve(c, 1, 2, n_msgs * n_signers))) return rc;
  if ((rc = pool_reserve(c, 2, 2, n_msgs))) return rc;
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
          const size_t groups2 = 2 * groups4;
          if (pool_reserve(c, 7, 2, n_msgs * groups2 * 4) == 0) {          // pairs first, then quads by batched affine additions
            k_pool_pairs_g1<<<grid_for(n_msgs * groups2), BN_WAVE, 0, s>>>(c->pool[1], n_signers, groups2, n_msgs, c->pool[7]);
            k_pool_quads_g1<<<grid_for(n_msgs * groups4 * 4), BN_WAVE, 0, s>>>(c->pool[7], groups2, groups4, n_msgs, c->pool[4]);
          } else {
            (void)hipGetLastError();
            k_pool_subsets_g1<<<grid_for(entries), BN_WAVE, 0, s>>>(c->pool[1], n_signers, groups4, n_msgs, c->pool[4]);
          }
        } else {
          (void)hipGetLastError();     // no HBM for the table: the signatures are added one by one (groups4 = 0), same statuses
        }
      }
    }
    // the largest batches: tables of twice the window, built from the ones above by one batched affine addition per entry
    // (k_pool_widen_*): half the additions per tuple.  A table that does not fit its budget (or HBM) is simply not used.
    if (n_groups != 0 && c->agg_wide_min_tuples > 0 && n >= (size_t)c->agg_wide_min_tuples) {
      const size_t n_chunks = (n_groups + 1) / 2;
      const size_t e2 = n_chunks * 65536, bytes2 = e2 * (2 * BN_POOL_HALF_WORDS * sizeof(int32_t) + 1);
      if (bytes2 <= AGG_WIDE_G2_MAX_BYTES) {
        if (pool_reserve(c, 5, 4, e2) == 0) {
          k_pool_widen_g2<<<grid_for(n_chunks * 256 * (256 / BN_WIDEN_G2_NLO)), BN_WAVE, 0, s>>>(c->pool[3], n_groups, n_chunks, c->pool[5]);
          t->wide2 = 1;
        } else {
          (void)hipGetLastError();
        }
      }
      const size_t e1 = n_msgs * n_groups * 256, bytes1 = ((e1 + 255) & ~(size_t)255) * (BN_POOL_HALF_WORDS * sizeof(int32_t) + 1);
      if (groups4 != 0 && n >= AGG_WIDE_G1_TUPLES_PER_MSG * n_msgs && bytes1 <= AGG_SUBSET_G1_MAX_BYTES) {
        if (pool_reserve(c, 6, 2, e1) == 0) {
          k_pool_widen_g1<<<grid_for(n_msgs * n_groups * 16), BN_WAVE, 0, s>>>(c->pool[4], groups4, n_groups, n_msgs, c->pool[6]);
          t->wide1 = 1;
        } else {
          (void)hipGetLastError();
        }
      }
    }
    t->n_groups = n_groups; t->groups4 = groups4;
  }
  HIP_TRY(hipGetLastError());
  t->valid = 1;
  return 0;
}
// the tuples' side: (optionally) bucket them by message, the aggregation kernel on the tables named by `t`, then the verify kernels
static int agg_run(bn254_ctx* c, hipStream_t s, const AggTables& t, const uint32_t* d_tuple_msg, const uint64_t* d_tuple_off, const uint32_t* d_signer_idx,
                   size_t n, uint8_t* d_status) {
  int rc;
  const size_t n_msgs = t.n_msgs, n_signers = t.n_signers;
  if (c->pair_lanes) {
    // with the per-message signature tables in use: bucket the tuples by message (see k_agg_sort_count).  The hash rounds of the
    // messages are done with ws.h_list (2 x stride words): its first n words take the index map, the counters sit behind.
    const uint32_t* perm = nullptr;
    if (t.groups4 != 0 && c->agg_sort_by_msg && n >= 4 * n_msgs && n <= 0xFFFFFFFFull && n_msgs < 0xFFFFFFFFull && c->ws.stride >= 2 * (n_msgs + 1)) {
      uint32_t* map = c->ws.h_list;
      uint32_t* cnt = c->ws.h_list + c->ws.stride;
      uint32_t* cursor = cnt + (n_msgs + 1);
      HIP_TRY(hipMemsetAsync(cnt, 0, sizeof(uint32_t) * (n_msgs + 1), s));
      k_agg_sort_count<<<grid_for(n), BN_WAVE, 0, s>>>(d_tuple_msg, n, (uint32_t)n_msgs, cnt);
      k_agg_sort_scan<<<1, BN_WAVE, 0, s>>>((uint32_t)n_msgs + 1, cnt, cursor);
      k_agg_sort_scatter<<<grid_for(n), BN_WAVE, 0, s>>>(d_tuple_msg, n, (uint32_t)n_msgs, cursor, map);
      perm = map;
    }
    PROF_MARK(1);
    if ((rc = bn254_pair_aggregate(d_tuple_msg, d_tuple_off, d_signer_idx, n, n_signers, n_msgs, c->pool[0], c->pool[1], c->pool[2], c->pool[3], t.n_groups,
                                   c->pool[4], t.groups4, c->ws, s, perm, t.wide2 ? &c->pool[5] : nullptr, t.wide1 ? &c->pool[6] : nullptr))) return rc;
  } else {
    PROF_MARK(1);
    k_aggregate<<<grid_for(n), BN_WAVE, 0, s>>>(d_tuple_msg, d_tuple_off, d_signer_idx, n, n_signers, n_msgs, c->pool[0], c->pool[1], c->pool[2], c->ws);
  }
  PROF_MARK(2);
  if (c->pair_lanes) {
    // the aggregated tuples are verify-shaped: batches that cannot fill the chip take the small-batch kernels (one aggregate verify: the
    // pairing part 9.8 -> 1.0 ms)
    if ((rc = launch_pair_or_trio(c, s, n, 1, d_status, 0, true))) return rc;
  } else {
    { int rc_ = launch_miller_verify_lane(c, s, n, nullptr, nullptr); if (rc_) return rc_; }
    PROF_MARK(3);
    { int rc_ = launch_final_exp_lane(c, s, n, 1, 1, 1, 1, nullptr, d_status, 0, 0, nullptr, nullptr); if (rc_) return rc_; }
  }
  PROF_MARK(4);
  if (c->profiling) { c->ev_valid = 1; c->ev_hash_first = 0; }
  HIP_TRY(hipGetLastError());
  return 0;
}

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
  hipStream_t s = stream ? (hipStream_t)stream : c->stream;
  c->reg_pools.valid = 0;                              // raw pools overwrite the context's pool buffers: a registration does not survive them
  AggTables t;
  int rc = ws_reserve(c, n > n_msgs ? n : n_msgs);
  if (rc) return rc;
  CallDone call_done(c, s);
  PROF_MARK(0);                                        // ms[0] = pools (decode, hash of the messages, subset-sum tables), ms[1] = the aggregation kernel
  if ((rc = agg_build_tables(c, s, d_msgs, d_msg_off, n_msgs, d_pk_pool, n_signers, d_sig_pool, flags, n, &t))) return rc;
  return agg_run(c, s, t, d_tuple_msg, d_tuple_off, d_signer_idx, n, d_status);
}
// Registered pools: decode, H(m) and every subset-sum table built ONCE, for batches of about `expect_tuples` tuples (the table thresholds of
// bn254_ws.h are applied to that figure); the tables stay in the context until the next registration or the next call with raw pools.
int bn254_ctx_register_pools_device(bn254_ctx* c, const uint8_t* d_msgs, const uint64_t* d_msg_off, size_t n_msgs, const uint8_t* d_pk_pool, size_t n_signers,
                                    const uint8_t* d_sig_pool, uint32_t flags, size_t expect_tuples, void* stream) {
  MsgsLenScope msgs_len_scope(c);
  if (!c || !n_msgs || !n_signers || !d_msgs || !d_msg_off || !d_pk_pool || !d_sig_pool) return BN254_E_BAD_ARGUMENT;
  if (misaligned(d_pk_pool) || misaligned(d_sig_pool) || ((uintptr_t)d_msg_off & 7u)) return BN254_E_MISALIGNED;
  HIP_TRY(hipSetDevice(c->device));
  { int rc_ = ctx_quiesce(c); if (rc_) return rc_; }   // no aggregate verify may still be reading the previous tables
  hipStream_t s = stream ? (hipStream_t)stream : c->stream;
  c->reg_pools.valid = 0;
  CallDone call_done(c, s);
  return agg_build_tables(c, s, d_msgs, d_msg_off, n_msgs, d_pk_pool, n_signers, d_sig_pool, flags, expect_tuples ? expect_tuples : 1, &c->reg_pools);
}
int bn254_ctx_register_pools(bn254_ctx* c, const uint8_t* msgs, const uint64_t* msg_off, size_t n_msgs, const uint8_t* pk_pool, size_t n_signers,
                             const uint8_t* sig_pool, uint32_t flags, size_t expect_tuples) {
  MsgsLenScope msgs_len_scope(c);
  if (!c || !n_msgs || !n_signers || !msg_off || !pk_pool || !sig_pool) return BN254_E_BAD_ARGUMENT;
  if (!offsets_ok(msg_off, n_msgs) || (msg_off[n_msgs] && !msgs)) return BN254_E_BAD_ARGUMENT;
  HIP_TRY(hipSetDevice(c->device));
  int rc;
  if ((rc = stage_in(c, 0, msgs, (size_t)msg_off[n_msgs]))) return rc;
  if ((rc = stage_in(c, 1, msg_off, (n_msgs + 1) * sizeof(uint64_t)))) return rc;
  if ((rc = stage_in(c, 2, pk_pool, n_signers * 128))) return rc;
  if ((rc = stage_in(c, 3, sig_pool, n_msgs * n_signers * 64))) return rc;
  rc = bn254_ctx_register_pools_device(c, c->stage[0], (const uint64_t*)c->stage[1], n_msgs, c->stage[2], n_signers, c->stage[3], flags, expect_tuples, nullptr);
  hipError_t e = hipStreamSynchronize(c->stream);     // also on failure: the staged copies read the caller's buffers
  return rc ? rc : -(int)e;
}
int bn254_batch_aggregate_verify_registered_device(bn254_ctx* c, const uint32_t* d_tuple_msg, const uint64_t* d_tuple_off, const uint32_t* d_signer_idx, size_t n,
                                                   uint8_t* d_status, void* stream) {
  if (!c || (n && (!d_tuple_msg || !d_tuple_off || !d_signer_idx || !d_status))) return BN254_E_BAD_ARGUMENT;
  if (!c->reg_pools.valid) return BN254_E_BAD_ARGUMENT;        // nothing registered (or a call with raw pools has replaced the tables since)
  if (n == 0) return 0;
  if (misaligned(d_tuple_msg) || misaligned(d_signer_idx) || ((uintptr_t)d_tuple_off & 7u)) return BN254_E_MISALIGNED;
  HIP_TRY(hipSetDevice(c->device));
  int rc = ws_reserve(c, n > c->reg_pools.n_msgs ? n : c->reg_pools.n_msgs);
  if (rc) return rc;
  hipStream_t s = stream ? (hipStream_t)stream : c->stream;
  CallDone call_done(c, s);
  PROF_MARK(0);                                        // ms[0] = 0: the pools' side was paid at registration
  return agg_run(c, s, c->reg_pools, d_tuple_msg, d_tuple_off, d_signer_idx, n, d_status);
}
int bn254_batch_aggregate_verify_registered(bn254_ctx* c, const uint32_t* tuple_msg, const uint64_t* tuple_off, const uint32_t* signer_idx, size_t n, uint8_t* status) {
  if (!c || (n && (!tuple_msg || !tuple_off || !signer_idx || !status))) return BN254_E_BAD_ARGUMENT;
  if (!c->reg_pools.valid) return BN254_E_BAD_ARGUMENT;
  if (n == 0) return 0;
  if (!offsets_ok(tuple_off, n)) return BN254_E_BAD_ARGUMENT;
  HIP_TRY(hipSetDevice(c->device));
  int rc;
  if ((rc = stage_in(c, 4, tuple_msg, n * sizeof(uint32_t)))) return rc;
  if ((rc = stage_in(c, 5, tuple_off, (n + 1) * sizeof(uint64_t)))) return rc;
  if ((rc = stage_in(c, 6, signer_idx, (size_t)tuple_off[n] * sizeof(uint32_t)))) return rc;
  if ((rc = stage_reserve(c, 7, n))) return rc;
  rc = bn254_batch_aggregate_verify_registered_device(c, (const uint32_t*)c->stage[4], (const uint64_t*)c->stage[5], (const uint32_t*)c->stage[6], n, c->stage[7], nullptr);
  if (!rc) rc = stage_out(c, 7, status, n);
  hipError_t e = hipStreamSynchronize(c->stream);
  return rc ? rc : -(int)e;
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


}  // extern "C"
