// G1 / G2 group arithmetic (Jacobian coordinates), generic over the coordinate field.
//
// Replaces, for the hot path, `bn::{G1,G2}` point addition / doubling / scalar multiplication
// as used by aggregation (/root/reference/src/types.rs:126-132, :264-270), signing and key
// derivation (/root/reference/src/ecdsa.rs:31, /root/reference/src/types.rs:86, :156) and the
// curve / subgroup checks of the decoders (/root/reference/src/utils.rs:113, :125).
#pragma once
#include "bn254_field.h"

namespace bn254 {

enum Status : uint8_t {   // 1 + index of the variant in /root/reference/src/error.rs:6-29
  ST_OK = 0, ST_HASH_TO_POINT = 1, ST_INDEX_OOB = 2, ST_INVALID_ENCODING = 3, ST_INVALID_GROUP_POINT = 4,
  ST_INVALID_LENGTH = 5, ST_NOT_MEMBER = 6, ST_TO_AFFINE = 7, ST_POINT_IN_JACOBIAN = 8,
  ST_VERIFICATION_FAILED = 9, ST_SERIALIZATION = 10, ST_HEX_DECODE = 11
};
enum Flags : uint32_t { FLAG_G2_SUBGROUP_CHECK = 1u, FLAG_REJECT_IDENTITY = 2u };

// field traits so one template serves G1 (Fq) and G2 (Fq2)
BN_DEV Fp f_add(const Fp& a, const Fp& b) { return fp_add(a, b); }
BN_DEV Fp f_sub(const Fp& a, const Fp& b) { return fp_sub(a, b); }
BN_DEV Fp f_dbl(const Fp& a) { return fp_dbl(a); }
BN_DEV Fp f_mul(const Fp& a, const Fp& b) { return fp_mul(a, b); }
BN_DEV Fp f_sqr(const Fp& a) { return fp_sqr(a); }
BN_DEV Fp f_neg(const Fp& a) { return fp_neg(a); }
BN_DEV bool f_is_zero(const Fp& a) { return fp_is_zero(a); }
BN_DEV bool f_eq(const Fp& a, const Fp& b) { return fp_eq(a, b); }
BN_DEV Fp f_select(bool c, const Fp& a, const Fp& b) { return fp_select(c, a, b); }
BN_DEV Fp f_inv(const Fp& a) { return fp_inv(a); }
BN_DEV void f_set_one(Fp& a) { a = fp_one(); }
BN_DEV void f_set_zero(Fp& a) { a = fp_zero(); }
BN_DEV Fp f_norm(const Fp& a) { return fp_norm(a); }
BN_DEV Fp f_mul8(const Fp& a) { return fp_mul8_spread(a); }
BN_DEV Fp f_reduce(const Fp& a) { return fp_reduce_weak(a); }

BN_DEV Fp2 f_add(const Fp2& a, const Fp2& b) { return fp2_add(a, b); }
BN_DEV Fp2 f_sub(const Fp2& a, const Fp2& b) { return fp2_sub(a, b); }
BN_DEV Fp2 f_dbl(const Fp2& a) { return fp2_dbl(a); }
BN_DEV Fp2 f_mul(const Fp2& a, const Fp2& b) { return fp2_mul(a, b); }
BN_DEV Fp2 f_sqr(const Fp2& a) { return fp2_sqr(a); }
BN_DEV Fp2 f_neg(const Fp2& a) { return fp2_neg(a); }
BN_DEV bool f_is_zero(const Fp2& a) { return fp2_is_zero(a); }
BN_DEV bool f_eq(const Fp2& a, const Fp2& b) { return fp2_eq(a, b); }
BN_DEV Fp2 f_select(bool c, const Fp2& a, const Fp2& b) { return fp2_select(c, a, b); }
BN_DEV Fp2 f_inv(const Fp2& a) { return fp2_inv(a); }
BN_DEV void f_set_one(Fp2& a) { a = fp2_one(); }
BN_DEV void f_set_zero(Fp2& a) { a = fp2_zero(); }
BN_DEV Fp2 f_norm(const Fp2& a) { return fp2_norm(a); }
BN_DEV Fp2 f_mul8(const Fp2& a) { return fp2_mul8(a); }
BN_DEV Fp2 f_reduce(const Fp2& a) { return fp2_reduce_weak(a); }

template <class F> struct Affine { F x, y; bool inf; };
template <class F> struct Jac { F x, y, z; };   // z == 0 <=> identity
typedef Affine<Fp> G1Affine;
typedef Affine<Fp2> G2Affine;
typedef Jac<Fp> G1Jac;
typedef Jac<Fp2> G2Jac;

template <class F> BN_DEV void jac_set_identity(Jac<F>& r) { f_set_one(r.x); f_set_one(r.y); f_set_zero(r.z); }
template <class F> BN_DEV bool jac_is_identity(const Jac<F>& p) { return f_is_zero(p.z); }
template <class F> BN_DEV void jac_from_affine(Jac<F>& r, const Affine<F>& p) {
  if (p.inf) { jac_set_identity(r); return; }
  r.x = p.x; r.y = p.y; f_set_one(r.z);
}
template <class F> BN_DEV void jac_select(Jac<F>& r, bool c, const Jac<F>& a, const Jac<F>& b) {
  r.x = f_select(c, a.x, b.x); r.y = f_select(c, a.y, b.y); r.z = f_select(c, a.z, b.z);
}

// dbl-2009-l (a = 0); the identity (z = 0) maps to z = 0 without a branch.
// Coordinates in and out are tight (carried limbs).  Products and squares give tight results; f_norm stands where a
// lazy sum meets a product whose 64-bit columns it would overflow (an Fq2 square takes a tight operand, an Fq2
// product limb magnitudes with A * B <= 6 units of 2^28 — bn254_field.h; tests/test_bounds.py proves every flow).
template <class F> BN_DEV void jac_dbl_body(Jac<F>& r, const Jac<F>& p) {
  F a = f_sqr(p.x), b = f_sqr(p.y), c = f_sqr(b);
  F d = f_norm(f_dbl(f_sub(f_sub(f_sqr(f_norm(f_add(p.x, b))), a), c)));
  F e = f_norm(f_add(f_dbl(a), a)), f = f_sqr(e);
  F x3 = f_reduce(f_sub(f, f_dbl(d)));                                // f_reduce: carry + weak reduction — with R / q = 169 the
  F z3 = f_norm(f_dbl(f_mul(p.y, p.z)));                              // coordinates of a long chain of group operations would
  F y3 = f_reduce(f_sub(f_mul(e, f_norm(f_sub(d, x3))), f_mul8(c)));  // otherwise grow; 8 C through the limb-crossing shift
  r.x = x3; r.y = y3; r.z = z3;
}
template <class F> BN_DEVN void jac_dbl(Jac<F>& r, const Jac<F>& p) { jac_dbl_body(r, p); }
// the same IN PLACE on an accumulator the caller keeps in LDS (the subgroup ladder of the lane-pair decoders): as a real function taking
// its operand by reference the doubling moved 27 words in and out of the private segment per call
template <class F> BN_DEVN void jac_dbl_lds(Jac<F>& acc) {
  BN_ASSUME_LDS(&acc);
  Jac<F> o;
  jac_dbl_body(o, acc);
  acc = o;
}

#if defined(__HIPCC__)
#define BN_WAVE_ANY(x) (__any((int)(x)) != 0)
#else
#define BN_WAVE_ANY(x) (x)
#endif
// add-2007-bl with every exceptional case resolved by selects (lanes never diverge):
// P = O -> Q, Q = O -> P, P = Q -> 2P, P = -Q -> O.
// Round 6: the doubling that serves P = Q is computed only when some lane of the wave needs it (a wave vote; it was 7 of the 23 products of
// EVERY addition).  In the windowed ladders below the accumulator is 16 x (prefix) x P and the entry d P with d <= 8: they coincide only
// for a scalar built around the group order — nothing an honest caller holds, and nothing anyone can aim at without knowing the scalar —,
// so a ladder's schedule still does not depend on its (secret) scalar.
template <class F> BN_DEVN void jac_add(Jac<F>& r, const Jac<F>& p, const Jac<F>& q) {
  F z1z1 = f_sqr(p.z), z2z2 = f_sqr(q.z);
  F u1 = f_mul(p.x, z2z2), u2 = f_mul(q.x, z1z1);
  F s1 = f_mul(f_mul(p.y, q.z), z2z2), s2 = f_mul(f_mul(q.y, p.z), z1z1);
  F h = f_norm(f_sub(u2, u1)), i = f_norm(f_dbl(f_dbl(f_sqr(h)))), j = f_mul(h, i);
  F rr = f_norm(f_dbl(f_sub(s2, s1))), v = f_mul(u1, i);
  Jac<F> o;
  o.x = f_reduce(f_sub(f_sub(f_sqr(rr), j), f_dbl(v)));
  o.y = f_reduce(f_sub(f_mul(rr, f_norm(f_sub(v, o.x))), f_dbl(f_mul(s1, j))));
  o.z = f_mul(f_norm(f_sub(f_sub(f_sqr(f_norm(f_add(p.z, q.z))), z1z1), z2z2)), h);
  bool p_inf = f_is_zero(p.z), q_inf = f_is_zero(q.z);
  bool same_x = f_is_zero(h), same_y = f_is_zero(rr);
  const bool same_point = same_x && same_y && !p_inf && !q_inf;
  Jac<F> d = p;
  if (BN_WAVE_ANY(same_point)) jac_dbl(d, p);
  Jac<F> id;
  jac_set_identity(id);
  // generic result, then overrides in increasing priority
  jac_select(o, same_point, d, o);
  jac_select(o, same_x && !same_y, id, o);
  jac_select(o, q_inf, p, o);
  jac_select(o, p_inf, q, o);
  r = o;
}

// mixed addition P + (x2, y2) with an affine second operand (madd-2007-bl, 7M + 4S): the inner
// operation of signature / public-key aggregation (/root/reference/src/types.rs:126-132, :264-270).
// Exceptional cases by selects, as in jac_add: P = O -> Q, P = Q -> 2Q, P = -Q -> O; q_inf skips.
template <class F> BN_DEVN void jac_madd(Jac<F>& r, const Jac<F>& p, const Affine<F>& q) {
  F z1z1 = f_sqr(p.z);
  F u2 = f_mul(q.x, z1z1);
  F s2 = f_mul(f_mul(q.y, p.z), z1z1);
  F h = f_norm(f_sub(u2, p.x)), hh = f_sqr(h);
  F i = f_norm(f_dbl(f_dbl(hh))), j = f_mul(h, i);
  F rr = f_norm(f_dbl(f_sub(s2, p.y))), v = f_mul(p.x, i);
  Jac<F> o;
  o.x = f_reduce(f_sub(f_sub(f_sqr(rr), j), f_dbl(v)));
  o.y = f_reduce(f_sub(f_mul(rr, f_norm(f_sub(v, o.x))), f_dbl(f_mul(p.y, j))));
  o.z = f_norm(f_sub(f_sub(f_sqr(f_norm(f_add(p.z, h))), z1z1), hh));
  bool p_inf = f_is_zero(p.z);
  bool same_x = f_is_zero(h), same_y = f_is_zero(rr);
  Jac<F> qj, d, id;
  qj.x = q.x; qj.y = q.y; f_set_one(qj.z);
  jac_dbl(d, qj);
  jac_set_identity(id);
  jac_select(o, same_x && same_y, d, o);
  jac_select(o, same_x && !same_y, id, o);
  jac_select(o, p_inf, qj, o);
  jac_select(o, q.inf, p, o);
  r = o;
}

// Mixed addition for running sums (aggregation): the common case without the doubling that jac_madd computes for
// every call just to be able to select it (40 % of its cost), and a flag for the rare lanes that did hit
// P = +-Q; the caller redoes the step with jac_madd when any lane of the wave raised it.
template <class F> BN_DEV void jac_madd_common_body(Jac<F>& r, bool& exceptional, const Jac<F>& p, const Affine<F>& q) {
  F z1z1 = f_sqr(p.z);
  F u2 = f_mul(q.x, z1z1);
  F s2 = f_mul(f_mul(q.y, p.z), z1z1);
  F h = f_norm(f_sub(u2, p.x)), hh = f_sqr(h);
  F i = f_norm(f_dbl(f_dbl(hh))), j = f_mul(h, i);
  F rr = f_norm(f_dbl(f_sub(s2, p.y))), v = f_mul(p.x, i);
  Jac<F> o;
  o.x = f_reduce(f_sub(f_sub(f_sqr(rr), j), f_dbl(v)));
  o.y = f_reduce(f_sub(f_mul(rr, f_norm(f_sub(v, o.x))), f_dbl(f_mul(p.y, j))));
  o.z = f_norm(f_sub(f_sub(f_sqr(f_norm(f_add(p.z, h))), z1z1), hh));
  const bool p_inf = f_is_zero(p.z);
  Jac<F> qj;
  qj.x = q.x; qj.y = q.y; f_set_one(qj.z);
  exceptional = f_is_zero(h) && !p_inf && !q.inf;
  jac_select(o, p_inf, qj, o);
  jac_select(o, q.inf, p, o);
  r = o;
}
template <class F> BN_DEVN void jac_madd_common(Jac<F>& r, bool& exceptional, const Jac<F>& p, const Affine<F>& q) { jac_madd_common_body(r, exceptional, p, q); }
// acc += q with identical control flow across the wave
template <class F> BN_DEV void jac_accumulate(Jac<F>& acc, const Affine<F>& q) {
  Jac<F> t;
  bool ex;
  jac_madd_common(t, ex, acc, q);
  if (BN_WAVE_ANY(ex)) jac_madd(acc, acc, q);   // some lane met P = +-Q: the complete formula for the whole wave (rare)
  else acc = t;
}

// acc += q IN PLACE in the lanes of the common case; returns true in the lanes that met P = +-Q, whose accumulator is left as
// it was.  For accumulators the caller keeps in LDS: as a real function the addition takes its operands through memory, and
// with the accumulator in the private segment that memory is HBM — in the aggregation kernel, whose loops are nothing but
// accumulations, 700 GB per 1 Mi tuples (7 TB/s: the kernel ran at the HBM roofline of its own argument passing,
// profiles/r03_h_pmc.json); inlining the formula instead made it spill more than the calls moved (114 vs 102 ms).
template <class F> BN_DEVN bool jac_madd_inplace(Jac<F>& acc, const Affine<F>& q) {
  Jac<F> o;
  bool ex;
  jac_madd_common_body(o, ex, acc, q);
  if (!ex) acc = o;
  return ex;
}
template <class F> BN_DEV void jac_accumulate_mem(Jac<F>& acc, Affine<F> q) {
  const bool ex = jac_madd_inplace(acc, q);
  if (BN_WAVE_ANY(ex)) {                            // rare: the complete formula for the lanes that need it, nothing for the others
    q.inf = q.inf || !ex;
    jac_madd(acc, acc, q);
  }
}

// The same with the affine operand PRODUCED INSIDE the function by `src(q)` (a small functor passed by value, i.e. in registers: a
// table-record pointer and an identity flag in the aggregation kernel).  An Affine passed by reference to a real function lives in the
// caller's private segment: 19 dwords written by the caller and read back by the callee per addition — in k_aggregate_pair 47 GB of
// writes per 1 Mi tuples and a store -> load round trip on the critical path of every call (profiles/r04_z_pmc.json).
// inlined into its (one) loop: with the sums in LDS and the operand fetched inside, what stays alive across the addition in
// k_aggregate_pair is a handful of pointers — 43.25 -> 41.35 ms per 1 Mi tuples same box (profiles/r05_b_ab_aggregate.log); -DBN_AGG_CALL_MADD
// restores the call
#if defined(BN_AGG_CALL_MADD)
#define BN_DEV_MADD_FROM BN_DEVN
#else
#define BN_DEV_MADD_FROM BN_DEV
#endif
template <class F, class Src> BN_DEV_MADD_FROM bool jac_madd_inplace_from(Jac<F>& acc, Src src) {
#if defined(BN_MADD_FROM_SELECTS)
  Affine<F> q;
  src(q);
  Jac<F> o;
  bool ex;
  jac_madd_common_body(o, ex, acc, q);
  if (!ex) acc = o;
  return ex;
#else
  // The formula of jac_madd_common_body STREAMED: every coordinate of the accumulator is read where it is needed and written back as soon as
  // it is final (z first, then x and y), so that at most four or five field elements are alive at once — the form with one result
  // triple and selects at the end kept q, the old and the new point alive together and spilled 35 registers per addition once inlined
  // (k_aggregate_pair: 26 GB of private-segment writes per 1 Mi tuples).  The lanes that must not take the common formula write nothing:
  // q = O keeps the accumulator; P = O and P = +-Q are reported to the caller, which redoes them with the complete formula.
  Affine<F> q;
  src(q);
  const F z = acc.z, z1z1 = f_sqr(z);
  const F u2 = f_mul(q.x, z1z1), s2 = f_mul(f_mul(q.y, z), z1z1);
  const bool q_inf = q.inf;
  const F x = acc.x;
  const F h = f_norm(f_sub(u2, x));
  const bool ex = (f_is_zero(h) || f_is_zero(z)) && !q_inf;
  const bool wr = !ex && !q_inf;
  const F hh = f_sqr(h), i = f_norm(f_dbl(f_dbl(hh)));
  {
    const F z3 = f_norm(f_sub(f_sub(f_sqr(f_norm(f_add(z, h))), z1z1), hh));
    if (wr) acc.z = z3;
  }
  const F j = f_mul(h, i);
  const F y = acc.y;
  const F rr = f_norm(f_dbl(f_sub(s2, y))), v = f_mul(x, i);
  const F x3 = f_reduce(f_sub(f_sub(f_sqr(rr), j), f_dbl(v)));
  const F y3 = f_reduce(f_sub(f_mul(rr, f_norm(f_sub(v, x3))), f_dbl(f_mul(y, j))));
  if (wr) { acc.x = x3; acc.y = y3; }
  return ex;
#endif
}
template <class F, class Src> BN_DEV void jac_accumulate_from(Jac<F>& acc, Src src) {
  const bool ex = jac_madd_inplace_from(acc, src);
  if (BN_WAVE_ANY(ex)) {                            // rare: the complete formula for the lanes that need it, nothing for the others
    Affine<F> q;
    src(q);
    q.inf = q.inf || !ex;
    jac_madd(acc, acc, q);
  }
}

// P + Q for operands known to satisfy P != +-Q unless one of them is the identity (add-2007-bl without
// the doubling / cancellation overrides of jac_add, which cost a jac_dbl per call).
template <class F> BN_DEVN void jac_add_distinct(Jac<F>& r, const Jac<F>& p, const Jac<F>& q) {
  F z1z1 = f_sqr(p.z), z2z2 = f_sqr(q.z);
  F u1 = f_mul(p.x, z2z2), u2 = f_mul(q.x, z1z1);
  F s1 = f_mul(f_mul(p.y, q.z), z2z2), s2 = f_mul(f_mul(q.y, p.z), z1z1);
  F h = f_norm(f_sub(u2, u1)), i = f_norm(f_dbl(f_dbl(f_sqr(h)))), j = f_mul(h, i);
  F rr = f_norm(f_dbl(f_sub(s2, s1))), v = f_mul(u1, i);
  Jac<F> o;
  o.x = f_reduce(f_sub(f_sub(f_sqr(rr), j), f_dbl(v)));
  o.y = f_reduce(f_sub(f_mul(rr, f_norm(f_sub(v, o.x))), f_dbl(f_mul(s1, j))));
  o.z = f_mul(f_norm(f_sub(f_sub(f_sqr(f_norm(f_add(p.z, q.z))), z1z1), z2z2)), h);
  bool p_inf = f_is_zero(p.z), q_inf = f_is_zero(q.z);
  jac_select(o, q_inf, p, o);
  jac_select(o, p_inf, q, o);
  r = o;
}

// k * P for a per-lane scalar k < 2^(32*WORDS) (little-endian words): signed fixed 4-bit windows (digits in
// [-8, 8]) over the table P..8P — 32*WORDS doublings + 8*WORDS+1 additions, identical control flow in every lane.
// COMPLETE = false: P in a group of prime order r > 2^253 (G1: cofactor 1) and k < 2^128.  While a window is
//   added the accumulator is 16 * (a prefix < 2^125) * P, never +-(digit * P) with digit <= 8 unless it is the
//   identity, so the additions cannot hit the doubling case and jac_add_distinct applies (randomised batch
//   verification: r_i * H(m_i), r_i * sig_i).
// COMPLETE = true: any point, any scalar (used as-is, not reduced — /root/reference/src/bn256.json:54-159 has
//   scalars up to 2^256-1): every addition is the complete jac_add.  Signing and key derivation
//   (/root/reference/src/ecdsa.rs:31, src/types.rs:86, :156): 3.7 k products against 8.2 k for the
//   bit-by-bit ladder this replaces.
// entry |d| of the window table, negated for d < 0, the identity for d = 0.  The table lives in the lane's private segment.
// SECRET = false: the entry is INDEXED per lane (27 words read) — for the throw-away random multipliers of the randomised
//   verifies (jac_mul_u128 / jac_mul_u64 / g1_mul_glv), which are no secret: the scan below was most of those kernels'
//   private-segment traffic (216 words per window).
// SECRET = true: every entry is read and the wanted one kept by selects, so neither the addresses nor the control flow
//   depend on the digit — jac_mul is ECDSA::sign (scalar = the private key, /root/reference/src/ecdsa.rs:31) and
//   PublicKey::from_private_key (src/types.rs:86, :156).
template <bool SECRET, class F> BN_DEV void jac_window_entry(Jac<F>& t, const Jac<F>* tab, int d) {
  const int m = d < 0 ? -d : d;
  if constexpr (SECRET) {
    t = tab[0];
#pragma unroll 1
    for (int j = 1; j < 8; ++j) jac_select(t, m == j + 1, tab[j], t);
  } else {
    t = tab[m == 0 ? 0 : m - 1];
  }
  Jac<F> id;
  jac_set_identity(id);
  jac_select(t, m == 0, id, t);
  t.y = f_select(d < 0, f_neg(t.y), t.y);
}
// `r` is the running accumulator itself (doubled and added to in place through the reference): a caller that hands over an LDS
// slot keeps the 4 doublings + 1 addition per window out of the private segment
template <int WORDS, bool COMPLETE, class F> BN_DEVN void jac_mul_window(Jac<F>& r, const Affine<F>& p, const uint32_t* k) {
  constexpr int NW = 8 * WORDS;
  Jac<F> tab[8], t;
  Jac<F>& acc = r;
  jac_from_affine(tab[0], p);
  jac_dbl(tab[1], tab[0]);
  for (int j = 2; j < 8; ++j) {
    if constexpr (COMPLETE) jac_add(tab[j], tab[j - 1], tab[0]); else jac_add_distinct(tab[j], tab[j - 1], tab[0]);
  }
  signed char digit[NW + 1];
  int carry = 0;
  for (int j = 0; j < NW; ++j) {
    int v = (int)((k[j >> 3] >> (4 * (j & 7))) & 15u) + carry;
    carry = v > 8;
    digit[j] = (signed char)(v - 16 * carry);
  }
  digit[NW] = (signed char)carry;
  jac_set_identity(acc);
  for (int j = NW; j >= 0; --j) {
    if (j != NW) { jac_dbl(acc, acc); jac_dbl(acc, acc); jac_dbl(acc, acc); jac_dbl(acc, acc); }
    jac_window_entry<COMPLETE>(t, tab, (int)digit[j]);      // COMPLETE = the 256-bit multiplication of sign / keygen: secret scalar
    if constexpr (COMPLETE) jac_add(acc, acc, t); else jac_add_distinct(acc, acc, t);
  }
}
template <class F> BN_DEV void jac_mul_u128(Jac<F>& r, const Affine<F>& p, const uint32_t* k) { jac_mul_window<4, false>(r, p, k); }
template <class F> BN_DEV void jac_mul_u64(Jac<F>& r, const Affine<F>& p, const uint32_t* k) { jac_mul_window<2, false>(r, p, k); }
// k * P, 256-bit scalar, any point
template <class F> BN_DEV void jac_mul(Jac<F>& r, const Affine<F>& p, const uint32_t* k) { jac_mul_window<8, true>(r, p, k); }

// (k1 + k2 * lambda) * P for 64-bit k1, k2 on G1 through the endomorphism phi(x, y) = (beta x, y) = lambda (x, y):
// k1 * P + k2 * phi(P) by one joint ladder — 64 doublings and 2 x 17 additions over the tables j*P and
// phi(j*P) = (beta X, Y, Z), j = 1..8 — instead of 128 doublings + 33 additions for a 128-bit scalar.
// (k1, k2) -> k1 + k2 lambda mod r is injective on 64-bit pairs (the shortest vector of the GLV lattice has
// norm ~2^127), so 128 random bits still give 2^128 distinct multipliers.  jac_add_distinct applies: while a
// window is added the accumulator is (16a + 16b lambda) P with a, b < 2^60 prefixes, which equals +-d P or
// +-d lambda P (d <= 8) only if it is the identity, again by the shortest-vector bound.
BN_DEVN void g1_mul_glv(G1Jac& r, const G1Affine& p, const uint32_t* k1, const uint32_t* k2) {
  G1Jac tab[8], t;
  G1Jac& acc = r;                                               // accumulated in place, as in jac_mul_window
  jac_from_affine(tab[0], p);
  jac_dbl(tab[1], tab[0]);
  for (int j = 2; j < 8; ++j) jac_add_distinct(tab[j], tab[j - 1], tab[0]);
  const Fp beta = fp_load_const(C_GLV_BETA);
  signed char d1[17], d2[17];
  int c1 = 0, c2 = 0;
  for (int j = 0; j < 16; ++j) {
    int v = (int)((k1[j >> 3] >> (4 * (j & 7))) & 15u) + c1;
    c1 = v > 8; d1[j] = (signed char)(v - 16 * c1);
    v = (int)((k2[j >> 3] >> (4 * (j & 7))) & 15u) + c2;
    c2 = v > 8; d2[j] = (signed char)(v - 16 * c2);
  }
  d1[16] = (signed char)c1; d2[16] = (signed char)c2;
  jac_set_identity(acc);
  for (int j = 16; j >= 0; --j) {
    if (j != 16) { jac_dbl(acc, acc); jac_dbl(acc, acc); jac_dbl(acc, acc); jac_dbl(acc, acc); }
    jac_window_entry<false>(t, tab, (int)d1[j]);
    jac_add_distinct(acc, acc, t);
    jac_window_entry<false>(t, tab, (int)d2[j]);
    t.x = fp_mul(t.x, beta);                                   // phi: x -> beta x (the identity keeps z = 0)
    jac_add_distinct(acc, acc, t);
  }
}

// ---- k * P on G1 for a FULL scalar through the endomorphism (round 6): ECDSA::sign (/root/reference/src/ecdsa.rs:31, sk * H(m)) and the
// variable-base bn254_batch_g1_mul.  G1 has prime order r (cofactor 1), so k * P = (k mod r) * P for every point of the curve, and
// k mod r = k1 + k2 lambda with 0 <= k1 < 2^128, |k2| < 2^127 (constants and bounds: gen_constants.py): one joint ladder of 33 windows —
// 128 doublings and 2 x 33 additions over the table j * P (the second addend is phi(j * P) = (beta X, Y, Z), its sign k2's) — where
// jac_mul spends 256 doublings and 65 additions.  The scalar may be a private key: the decomposition is straight-line integer arithmetic
// with selects, the window entries are found by scans (jac_window_entry<true>), the additions are the complete jac_add.
// out = the low `no` words of a (na words) times b (nb words)
BN_DEV void bn_mul_words(uint32_t* out, int no, const uint32_t* a, int na, const uint32_t* b, int nb) {
  uint64_t carry = 0;
  for (int k = 0; k < no; ++k) {
    uint64_t lo = carry & 0xFFFFFFFFull, hi = carry >> 32;
    for (int i = 0; i < na; ++i) {
      const int j = k - i;
      if (j < 0 || j >= nb) continue;
      const uint64_t t = (uint64_t)a[i] * b[j];
      lo += t & 0xFFFFFFFFull;
      hi += t >> 32;
    }
    out[k] = (uint32_t)lo;
    carry = hi + (lo >> 32);
  }
}
// k in [0, r) -> k1 (four words), |k2| (four words), the sign of k2
BN_DEV void glv_decompose(const uint32_t* k, uint32_t* k1, uint32_t* k2, bool& k2_neg) {
  uint32_t t[13], c1[2], c2[4], u[5], v[5];
  bn_mul_words(t, 11, k, 8, C_GLV_G1, 3);
  c1[0] = t[8]; c1[1] = t[9];                                  // < 2^64 (k < 2^254, g1 < 2^66)
  bn_mul_words(t, 13, k, 8, C_GLV_G2, 5);
  for (int i = 0; i < 4; ++i) c2[i] = t[8 + i];                // < 2^128
  // k1 = k - c1 a1 - c2 a2 (mod 2^160; the value is in [0, 2^128))
  bn_mul_words(u, 5, c1, 2, C_GLV_A1, 2);
  bn_mul_words(v, 5, c2, 4, C_GLV_A2, 4);
  uint32_t w[5];
  uint64_t bw = 0;
  for (int i = 0; i < 5; ++i) {
    const uint64_t d = (uint64_t)k[i] - u[i] - bw;
    w[i] = (uint32_t)d; bw = (d >> 63) & 1;
  }
  bw = 0;
  for (int i = 0; i < 5; ++i) {
    const uint64_t d = (uint64_t)w[i] - v[i] - bw;
    w[i] = (uint32_t)d; bw = (d >> 63) & 1;
  }
  for (int i = 0; i < 4; ++i) k1[i] = w[i];
  // k2 = c1 b1n - c2 a1 (mod 2^160, two's complement; |k2| < 2^127)
  bn_mul_words(u, 5, c1, 2, C_GLV_B1N, 4);
  bn_mul_words(v, 5, c2, 4, C_GLV_A1, 2);
  bw = 0;
  for (int i = 0; i < 5; ++i) {
    const uint64_t d = (uint64_t)u[i] - v[i] - bw;
    w[i] = (uint32_t)d; bw = (d >> 63) & 1;
  }
  k2_neg = (w[4] >> 31) != 0;
  const uint32_t m = k2_neg ? 0xFFFFFFFFu : 0u;
  uint64_t cy = k2_neg ? 1 : 0;
  for (int i = 0; i < 4; ++i) {                               // |k2| = (w ^ m) + (m & 1)
    const uint64_t d = (uint64_t)(w[i] ^ m) + cy;
    k2[i] = (uint32_t)d; cy = d >> 32;
  }
}
BN_DEVN void g1_mul_glv_full(G1Jac& r, const G1Affine& p, const uint32_t* kin) {
  uint32_t k[8], k1[4], k2[4];
  for (int i = 0; i < 8; ++i) k[i] = kin[i];
  for (int it = 0; it < 6; ++it) {                           // 2^256 / r < 6; the subtraction is applied by selects
    const bool ge = u256_geq(k, C_ORDER_R);
    uint32_t bw = 0;
    for (int i = 0; i < 8; ++i) {
      const uint64_t d = (uint64_t)k[i] - C_ORDER_R[i] - bw;
      bw = (uint32_t)(d >> 63) & 1u;
      k[i] = ge ? (uint32_t)d : k[i];
    }
  }
  bool neg2;
  glv_decompose(k, k1, k2, neg2);
  G1Jac tab[8], t;
  G1Jac& acc = r;
  jac_from_affine(tab[0], p);
  jac_dbl(tab[1], tab[0]);
  for (int j = 2; j < 8; ++j) jac_add(tab[j], tab[j - 1], tab[0]);
  const Fp beta = fp_load_const(C_GLV_BETA);
  signed char d1[33], d2[33];
  int c1 = 0, c2 = 0;
  for (int j = 0; j < 32; ++j) {
    int v = (int)((k1[j >> 3] >> (4 * (j & 7))) & 15u) + c1;
    c1 = v > 8; d1[j] = (signed char)(v - 16 * c1);
    v = (int)((k2[j >> 3] >> (4 * (j & 7))) & 15u) + c2;
    c2 = v > 8; d2[j] = (signed char)(v - 16 * c2);
  }
  d1[32] = (signed char)c1; d2[32] = (signed char)c2;
  jac_set_identity(acc);
  for (int j = 32; j >= 0; --j) {
    if (j != 32) { jac_dbl(acc, acc); jac_dbl(acc, acc); jac_dbl(acc, acc); jac_dbl(acc, acc); }
    jac_window_entry<true>(t, tab, (int)d1[j]);
    jac_add(acc, acc, t);
    jac_window_entry<true>(t, tab, (int)d2[j]);
    t.y = fp_select(neg2, fp_neg(t.y), t.y);
    t.x = fp_mul(t.x, beta);                                   // phi: x -> beta x (the identity keeps z = 0)
    jac_add(acc, acc, t);
  }
}

// A + B for two AFFINE points with x_A != x_B (neither the identity), given dinv = 1 / (x_B - x_A): the chord formula, 2 products + 1
// square — what an addition costs once the inversion is shared (Montgomery's trick over a batch of denominators: the table builders
// k_pool_widen_*, bn254_hip.hip).  Outputs carried and weakly reduced: they are stored as table entries and meet jac_madd next.
template <class F> BN_DEV void aff_add_given_inv(Affine<F>& r, const Affine<F>& a, const Affine<F>& b, const F& dinv) {
  const F lam = f_mul(f_norm(f_sub(b.y, a.y)), dinv);
  const F x3 = f_reduce(f_sub(f_sub(f_sqr(lam), a.x), b.x));
  r.y = f_reduce(f_sub(f_mul(lam, f_norm(f_sub(a.x, x3))), a.y));
  r.x = x3;
  r.inf = false;
}

template <class F> BN_DEVN void jac_to_affine(Affine<F>& r, const Jac<F>& p) {
  bool inf = f_is_zero(p.z);
  F zi = f_norm(f_inv(p.z)), zi2 = f_sqr(zi);
  r.x = f_mul(p.x, zi2);
  r.y = f_mul(p.y, f_mul(zi2, zi));
  r.inf = inf;
  if (inf) { f_set_zero(r.x); f_set_zero(r.y); }
}

BN_DEV bool g1_on_curve(const G1Affine& p) {   // y^2 = x^3 + 3
  return p.inf || fp_eq(fp_sqr(p.y), fp_add(fp_mul(fp_sqr(p.x), p.x), fp_load_const(C_THREE)));
}
BN_DEV bool g2_on_curve(const G2Affine& p) {   // y^2 = x^3 + 3/xi
  return p.inf || fp2_eq(fp2_sqr(p.y), fp2_add(fp2_mul(fp2_sqr(p.x), p.x), fp2_load_const(C_TWIST_B)));
}
// psi = twist o Frobenius o untwist on Jacobian coordinates: (conj X * g_x, conj Y * g_y, conj Z)
BN_DEV void g2_psi(G2Jac& r, const G2Jac& p) {
  r.x = fp2_mul(fp2_conj(p.x), fp2_load_const(C_TW_FROB_X1));
  r.y = fp2_mul(fp2_conj(p.y), fp2_load_const(C_TW_FROB_Y1));
  r.z = fp2_conj(p.z);
}
// equality of two Jacobian points (cross-multiplied; identity only equals identity)
BN_DEV bool g2_jac_equal(const G2Jac& a, const G2Jac& b) {
  bool ai = fp2_is_zero(a.z), bi = fp2_is_zero(b.z);
  Fp2 za2 = fp2_sqr(a.z), zb2 = fp2_sqr(b.z);
  bool ex = fp2_eq(fp2_mul(a.x, zb2), fp2_mul(b.x, za2));
  bool ey = fp2_eq(fp2_mul(a.y, fp2_mul(zb2, b.z)), fp2_mul(b.y, fp2_mul(za2, a.z)));
  return (ai && bi) || (!ai && !bi && ex && ey);
}
// order-r subgroup membership of a twist point (what AffineG2::new enforces,
// /root/reference/src/utils.rs:113).  Instead of the 254-bit ladder [r]P == O this uses the BN
// endomorphism test  [u+1]P + psi([u]P) + psi^2([u]P) == psi^3([2u]P)  (Scott, "A note on group
// membership tests for G1, G2 and GT on BLS pairing-friendly curves", BN case): one 63-bit ladder
// (NAF of u, mixed additions).  tests/test_oracle_model.py checks the identity against [r]P == O on
// random twist points in and out of the subgroup; the CPU suite also runs this implementation on them.
// the tail shared by both forms below: given up = [u]P
BN_DEV bool g2_in_subgroup_tail(const G2Affine& p, const G2Jac& up) {
  G2Jac t, lhs, rhs;
  lhs = up;
  jac_accumulate(lhs, p);                          // [u+1]P
  g2_psi(t, up);
  jac_add(lhs, lhs, t);                            // + psi([u]P)
  g2_psi(t, t);
  jac_add(lhs, lhs, t);                            // + psi^2([u]P)
  jac_dbl(rhs, up);
  g2_psi(rhs, rhs); g2_psi(rhs, rhs); g2_psi(rhs, rhs);
  return p.inf || g2_jac_equal(lhs, rhs);
}
BN_DEVN bool g2_in_subgroup(const G2Affine& p) {
  G2Affine pn = p;
  pn.y = fp2_neg(p.y);
  G2Jac up;
  jac_from_affine(up, p);
  for (int i = 0; i < BN_U_NAF_LEN; ++i) {        // wave-uniform: u is a public constant
    BN_SET_STEP_PRIORITY(i);
    jac_dbl(up, up);
    int d = C_U_NAF[i];
    if (d > 0) jac_accumulate(up, p);              // common-case addition; complete formula if any lane needs it
    else if (d < 0) jac_accumulate(up, pn);        // (a crafted low-order twist point can reach P = +-Q)
  }
  return g2_in_subgroup_tail(p, up);
}
// The same with the ladder's accumulator in an LDS slot of the caller (the lane-pair decoders: k_decode_g2_pair, k_decompress_g2_pair):
// 63 doublings and 22 additions update it in place — no Jacobian point crosses the private segment inside the ladder; the point itself
// rides into the additions in registers (AffSrc).
struct G2AffSrc {
  Fp2 x, y;
  bool inf;
#if defined(__HIPCC__)
  __device__ __forceinline__
#endif
  void operator()(G2Affine& q) const { q.x = x; q.y = y; q.inf = inf; }
};
BN_DEVN bool g2_in_subgroup_lds(const G2Affine& p, G2Jac& up) {
  BN_ASSUME_LDS(&up);
  const G2AffSrc sp = {p.x, p.y, p.inf}, sn = {p.x, fp2_neg(p.y), p.inf};
  jac_from_affine(up, p);
  for (int i = 0; i < BN_U_NAF_LEN; ++i) {
    BN_SET_STEP_PRIORITY(i);
    jac_dbl_lds(up);
    int d = C_U_NAF[i];
    if (d > 0) jac_accumulate_from(up, sp);
    else if (d < 0) jac_accumulate_from(up, sn);
  }
  return g2_in_subgroup_tail(p, up);
}

}  // namespace bn254
