// Optimal-ate pairing on BN254: Miller loop (variable Q, and the table-driven constant
// Q = -G2::one() of every verify) and the final exponentiation.
//
// This is the device-side replacement for `bn::pairing_batch` at its two call sites
// /root/reference/src/ecdsa.rs:57 (ECDSA::verify) and :86 (check_public_keys): a product of
// Miller functions sharing one accumulator f (one squaring per loop step for all pairs), pairs
// with an identity member skipped, then f^((q^12-1)/r) and a comparison with one.
//
// Line shape (D-type twist, untwist (x',y') -> (x' w^2, y' w^3)):
//   l(P) = l0 + l1 w + l2 w^3,  l0 = c0 * yP, l1 = c1 * xP, l2 = c2   with (c0,c1,c2) in Fq2
// Lines are scaled by Fq2 factors (killed by the final exponentiation).
#pragma once
#include "bn254_curve.h"


namespace bn254 {

struct G2Proj { Fp2 x, y, z; };            // homogeneous projective twist point
struct LineCoef { Fp2 c0, c1, c2; };

// T <- 2T;  c0 = 2YZ, c1 = -3X^2, c2 = Y^2 - 3b'Z^2.
// Contract of the step functions: T tight on entry and on exit; the line coefficients come out lazy (up to 3 units)
// and are made tight by the line products below.  Carry sites (NS) mark every place a lazy value meets a product
// whose column budget (fp2_mul: A * B <= 6, fp2_sqr: A <= 1.8 units of 2^28, bn254_field.h) it might exceed; the
// search under the bound tracker (tests/norm_site_search.py -> bn254_norm_sites.h) keeps the ones that are needed.
// The values of T stay small by themselves (every coordinate is a short combination of fresh products).
BN_DEVN void dbl_step(G2Proj& t, LineCoef& l) {                       // sites 200 .. 209
  Fp2 xy = fp2_mul(t.x, t.y), b = fp2_sqr(t.y), c = fp2_sqr(t.z);
  Fp2 e = fp2_mul(c, fp2_load_const(C_TWIST_3B));
  Fp2 f = fp2_add(fp2_dbl(e), e);
  Fp2 h = fp2_sub(fp2_sub(fp2_sqr(NS(200, fp2_add(t.y, t.z))), b), c);
  Fp2 x2 = fp2_sqr(t.x);
  Fp2 e2x4 = NS(201, fp2_dbl(fp2_dbl(fp2_sqr(e))));                 // 12 e^2 = 3 * carry(4 e^2): 12-fold limbs would leave int32
  Fp2 e2x12 = fp2_add(fp2_dbl(e2x4), e2x4);
  G2Proj o;
  o.x = NS(203, fp2_mul(fp2_dbl(xy), NS(202, fp2_sub(b, f))));
  o.y = NS(205, fp2_sub(fp2_sqr(NS(204, fp2_add(b, f))), e2x12));
  o.z = NS(206, fp2_dbl(fp2_dbl(fp2_mul(b, h))));
  l.c0 = h;
  l.c1 = fp2_neg(fp2_add(fp2_dbl(x2), x2));
  l.c2 = fp2_sub(b, e);
  t = o;
}
// T <- T + Q (Q affine, tight);  c0 = mu, c1 = -theta, c2 = theta*x2 - mu*y2
BN_DEVN void add_step(G2Proj& t, LineCoef& l, const Fp2& qx, const Fp2& qy) {   // sites 210 .. 219
  Fp2 theta = NS(210, fp2_sub(t.y, fp2_mul(qy, t.z)));
  Fp2 mu = NS(211, fp2_sub(t.x, fp2_mul(qx, t.z)));
  Fp2 c = fp2_sqr(theta), d = fp2_sqr(mu), e = fp2_mul(mu, d);
  Fp2 f = fp2_mul(t.z, c), g = fp2_mul(t.x, d);
  Fp2 h = NS(212, fp2_sub(fp2_sub(fp2_add(e, f), g), g));
  G2Proj o;
  o.x = fp2_mul(mu, h);
  o.y = NS(214, fp2_sub(fp2_mul(theta, NS(213, fp2_sub(g, h))), fp2_mul(e, t.y)));
  o.z = fp2_mul(t.z, e);
  l.c0 = mu;
  l.c1 = fp2_neg(theta);
  l.c2 = fp2_sub(fp2_mul(theta, qx), fp2_mul(mu, qy));
  t = o;
}
// f <- f * line(P); a skipped pair multiplies by one
BN_DEV void mul_by_line(Fp12& f, const LineCoef& l, const Fp& px, const Fp& py, bool skip) {   // sites 220 .. 222
  Fp2 l0 = fp2_mul_fp(l.c0, py), l1 = fp2_mul_fp(l.c1, px), l2 = NS(220, l.c2);
  l0 = fp2_select(skip, fp2_one(), l0);
  l1 = fp2_select(skip, fp2_zero(), l1);
  l2 = fp2_select(skip, fp2_zero(), l2);
  fp12_mul_line(f, f, l0, l1, l2);
}
// f <- f * lineA(pa) * lineB(pb), lineB = entry idx of the constant table (whose c2 is one).
//   (l0 + l1 w + l2 w^3)(m0 + m1 w + w^3) =
//     (l0 m0 + xi l2) + (l0 m1 + l1 m0) w + l1 m1 w^2 + (l0 + l2 m0) w^3 + (l1 + l2 m1) w^4
// 5 Fq2 products for the line product and 17 for f * (5-term element), against 2 x 13.
// `any_skip` (wave-uniform): some lane of the wave has a skipped pair; without one the selects are branched over.
BN_DEV void mul_by_two_lines(Fp12& f, const LineCoef& l, const Fp& pax, const Fp& pay, bool skip_a, int idx, const Fp& pbx,
                             const Fp& pby, bool skip_b, bool any_skip) {   // sites 223 .. 229
  Fp2 l0 = fp2_mul_fp(l.c0, pay), l1 = fp2_mul_fp(l.c1, pax), l2 = NS(223, l.c2);
  if (any_skip) {
    l0 = fp2_select(skip_a, fp2_one(), l0);
    l1 = fp2_select(skip_a, fp2_zero(), l1);
    l2 = fp2_select(skip_a, fp2_zero(), l2);
  }
  Fp2 m0 = fp2_mul_fp(fp2_load_const(C_NEG_G2_LINES[idx][0]), pby), m1 = fp2_mul_fp(fp2_load_const(C_NEG_G2_LINES[idx][1]), pbx);
  Fp2 v0 = fp2_mul(l0, m0), v1 = fp2_mul(l1, m1);
  Fp2 x01 = fp2_sub(fp2_sub(fp2_mul(fp2_add(l0, l1), fp2_add(m0, m1)), v0), v1);
  Fp2 w0 = fp2_add(v0, fp2_mul_xi(l2));
  Fp2 w3 = fp2_add(l0, fp2_mul(l2, m0));
  Fp2 w4 = fp2_add(l1, fp2_mul(l2, m1));
  Fp6 b0;
  b0.c0 = NS(224, w0); b0.c1 = v1; b0.c2 = NS(225, w4);
  Fp2 b10 = NS(226, x01), b11 = NS(227, w3);
  if (any_skip) {
    b0.c0 = fp2_select(skip_b, l0, b0.c0);
    b0.c1 = fp2_select(skip_b, fp2_zero(), b0.c1);
    b0.c2 = fp2_select(skip_b, fp2_zero(), b0.c2);
    b10 = fp2_select(skip_b, l1, b10);
    b11 = fp2_select(skip_b, l2, b11);
  }
  fp12_mul_line2(f, f, b0, b10, b11);
}
BN_DEV void fixed_line(LineCoef& l, int idx) {
  l.c0 = fp2_load_const(C_NEG_G2_LINES[idx][0]);
  l.c1 = fp2_load_const(C_NEG_G2_LINES[idx][1]);
  l.c2 = fp2_load_const(C_NEG_G2_LINES[idx][2]);
}

// Miller loop over up to two pairs sharing f:
//   pair A: (pa, qa) with a variable twist point qa   (enabled by has_a; skipped if skip_a)
//   pair B: (pb, -G2::one()) through the constant line table (enabled by HAS_B; skipped if skip_b)
// "skipped" = the pair has an identity member and contributes 1; the
// lane still walks the loop with dummy coordinates so the wave stays convergent.
// F_LDS: the caller's f is a __shared__ object (every kernel of bn254_pair.hip) -> LDS instructions for it
template <bool HAS_A, bool HAS_B, bool F_LDS = false>
BN_DEVM void miller_loop(Fp12& f, const G1Affine& pa, const G2Affine& qa, const G1Affine& pb) {
  if constexpr (F_LDS) BN_ASSUME_LDS(&f);
  fp12_set_one(f);
  G2Proj t;
  LineCoef l;
  Fp2 qa_yneg;
  bool skip_a = !HAS_A || pa.inf || qa.inf;
  bool skip_b = !HAS_B || pb.inf;
#if defined(__HIPCC__)
  const bool any_skip = __builtin_amdgcn_ballot_w64(skip_a || skip_b) != 0;   // wave-uniform
#else
  const bool any_skip = skip_a || skip_b;
#endif
  if constexpr (HAS_A) { t.x = qa.x; t.y = qa.y; t.z = fp2_one(); qa_yneg = fp2_neg(qa.y); }
  int idx = 0;
  constexpr bool BOTH = HAS_A && HAS_B;      // both pairs: one merged multiplication per step
  for (int d = 0; d < 64; ++d) {
    BN_SET_STEP_PRIORITY(d);
    fp12_sqr(f, f);
    if constexpr (HAS_A) { dbl_step(t, l); if constexpr (!BOTH) mul_by_line(f, l, pa.x, pa.y, skip_a); }
    if constexpr (BOTH) mul_by_two_lines(f, l, pa.x, pa.y, skip_a, idx++, pb.x, pb.y, skip_b, any_skip);
    else if constexpr (HAS_B) { fixed_line(l, idx++); mul_by_line(f, l, pb.x, pb.y, skip_b); }
    int digit = C_ATE_NAF[d];
    if (digit != 0) {   // wave-uniform
      if constexpr (HAS_A) {
        Fp2 qy = fp2_select(digit > 0, qa.y, qa_yneg);
        add_step(t, l, qa.x, qy);
        if constexpr (!BOTH) mul_by_line(f, l, pa.x, pa.y, skip_a);
      }
      if constexpr (BOTH) mul_by_two_lines(f, l, pa.x, pa.y, skip_a, idx++, pb.x, pb.y, skip_b, any_skip);
      else if constexpr (HAS_B) { fixed_line(l, idx++); mul_by_line(f, l, pb.x, pb.y, skip_b); }
    }
  }
  // + pi(Q), - pi^2(Q)
  if constexpr (HAS_A) {
    Fp2 q1x = fp2_mul(fp2_conj(qa.x), fp2_load_const(C_TW_FROB_X1));
    Fp2 q1y = fp2_mul(fp2_conj(qa.y), fp2_load_const(C_TW_FROB_Y1));
    add_step(t, l, q1x, q1y);
    if constexpr (!BOTH) mul_by_line(f, l, pa.x, pa.y, skip_a);
  }
  if constexpr (BOTH) mul_by_two_lines(f, l, pa.x, pa.y, skip_a, idx++, pb.x, pb.y, skip_b, any_skip);
  else if constexpr (HAS_B) { fixed_line(l, idx++); mul_by_line(f, l, pb.x, pb.y, skip_b); }
  if constexpr (HAS_A) {
    Fp2 q2x = fp2_mul(qa.x, fp2_load_const(C_TW_FROB_X2));
    add_step(t, l, q2x, qa.y);
    if constexpr (!BOTH) mul_by_line(f, l, pa.x, pa.y, skip_a);
  }
  if constexpr (BOTH) mul_by_two_lines(f, l, pa.x, pa.y, skip_a, idx++, pb.x, pb.y, skip_b, any_skip);
  else if constexpr (HAS_B) { fixed_line(l, idx++); mul_by_line(f, l, pb.x, pb.y, skip_b); }
}

// ---- keyed verify: the line functions of a REGISTERED public key ----------------------------------------------------------
// The twist-point arithmetic of pair A = (H(m), pk) — 64 doublings and 23 additions, ~2.5 k of the 11.1 k Fq products of a
// verify's Miller loop — depends on pk alone.  For a key that verifies many messages (a validator set; the reference's
// PublicKey is Copy and validated once at construction, /root/reference/src/types.rs:80-99) the 87 lines are computed
// once (g2_line_table) in the shape of the constant table of -G2::one(): scaled so that c2 = 1, (c0, c1) stored.  A verify
// then multiplies two table lines per step:
//   (l0 + l1 w + w^3)(m0 + m1 w + w^3) = (l0 m0 + xi) + (l0 m1 + l1 m0) w + l1 m1 w^2 + (l0 + m0) w^3 + (l1 + m1) w^4
// 3 Fq2 products (Karatsuba for the w term) + the four scalings by the G1 coordinates, against 10 for the step of T, 4
// scalings and 5 for the line product: 36 product slots per doubling step instead of 48.
// c2 = Y^2 - 3b'Z^2 (tangent) or theta x_Q - mu y_Q (chord) vanishes only on a handful of algebraic points, none of
// which lies in the order-r subgroup except with probability ~2^-250 (keys are subgroup-checked at registration, so an
// adversarial key cannot aim for one); g2_line_table reports it and the key is refused.
struct KeyLine { Fp2 c0, c1; };
template <class Store>
BN_DEV bool g2_line_table(const G2Affine& q, Store&& store) {            // sites 290 .. 293
  G2Proj t;
  LineCoef l;
  t.x = q.x; t.y = q.y; t.z = fp2_one();
  const Fp2 q_yneg = fp2_neg(q.y);
  bool ok = true;
  int idx = 0;
  auto emit = [&]() {
    const Fp2 c2 = NR(290, l.c2);
    ok = !fp2_is_zero(c2) && ok;
    const Fp2 inv = NS(291, fp2_inv(c2));
    KeyLine kl;
    kl.c0 = fp2_mul(NS(292, l.c0), inv);
    kl.c1 = fp2_mul(NS(293, l.c1), inv);
    store(idx++, kl);
  };
  for (int d = 0; d < 64; ++d) {
    dbl_step(t, l); emit();
    const int digit = C_ATE_NAF[d];
    if (digit != 0) { add_step(t, l, q.x, fp2_select(digit > 0, q.y, q_yneg)); emit(); }
  }
  add_step(t, l, fp2_mul(fp2_conj(q.x), fp2_load_const(C_TW_FROB_X1)), fp2_mul(fp2_conj(q.y), fp2_load_const(C_TW_FROB_Y1))); emit();
  add_step(t, l, fp2_mul(q.x, fp2_load_const(C_TW_FROB_X2)), q.y); emit();
  return ok;
}
// f <- f * lineA(pa) * lineB(pb): lineA = (la.c0, la.c1, 1) of the registered key, lineB = entry idx of the -G2 table
BN_DEV void mul_by_two_table_lines(Fp12& f, const KeyLine& la, const Fp& pax, const Fp& pay, bool skip_a, int idx, const Fp& pbx, const Fp& pby,
                                   bool skip_b, bool any_skip) {   // sites 280 .. 284
  const Fp2 l0 = fp2_mul_fp(la.c0, pay), l1 = fp2_mul_fp(la.c1, pax);
  const Fp2 m0 = fp2_mul_fp(fp2_load_const(C_NEG_G2_LINES[idx][0]), pby), m1 = fp2_mul_fp(fp2_load_const(C_NEG_G2_LINES[idx][1]), pbx);
  const Fp2 v0 = fp2_mul(l0, m0), v1 = fp2_mul(l1, m1);
  const Fp2 x01 = fp2_sub(fp2_sub(fp2_mul(NS(280, fp2_add(l0, l1)), NS(281, fp2_add(m0, m1))), v0), v1);
  Fp6 b0;
  b0.c0 = NS(282, fp2_add(v0, fp2_load_const(C_XI_MONT))); b0.c1 = v1; b0.c2 = NS(283, fp2_add(l1, m1));
  Fp2 b10 = NS(284, x01), b11 = NS(285, fp2_add(l0, m0));
  if (any_skip) {
    // a skipped pair contributes the line 1: both skipped -> 1; A skipped -> lineB = m0 + m1 w + w^3; B skipped -> lineA
    const Fp2 one = fp2_one(), zero = fp2_zero();
    const bool both = skip_a && skip_b, only_a = skip_a && !skip_b, only_b = skip_b && !skip_a;
    b0.c0 = fp2_select(both, one, fp2_select(only_a, m0, fp2_select(only_b, l0, b0.c0)));
    b0.c1 = fp2_select(skip_a || skip_b, zero, b0.c1);
    b0.c2 = fp2_select(skip_a || skip_b, zero, b0.c2);
    b10 = fp2_select(both, zero, fp2_select(only_a, m1, fp2_select(only_b, l1, b10)));
    b11 = fp2_select(both, zero, fp2_select(skip_a || skip_b, one, b11));
  }
  fp12_mul_line2(f, f, b0, b10, b11);
}
// f = miller(pa, key) * miller(pb, -G2): `tab` = the key's 87 x (c0, c1) as [line][coefficient][re / im][limb]
template <bool F_LDS = false>
BN_DEVM void miller_loop_keyed(Fp12& f, const G1Affine& pa, bool key_inf, const int32_t (*tab)[2][2][BN_LIMBS], const G1Affine& pb) {
  if constexpr (F_LDS) BN_ASSUME_LDS(&f);
  fp12_set_one(f);
  const bool skip_a = pa.inf || key_inf, skip_b = pb.inf;
#if defined(__HIPCC__)
  const bool any_skip = __builtin_amdgcn_ballot_w64(skip_a || skip_b) != 0;   // wave-uniform
#else
  const bool any_skip = skip_a || skip_b;
#endif
  int idx = 0;
  auto step = [&]() {
    KeyLine la;
    la.c0 = fp2_load_const(tab[idx][0]); la.c1 = fp2_load_const(tab[idx][1]);
    mul_by_two_table_lines(f, la, pa.x, pa.y, skip_a, idx, pb.x, pb.y, skip_b, any_skip);
    ++idx;
  };
  for (int d = 0; d < 64; ++d) {
    BN_SET_STEP_PRIORITY(d);
    fp12_sqr(f, f);
    step();
    if (C_ATE_NAF[d] != 0) step();      // wave-uniform
  }
  step();
  step();
}

#if defined(BN_TRIO_FORMULAS)
// ---- ECDSA::verify's Miller loop in ROUNDS (octet layout, bn254_trio.hip) ------------------------------------------
// f = miller(pa, qa) * miller(pb, -G2), the same values as miller_loop<true, true>, with EVERY Fq2 product of a loop step
// scheduled into rounds of four independent products (trio4: one per lane pair of the octet, results exchanged):
//   doubling step  48 products = 12 rounds: the line of 2T and its product with the table line (16), f^2 by the complex
//                  method as two Karatsuba Fq6 products (12), (f^2) * (line product) as Karatsuba over Fq6 with the
//                  sparse half (17), and the three products that only the NEXT step's T needs in the free slots (3);
//   addition step  39 products = 10 rounds: lines first (13), then the rest of T + Q (9) beside f * (line product) (17).
// The serial pair layout spends 48 / 41 product-times on the same steps.  The formulas of the twist point and the lines
// are those of dbl_step / add_step / mul_by_two_lines with their carry sites (200 .. 227); the Fq12 parts have their own
// (400 ..).  The scalings by the coordinates of pa / pb ride along as Fq2 products with (k + 0 i).
#define BN_KOP(X, Y, K) fp6_kop<K>(X), fp6_kop<K>(Y)
struct TrioLineProduct { Fp6 b0; Fp2 b10, b11; };      // (l0 + l1 w + l2 w^3)(m0 + m1 w + w^3) = b0 + (b10 + b11 v) w
// from the round results v0 = l0 m0, v1 = l1 m1, w3p = l2 m0, w4p = l2 m1: everything but b10 ...
BN_DEV void trio_line_product(TrioLineProduct& L, const Fp2& l0, const Fp2& l1, const Fp2& l2, const Fp2& v0, const Fp2& v1, const Fp2& w3p,
                              const Fp2& w4p, bool skip_b, bool any_skip) {
  L.b0.c0 = NS(224, fp2_add(v0, fp2_mul_xi(l2))); L.b0.c1 = v1; L.b0.c2 = NS(225, fp2_add(l1, w4p));
  L.b11 = NS(227, fp2_add(l0, w3p));
  if (any_skip) {
    L.b0.c0 = fp2_select(skip_b, l0, L.b0.c0);
    L.b0.c1 = fp2_select(skip_b, fp2_zero(), L.b0.c1);
    L.b0.c2 = fp2_select(skip_b, fp2_zero(), L.b0.c2);
    L.b11 = fp2_select(skip_b, l2, L.b11);
  }
}
// ... and b10 = l0 m1 + l1 m0 from x01p = (l0 + l1)(m0 + m1)
BN_DEV void trio_line_product_b10(TrioLineProduct& L, const Fp2& l1, const Fp2& v0, const Fp2& v1, const Fp2& x01p, bool skip_b, bool any_skip) {
  L.b10 = NS(226, fp2_sub(fp2_sub(x01p, v0), v1));
  if (any_skip) L.b10 = fp2_select(skip_b, l1, L.b10);
}
// g * (b0 + b1 w) from its 17 products: t0 = g0 b0 (Karatsuba, 6), u = (g0 + g1)(b0 + b1) (6), t1 = g1 (b10 + b11 v) (5:
// g10 b10, g11 b11, g12 b11, (g10 + g11)(b10 + b11), g12 b10).  Sites 426 .. 445.
struct TrioLineMul { Fp6 sg, bs; Fp2 sb; };
BN_DEV void trio_line_mul_prepare(TrioLineMul& M, const Fp12& g, const TrioLineProduct& L) {
  fp6_add(M.sg, g.c0, g.c1); fp6_site_n<420>(M.sg, M.sg);
  M.bs.c0 = NS(423, fp2_add(L.b0.c0, L.b10)); M.bs.c1 = NS(424, fp2_add(L.b0.c1, L.b11)); M.bs.c2 = L.b0.c2;
  M.sb = fp2_add(L.b10, L.b11);
}
BN_DEV void trio_line_mul_finish(Fp12& r, const Fp2 (&pt0)[6], const Fp2 (&pu)[6], const Fp2 (&pt1)[5]) {
  Fp6 t0, t1, u, s;
  fp6_kfin<426>(t0, pt0);
  fp6_kfin<430>(u, pu);
  t1.c0 = NS(434, fp2_add(fp2_mul_xi(pt1[2]), pt1[0]));
  t1.c1 = NS(435, fp2_sub(fp2_sub(pt1[3], pt1[0]), pt1[1]));
  t1.c2 = NS(436, fp2_add(pt1[4], pt1[1]));
  fp6_sub(u, u, t0);
  fp6_sub(u, u, t1);
  fp6_mul_v(s, t1);
  fp6_add(s, t0, s);
  fp6_site_r<437>(r.c0, s);
  fp6_site_r<440>(r.c1, u);
}
// one doubling step: f <- f^2 * line_{2T}(pa) * tableline_idx(pb), T <- 2T
BN_DEV void trio_dbl_iter(Fp12& f, G2Proj& t, int idx, const Fp2& PAX, const Fp2& PAY, const Fp2& PBX, const Fp2& PBY, bool skip_a,
                          bool skip_b, bool any_skip) {
  const Fp2 C0 = fp2_load_const(C_NEG_G2_LINES[idx][0]), C1 = fp2_load_const(C_NEG_G2_LINES[idx][1]);
  // rounds 1-4: the two lines and their product (dbl_step's formulas; what only T needs is left for round 12)
  Fp2 b, c, m0, x2, e, hh, m1, l1, l0, w3p, w4p, xy, v0, v1, x01p, e2;
  trio4(b, c, m0, x2, t.y, t.y, t.z, t.z, C0, PBY, t.x, t.x);
  const Fp2 yz = NS(200, fp2_add(t.y, t.z));
  trio4(e, hh, m1, l1, c, fp2_load_const(C_TWIST_3B), yz, yz, C1, PBX, fp2_neg(fp2_add(fp2_dbl(x2), x2)), PAX);
  const Fp2 h = fp2_sub(fp2_sub(hh, b), c);
  Fp2 l2 = NS(223, fp2_sub(b, e));
  if (any_skip) l2 = fp2_select(skip_a, fp2_zero(), l2);
  trio4(l0, w3p, w4p, xy, h, PAY, l2, m0, l2, m1, t.x, t.y);
  if (any_skip) {
    l0 = fp2_select(skip_a, fp2_one(), l0);
    l1 = fp2_select(skip_a, fp2_zero(), l1);
  }
  trio4(v0, v1, x01p, e2, l0, m0, l1, m1, fp2_add(l0, l1), fp2_add(m0, m1), e, e);
  TrioLineProduct L;
  trio_line_product(L, l0, l1, l2, v0, v1, w3p, w4p, skip_b, any_skip);
  trio_line_product_b10(L, l1, v0, v1, x01p, skip_b, any_skip);
  // rounds 5-7: g = f^2, complex method: ab = f0 f1, u = (f0 + f1)(f0 + v f1); g0 = u - ab - v ab, g1 = 2 ab.  Sites 400 .. 419
  Fp12 g;
  {
    Fp6 s, w, ab, u;
    fp6_add(s, f.c0, f.c1); fp6_site_n<400>(s, s);
    fp6_mul_v(w, f.c1); fp6_add(w, w, f.c0); fp6_site_n<403>(w, w);
    Fp2 pa[6], pu[6];
    trio4(pa[0], pa[1], pa[2], pa[3], BN_KOP(f.c0, f.c1, 0), BN_KOP(f.c0, f.c1, 1), BN_KOP(f.c0, f.c1, 2), BN_KOP(f.c0, f.c1, 3));
    trio4(pa[4], pa[5], pu[0], pu[1], BN_KOP(f.c0, f.c1, 4), BN_KOP(f.c0, f.c1, 5), BN_KOP(s, w, 0), BN_KOP(s, w, 1));
    trio4(pu[2], pu[3], pu[4], pu[5], BN_KOP(s, w, 2), BN_KOP(s, w, 3), BN_KOP(s, w, 4), BN_KOP(s, w, 5));
    fp6_kfin<406>(ab, pa);
    fp6_kfin<410>(u, pu);
    fp6_sub(u, u, ab);
    fp6_mul_v(s, ab);
    fp6_sub(u, u, s);
    fp6_site_r<414>(g.c0, u);
    fp6_add(s, ab, ab);
    fp6_site_r<417>(g.c1, s);
  }
  // rounds 8-12: f = g * L, and T <- 2T in the three free slots
  TrioLineMul M;
  trio_line_mul_prepare(M, g, L);
  Fp2 pt0[6], pu[6], pt1[5], ox, oz, oy2;
  trio4(pt0[0], pt0[1], pt0[2], pt0[3], BN_KOP(g.c0, L.b0, 0), BN_KOP(g.c0, L.b0, 1), BN_KOP(g.c0, L.b0, 2), BN_KOP(g.c0, L.b0, 3));
  trio4(pt0[4], pt0[5], pu[0], pu[1], BN_KOP(g.c0, L.b0, 4), BN_KOP(g.c0, L.b0, 5), BN_KOP(M.sg, M.bs, 0), BN_KOP(M.sg, M.bs, 1));
  trio4(pu[2], pu[3], pu[4], pu[5], BN_KOP(M.sg, M.bs, 2), BN_KOP(M.sg, M.bs, 3), BN_KOP(M.sg, M.bs, 4), BN_KOP(M.sg, M.bs, 5));
  trio4(pt1[0], pt1[1], pt1[2], pt1[3], g.c1.c0, L.b10, g.c1.c1, L.b11, g.c1.c2, L.b11, fp2_add(g.c1.c0, g.c1.c1), M.sb);
  const Fp2 f3 = fp2_add(fp2_dbl(e), e);
  const Fp2 bf = NS(204, fp2_add(b, f3));
  trio4(pt1[4], ox, oz, oy2, g.c1.c2, L.b10, fp2_dbl(xy), NS(202, fp2_sub(b, f3)), b, h, bf, bf);
  trio_line_mul_finish(f, pt0, pu, pt1);
  const Fp2 e2x4 = NS(201, fp2_dbl(fp2_dbl(e2)));
  t.y = NS(205, fp2_sub(oy2, fp2_add(fp2_dbl(e2x4), e2x4)));
  t.x = NS(203, ox);
  t.z = NS(206, fp2_dbl(fp2_dbl(oz)));
}
// one addition step: f <- f * line_{T+Q}(pa) * tableline_idx(pb), T <- T + (qx, qy)
BN_DEV void trio_add_iter(Fp12& f, G2Proj& t, const Fp2& qx, const Fp2& qy, int idx, const Fp2& PAX, const Fp2& PAY, const Fp2& PBX,
                          const Fp2& PBY, bool skip_a, bool skip_b, bool any_skip) {
  const Fp2 C0 = fp2_load_const(C_NEG_G2_LINES[idx][0]), C1 = fp2_load_const(C_NEG_G2_LINES[idx][1]);
  Fp2 t1, t2, m0, m1, ca, cb, l1, l0, w3p, w4p, v0, v1, x01p, c, d, e, ff, g, ox, oy1, oy2, oz;
  Fp2 pt0[6], pu[6], pt1[5], unused;
  trio4(t1, t2, m0, m1, qy, t.z, qx, t.z, C0, PBY, C1, PBX);
  const Fp2 theta = NS(210, fp2_sub(t.y, t1)), mu = NS(211, fp2_sub(t.x, t2));
  trio4(ca, cb, l1, l0, theta, qx, mu, qy, fp2_neg(theta), PAX, mu, PAY);
  Fp2 l2 = NS(223, fp2_sub(ca, cb));
  if (any_skip) {
    l2 = fp2_select(skip_a, fp2_zero(), l2);
    l0 = fp2_select(skip_a, fp2_one(), l0);
    l1 = fp2_select(skip_a, fp2_zero(), l1);
  }
  trio4(w3p, w4p, v0, v1, l2, m0, l2, m1, l0, m0, l1, m1);
  // b0 of the line product needs v0, v1, w4p only: its product with f0 starts beside the last line product
  TrioLineProduct L;
  trio_line_product(L, l0, l1, l2, v0, v1, w3p, w4p, skip_b, any_skip);
  trio4(x01p, c, d, pt0[0], fp2_add(l0, l1), fp2_add(m0, m1), theta, theta, mu, mu, BN_KOP(f.c0, L.b0, 0));
  trio_line_product_b10(L, l1, v0, v1, x01p, skip_b, any_skip);
  trio4(e, ff, g, pt0[1], mu, d, t.z, c, t.x, d, BN_KOP(f.c0, L.b0, 1));
  const Fp2 h = NS(212, fp2_sub(fp2_sub(fp2_add(e, ff), g), g));
  trio4(ox, oy1, oy2, oz, mu, h, theta, NS(213, fp2_sub(g, h)), e, t.y, t.z, e);
  TrioLineMul M;
  trio_line_mul_prepare(M, f, L);
  trio4(pt0[2], pt0[3], pt0[4], pt0[5], BN_KOP(f.c0, L.b0, 2), BN_KOP(f.c0, L.b0, 3), BN_KOP(f.c0, L.b0, 4), BN_KOP(f.c0, L.b0, 5));
  trio4(pu[0], pu[1], pu[2], pu[3], BN_KOP(M.sg, M.bs, 0), BN_KOP(M.sg, M.bs, 1), BN_KOP(M.sg, M.bs, 2), BN_KOP(M.sg, M.bs, 3));
  trio4(pu[4], pu[5], pt1[0], pt1[1], BN_KOP(M.sg, M.bs, 4), BN_KOP(M.sg, M.bs, 5), f.c1.c0, L.b10, f.c1.c1, L.b11);
  trio4(pt1[2], pt1[3], pt1[4], unused, f.c1.c2, L.b11, fp2_add(f.c1.c0, f.c1.c1), M.sb, f.c1.c2, L.b10, f.c1.c2, L.b10);
  trio_line_mul_finish(f, pt0, pu, pt1);
  t.x = ox;
  t.y = NS(214, fp2_sub(oy1, oy2));
  t.z = oz;
}
template <bool F_LDS = false>
BN_DEVM void miller_verify_rounds(Fp12& f, const G1Affine& pa, const G2Affine& qa, const G1Affine& pb) {
  if constexpr (F_LDS) BN_ASSUME_LDS(&f);
  fp12_set_one(f);
  G2Proj t;
  const bool skip_a = pa.inf || qa.inf, skip_b = pb.inf;
#if defined(__HIPCC__)
  const bool any_skip = __builtin_amdgcn_ballot_w64(skip_a || skip_b) != 0;   // wave-uniform
#else
  const bool any_skip = skip_a || skip_b;
#endif
  t.x = qa.x; t.y = qa.y; t.z = fp2_one();
  const Fp2 qa_yneg = fp2_neg(qa.y);
  const Fp2 PAX = fp2_from_fp(pa.x), PAY = fp2_from_fp(pa.y), PBX = fp2_from_fp(pb.x), PBY = fp2_from_fp(pb.y);
  int idx = 0;
  for (int d = 0; d < 64; ++d) {
    trio_dbl_iter(f, t, idx++, PAX, PAY, PBX, PBY, skip_a, skip_b, any_skip);
    const int digit = C_ATE_NAF[d];
    if (digit != 0)    // wave-uniform
      trio_add_iter(f, t, qa.x, fp2_select(digit > 0, qa.y, qa_yneg), idx++, PAX, PAY, PBX, PBY, skip_a, skip_b, any_skip);
  }
  // + pi(Q), - pi^2(Q)
  Fp2 q1x, q1y, q2x, unused;
  trio4(q1x, q1y, q2x, unused, fp2_conj(qa.x), fp2_load_const(C_TW_FROB_X1), fp2_conj(qa.y), fp2_load_const(C_TW_FROB_Y1), qa.x,
        fp2_load_const(C_TW_FROB_X2), qa.x, qa.x);
  trio_add_iter(f, t, q1x, q1y, idx++, PAX, PAY, PBX, PBY, skip_a, skip_b, any_skip);
  trio_add_iter(f, t, q2x, qa.y, idx++, PAX, PAY, PBX, PBY, skip_a, skip_b, any_skip);
}

// ---- the same loop as WAVE ROLES ("quad-wave" layout, bn254_trio.hip: k_miller_verify_quad) ------------------------------
// In the octet layout the four lane pairs of a verify share a wave, so every pair executes every linear instruction
// (Karatsuba sums, recombinations, carries): half of the loop's instructions.  Here the four pairs of a verify sit in the
// FOUR WAVES of a workgroup (one per SIMD of a CU, 32 verifies per workgroup) and run different instruction streams,
// exchanging Fq2 values through LDS mailboxes between workgroup barriers:
//   wave 3  the twist point: the line of the NEXT step (8 products for a doubling, 6 for an addition) and T's update
//           (4 / 9), one step ahead of the others — it depends on nothing they compute;
//   wave 2  the product of that line with the table line (7 products), then t1 = g1 * (b10 + b11 v) (5);
//   wave 0  ab = f0 f1 (6), then t0 = g0 b0 (6);
//   wave 1  u = (f0 + f1)(f0 + v f1) (6), then uu = (g0 + g1)(b0 + b1) (6);
//   the recombinations g0 = u - ab - v ab, g1 = 2ab and f0 <- t0 + v t1, f1 <- uu - t0 - t1 one coefficient per wave.
// Every wave does ~12 products per doubling step and only ITS share of the linear work.  The functions below are the
// roles' arithmetic — the same formulas and carry sites as the rounds above — as pure functions, so that the host
// emulation can run them one after the other (miller_verify_quad_model) and prove values and bounds.
struct QuadDblTmp { Fp2 b, e, h; };
// wave 3, doubling: the line (l0, l1, l2) of 2T evaluated at pa
BN_DEV void quad_dbl_line(Fp2& l0, Fp2& l1, Fp2& l2, QuadDblTmp& k, const G2Proj& t, const Fp2& PAX, const Fp2& PAY) {
  // a wave of its own: squarings as squarings (258 against 346 instructions) — except y^2, whose looser value bound would
  // reach the product of the lines through l2 = b - e (site 223 carries nothing)
  const Fp2 b = fp2_mul(t.y, t.y), c = fp2_sqr(t.z), x2 = fp2_sqr(t.x);
  const Fp2 yz = NS(200, fp2_add(t.y, t.z));
  const Fp2 e = fp2_mul(c, fp2_load_const(C_TWIST_3B)), hh = fp2_sqr(yz);
  l1 = fp2_mul(fp2_neg(fp2_add(fp2_dbl(x2), x2)), PAX);
  const Fp2 h = fp2_sub(fp2_sub(hh, b), c);
  l2 = NS(223, fp2_sub(b, e));
  l0 = fp2_mul(h, PAY);
  k.b = b; k.e = e; k.h = h;
}
BN_DEV void quad_dbl_update(G2Proj& t, const QuadDblTmp& k) {
  const Fp2 e2 = fp2_sqr(k.e), xy = fp2_mul(t.x, t.y);
  const Fp2 f3 = fp2_add(fp2_dbl(k.e), k.e);
  const Fp2 bf = NS(204, fp2_add(k.b, f3));
  const Fp2 ox = fp2_mul(fp2_dbl(xy), NS(202, fp2_sub(k.b, f3))), oz = fp2_mul(k.b, k.h), oy2 = fp2_mul(bf, bf);
  const Fp2 e2x4 = NS(201, fp2_dbl(fp2_dbl(e2)));
  t.y = NS(205, fp2_sub(oy2, fp2_add(fp2_dbl(e2x4), e2x4)));
  t.x = NS(203, ox);
  t.z = NS(206, fp2_dbl(fp2_dbl(oz)));
}
struct QuadAddTmp { Fp2 theta, mu, c, d; };
// wave 3, addition of (qx, qy): the line of T + Q evaluated at pa
BN_DEV void quad_add_line(Fp2& l0, Fp2& l1, Fp2& l2, QuadAddTmp& k, const G2Proj& t, const Fp2& qx, const Fp2& qy, const Fp2& PAX, const Fp2& PAY) {
  const Fp2 t1 = fp2_mul(qy, t.z), t2 = fp2_mul(qx, t.z);
  k.theta = NS(210, fp2_sub(t.y, t1)); k.mu = NS(211, fp2_sub(t.x, t2));
  const Fp2 ca = fp2_mul(k.theta, qx), cb = fp2_mul(k.mu, qy);
  l1 = fp2_mul(fp2_neg(k.theta), PAX);
  l0 = fp2_mul(k.mu, PAY);
  l2 = NS(223, fp2_sub(ca, cb));
}
// ... and the two squares the update starts from (in the same interval as the line, which is the shorter one of an addition)
BN_DEV void quad_add_squares(QuadAddTmp& k) { k.c = fp2_sqr(k.theta); k.d = fp2_sqr(k.mu); }
BN_DEV void quad_add_update(G2Proj& t, const QuadAddTmp& k) {
  const Fp2 e = fp2_mul(k.mu, k.d), ff = fp2_mul(t.z, k.c), g = fp2_mul(t.x, k.d);
  const Fp2 h = NS(212, fp2_sub(fp2_sub(fp2_add(e, ff), g), g));
  const Fp2 ox = fp2_mul(k.mu, h), oy1 = fp2_mul(k.theta, NS(213, fp2_sub(g, h))), oy2 = fp2_mul(e, t.y), oz = fp2_mul(t.z, e);
  t.x = ox;
  t.y = NS(214, fp2_sub(oy1, oy2));
  t.z = oz;
}
// wave 2: (l0 + l1 w + l2 w^3) * (table line idx at pb); m1 = C1 * x_B of the table line comes from the caller (it does not
// depend on the step's own line: the wave computes it in the lighter interval of the step before)
BN_DEV Fp2 quad_table_m1(int idx, const Fp2& PBX) { return fp2_mul(fp2_load_const(C_NEG_G2_LINES[idx][1]), PBX); }
BN_DEV void quad_line_product(TrioLineProduct& L, Fp2 l0, Fp2 l1, Fp2 l2, int idx, const Fp2& m1, const Fp2& PBY, bool skip_a, bool skip_b,
                              bool any_skip) {
  if (any_skip) {
    l2 = fp2_select(skip_a, fp2_zero(), l2);
    l0 = fp2_select(skip_a, fp2_one(), l0);
    l1 = fp2_select(skip_a, fp2_zero(), l1);
  }
  const Fp2 m0 = fp2_mul(fp2_load_const(C_NEG_G2_LINES[idx][0]), PBY);
  const Fp2 w3p = fp2_mul(l2, m0), w4p = fp2_mul(l2, m1), v0 = fp2_mul(l0, m0), v1 = fp2_mul(l1, m1);
  const Fp2 x01p = fp2_mul(fp2_add(l0, l1), fp2_add(m0, m1));
  trio_line_product(L, l0, l1, l2, v0, v1, w3p, w4p, skip_b, any_skip);
  trio_line_product_b10(L, l1, v0, v1, x01p, skip_b, any_skip);
}
// the six products of a Karatsuba Fq6 product, one after the other
BN_DEV void quad_kprod(Fp2 (&p)[6], const Fp6& x, const Fp6& y) {
  p[0] = fp2_mul(fp6_kop<0>(x), fp6_kop<0>(y)); p[1] = fp2_mul(fp6_kop<1>(x), fp6_kop<1>(y)); p[2] = fp2_mul(fp6_kop<2>(x), fp6_kop<2>(y));
  p[3] = fp2_mul(fp6_kop<3>(x), fp6_kop<3>(y)); p[4] = fp2_mul(fp6_kop<4>(x), fp6_kop<4>(y)); p[5] = fp2_mul(fp6_kop<5>(x), fp6_kop<5>(y));
}
// waves 0 / 1, squaring (sites 400 .. 419 as in trio_dbl_iter)
BN_DEV void quad_sqr_ab(Fp6& ab, const Fp6& f0, const Fp6& f1) { Fp2 p[6]; quad_kprod(p, f0, f1); fp6_kfin<406>(ab, p); }
BN_DEV void quad_sqr_u(Fp6& u, const Fp6& f0, const Fp6& f1) {
  Fp6 s, w;
  fp6_add(s, f0, f1); fp6_site_n<400>(s, s);
  fp6_mul_v(w, f1); fp6_add(w, w, f0); fp6_site_n<403>(w, w);
  Fp2 p[6]; quad_kprod(p, s, w); fp6_kfin<410>(u, p);
}
// the recombinations one COEFFICIENT at a time (the six coefficients of g and of the new f are spread over the four waves):
// coefficient K of v x is x_{K-1}, or xi x_2 for K = 0 — the caller passes that neighbour as `below`
template <int K> BN_DEV Fp2 quad_v_coef(const Fp2& below) { if constexpr (K == 0) return fp2_mul_xi(below); else return below; }
template <int K> BN_DEV Fp2 quad_g1_coef(const Fp2& ab_k) { return NR(417 + K, fp2_add(ab_k, ab_k)); }
template <int K> BN_DEV Fp2 quad_g0_coef(const Fp2& u_k, const Fp2& ab_k, const Fp2& ab_below) {
  return NR(414 + K, fp2_sub(fp2_sub(u_k, ab_k), quad_v_coef<K>(ab_below)));
}
template <int K> BN_DEV Fp2 quad_r0_coef(const Fp2& t0_k, const Fp2& t1_below) { return NR(437 + K, fp2_add(t0_k, quad_v_coef<K>(t1_below))); }
template <int K> BN_DEV Fp2 quad_r1_coef(const Fp2& uu_k, const Fp2& t0_k, const Fp2& t1_k) { return NR(440 + K, fp2_sub(fp2_sub(uu_k, t0_k), t1_k)); }
BN_DEV void quad_sqr_g1(Fp6& g1, const Fp6& ab) { g1.c0 = quad_g1_coef<0>(ab.c0); g1.c1 = quad_g1_coef<1>(ab.c1); g1.c2 = quad_g1_coef<2>(ab.c2); }
BN_DEV void quad_sqr_g0(Fp6& g0, const Fp6& u, const Fp6& ab) {
  g0.c0 = quad_g0_coef<0>(u.c0, ab.c0, ab.c2); g0.c1 = quad_g0_coef<1>(u.c1, ab.c1, ab.c0); g0.c2 = quad_g0_coef<2>(u.c2, ab.c2, ab.c1);
}
// waves 0 / 1 / 2, g * (b0 + (b10 + b11 v) w) (sites 420 .. 445 as in trio_line_mul_prepare / _finish)
BN_DEV void quad_mul_t0(Fp6& t0, const Fp6& g0, const Fp6& b0) { Fp2 p[6]; quad_kprod(p, g0, b0); fp6_kfin<426>(t0, p); }
BN_DEV void quad_mul_uu(Fp6& uu, const Fp6& g0, const Fp6& g1, const TrioLineProduct& L) {
  Fp12 g; g.c0 = g0; g.c1 = g1;
  TrioLineMul M;
  trio_line_mul_prepare(M, g, L);
  Fp2 p[6]; quad_kprod(p, M.sg, M.bs); fp6_kfin<430>(uu, p);
}
BN_DEV void quad_t1_from_products(Fp6& t1, const Fp2 (&p)[5]) {
  t1.c0 = NS(434, fp2_add(fp2_mul_xi(p[2]), p[0]));
  t1.c1 = NS(435, fp2_sub(fp2_sub(p[3], p[0]), p[1]));
  t1.c2 = NS(436, fp2_add(p[4], p[1]));
}
BN_DEV void quad_mul_t1(Fp6& t1, const Fp6& g1, const Fp2& b10, const Fp2& b11) {
  Fp2 p[5];
  p[0] = fp2_mul(g1.c0, b10); p[1] = fp2_mul(g1.c1, b11); p[2] = fp2_mul(g1.c2, b11);
  p[3] = fp2_mul(fp2_add(g1.c0, g1.c1), fp2_add(b10, b11)); p[4] = fp2_mul(g1.c2, b10);
  quad_t1_from_products(t1, p);
}
BN_DEV void quad_mul_r0(Fp6& r0, const Fp6& t0, const Fp6& t1) {
  r0.c0 = quad_r0_coef<0>(t0.c0, t1.c2); r0.c1 = quad_r0_coef<1>(t0.c1, t1.c0); r0.c2 = quad_r0_coef<2>(t0.c2, t1.c1);
}
BN_DEV void quad_mul_r1(Fp6& r1, const Fp6& uu, const Fp6& t0, const Fp6& t1) {
  r1.c0 = quad_r1_coef<0>(uu.c0, t0.c0, t1.c0); r1.c1 = quad_r1_coef<1>(uu.c1, t0.c1, t1.c1); r1.c2 = quad_r1_coef<2>(uu.c2, t0.c2, t1.c2);
}
// the sequence of steps of the loop: 64 doublings, an addition of +-Q after those with a non-zero digit, then + pi(Q), - pi^2(Q)
struct QuadSteps { int d, sub, k; };                      // digit index, 0 = doubling / 1 = its addition, step counter
BN_DEV QuadSteps quad_steps_begin() { QuadSteps s; s.d = 0; s.sub = 0; s.k = 0; return s; }
// type of the current step: 0 doubling, 1 / -1 addition of +-Q, 2 addition of pi(Q), 3 of -pi^2(Q), 4 = past the end
BN_DEV int quad_step_type(const QuadSteps& s) { return s.d < 64 ? (s.sub == 0 ? 0 : (int)C_ATE_NAF[s.d]) : s.d == 64 ? 2 : s.d == 65 ? 3 : 4; }
BN_DEV void quad_step_next(QuadSteps& s) {
  ++s.k;
  if (s.d >= 64) { ++s.d; return; }
  if (s.sub == 0 && C_ATE_NAF[s.d] != 0) { s.sub = 1; return; }
  s.sub = 0; ++s.d;
}
#if !defined(BN_TRIO_DEVICE)
// host model of the quad-wave kernel: the roles' functions in the order the barriers impose
BN_DEVM void miller_verify_quad_model(Fp12& f, const G1Affine& pa, const G2Affine& qa, const G1Affine& pb) {
  fp12_set_one(f);
  G2Proj t;
  const bool skip_a = pa.inf || qa.inf, skip_b = pb.inf, any_skip = skip_a || skip_b;
  t.x = qa.x; t.y = qa.y; t.z = fp2_one();
  const Fp2 qa_yneg = fp2_neg(qa.y);
  const Fp2 PAX = fp2_from_fp(pa.x), PAY = fp2_from_fp(pa.y), PBX = fp2_from_fp(pb.x), PBY = fp2_from_fp(pb.y);
  for (QuadSteps s = quad_steps_begin(); quad_step_type(s) != 4; quad_step_next(s)) {
    const int ty = quad_step_type(s);
    Fp2 l0, l1, l2;
    if (ty == 0) { QuadDblTmp k; quad_dbl_line(l0, l1, l2, k, t, PAX, PAY); quad_dbl_update(t, k); }
    else {
      Fp2 qx = qa.x, qy = ty > 0 ? qa.y : qa_yneg;
      if (ty == 2) { qx = fp2_mul(fp2_conj(qa.x), fp2_load_const(C_TW_FROB_X1)); qy = fp2_mul(fp2_conj(qa.y), fp2_load_const(C_TW_FROB_Y1)); }
      if (ty == 3) { qx = fp2_mul(qa.x, fp2_load_const(C_TW_FROB_X2)); qy = qa.y; }
      QuadAddTmp k; quad_add_line(l0, l1, l2, k, t, qx, qy, PAX, PAY); quad_add_squares(k); quad_add_update(t, k);
    }
    TrioLineProduct L;
    quad_line_product(L, l0, l1, l2, s.k, quad_table_m1(s.k, PBX), PBY, skip_a, skip_b, any_skip);
    Fp6 g0 = f.c0, g1 = f.c1;
    if (ty == 0) { Fp6 ab, u; quad_sqr_ab(ab, f.c0, f.c1); quad_sqr_u(u, f.c0, f.c1); quad_sqr_g1(g1, ab); quad_sqr_g0(g0, u, ab); }
    Fp6 t0, uu, t1;
    quad_mul_t0(t0, g0, L.b0); quad_mul_uu(uu, g0, g1, L); quad_mul_t1(t1, g1, L.b10, L.b11);
    quad_mul_r0(f.c0, t0, t1); quad_mul_r1(f.c1, uu, t0, t1);
  }
}
#endif

// ---- EIGHT wave roles (bn254_quad.hip: k_miller_verify_w8) ------------------------------------------------------------------
// With four waves every wave carries ~12 products per doubling step, one after the other.  Eight waves (two per SIMD of a
// CU, each still at the lone-wave issue rate) halve that: every Karatsuba Fq6 product of the f-chain is split over two
// waves (products K = 0, 1, 2 / K = 3, 4, 5), its coefficients are formed one per wave in a short phase after a barrier,
// the product of the lines and the twist-point step are split over two waves each:
//   A0, A1  ab = f0 f1, then t0 = g0 b0            B0, B1  u = (f0 + f1)(f0 + v f1), then uu = (g0 + g1)(b0 + b1)
//   L0, L1  (l0 + l1 w + l2 w^3)(m0 + m1 w + w^3) with b10 = l0 m1 + l1 m0 as two products (no cross dependency), then
//           t1 = g1 (b10 + b11 v)
//   T0, T1  the twist point, one step ahead: T0 the y / z side (b, c, e; theta), T1 the x side (y z, x^2; mu); h = 2 y z
// Phases of a step, each closed by a workgroup barrier: I1 (products) | C1 (coefficients of ab, u, the line product) |
// G (g0, g1; doubling steps only) | I2 (products) | C2 (coefficients of t0, uu, t1) | F (the new f).
// The functions are templates over the mailbox (`Box`: get(slot) / put(slot, value)): the kernel passes LDS, the host
// model an array — the same source runs in both, the model executing the eight roles of a phase one after the other.
enum { W8_F0 = 0, W8_F1 = 3, W8_P = 6, W8_C = 23, W8_LP = 32, W8_LINE = 37, W8_TC = 45, W8_TA = 48, W8_M0 = 51, W8_SLOTS = 52 };
// W8_P: 17 products (stage 1: ab 0..5, u 6..11; stage 2: t0 0..5, uu 6..11, t1 12..16).  W8_C: ab 0..2 | g0 3..5 | g1 6..8, in
// stage 2 t0 0..2 | t1 3..5; the six products of the line product use 3..8 between I1 and C1.  W8_LINE: 2 x (l0, l1, l2 | ca, cb).
// W8_M0: C0 * y_B of the next step's table line, computed by T1 (two products in its I2 against L0's three) for L0.
enum { W8_A0 = 0, W8_A1, W8_B0, W8_B1, W8_L0, W8_L1, W8_T0, W8_T1 };
struct W8In { Fp2 PAX, PAY, PBX, PBY; G2Affine pk; Fp2 pk_yneg; bool skip_a, skip_b, any_skip; };
struct W8Regs {                    // what a role keeps in registers between phases (each role uses its own subset)
  Fp2 p0, p1, p2;                  // A / B: own products of the current stage;  L: own products
  Fp2 c;                           // own coefficient (ab_k / u_k, later t0_k / uu_k)
  Fp2 l0, l1, l2, m;               // L: the step's line (after the skip selects), own table-line scaling m0 / m1
  Fp2 b10, b11;                    // L
  G2Proj t;                        // T
  Fp2 a0, a1, a2;                  // T: values of the step in flight (b, e | yz-side; theta, c | mu, d, e)
  Fp2 qx, qy;                      // T: the point added in the step in flight
};
template <class Box> BN_DEV void w8_get6(Fp6& x, Box& bx, int slot) { x.c0 = bx.get(slot); x.c1 = bx.get(slot + 1); x.c2 = bx.get(slot + 2); }
// operand K of a Karatsuba Fq6 product for K = 0..2 (h = 0) or 3..5 (h = 1), j = 0, 1, 2
template <int H, int J> BN_DEV Fp2 w8_kop(const Fp6& x) { return fp6_kop<3 * H + J>(x); }
template <int H> BN_DEV void w8_half_products(W8Regs& r, const Fp6& x, const Fp6& y) {
  r.p0 = fp2_mul(w8_kop<H, 0>(x), w8_kop<H, 0>(y)); r.p1 = fp2_mul(w8_kop<H, 1>(x), w8_kop<H, 1>(y)); r.p2 = fp2_mul(w8_kop<H, 2>(x), w8_kop<H, 2>(y));
}
// ---- phase I1
template <int ROLE, class Box> BN_DEV void w8_phase_i1(W8Regs& r, Box& bx, const W8In& in, int ty, int k, int ty_next) {
  if constexpr (ROLE == W8_A0 || ROLE == W8_A1 || ROLE == W8_B0 || ROLE == W8_B1) {
    if (ty != 0) return;                                        // no squaring in an addition step
    Fp6 f0, f1;
    w8_get6(f0, bx, W8_F0); w8_get6(f1, bx, W8_F1);
    constexpr int H = (ROLE == W8_A1 || ROLE == W8_B1) ? 1 : 0;
    if constexpr (ROLE == W8_A0 || ROLE == W8_A1) {
      w8_half_products<H>(r, f0, f1);
      bx.put(W8_P + 3 * H, r.p0); bx.put(W8_P + 3 * H + 1, r.p1); bx.put(W8_P + 3 * H + 2, r.p2);
    } else {
      Fp6 s, w;
      fp6_add(s, f0, f1); fp6_site_n<400>(s, s);
      fp6_mul_v(w, f1); fp6_add(w, w, f0); fp6_site_n<403>(w, w);
      w8_half_products<H>(r, s, w);
      bx.put(W8_P + 6 + 3 * H, r.p0); bx.put(W8_P + 6 + 3 * H + 1, r.p1); bx.put(W8_P + 6 + 3 * H + 2, r.p2);
    }
  } else if constexpr (ROLE == W8_L0 || ROLE == W8_L1) {
    const int at = W8_LINE + 4 * (k & 1);
    r.l0 = bx.get(at); r.l1 = bx.get(at + 1);
    if (ty == 0) r.l2 = bx.get(at + 2);
    else r.l2 = NS(223, fp2_sub(bx.get(at + 2), bx.get(at + 3)));        // theta x_Q - mu y_Q
    if (in.any_skip) {
      r.l2 = fp2_select(in.skip_a, fp2_zero(), r.l2);
      r.l0 = fp2_select(in.skip_a, fp2_one(), r.l0);
      r.l1 = fp2_select(in.skip_a, fp2_zero(), r.l1);
    }
    if constexpr (ROLE == W8_L0) {                               // m = m0 (from T1): w3p = l2 m0, v0 = l0 m0, l1 m0
      r.m = bx.get(W8_M0);
      r.p0 = fp2_mul(r.l2, r.m); r.p1 = fp2_mul(r.l0, r.m); r.p2 = fp2_mul(r.l1, r.m);
      bx.put(W8_C + 3, r.p0); bx.put(W8_C + 4, r.p1); bx.put(W8_C + 5, r.p2);
    } else {                                                     // m = m1: w4p = l2 m1, v1 = l1 m1, l0 m1
      r.p0 = fp2_mul(r.l2, r.m); r.p1 = fp2_mul(r.l1, r.m); r.p2 = fp2_mul(r.l0, r.m);
      bx.put(W8_C + 6, r.p0); bx.put(W8_C + 7, r.p1); bx.put(W8_C + 8, r.p2);
    }
  } else {                                                       // T0 / T1: the line of the NEXT step (type ty_next, index k + 1)
    if (ty_next == 4) return;
    r.t.x = bx.get(W8_TC); r.t.y = bx.get(W8_TC + 1); r.t.z = bx.get(W8_TC + 2);
    const int at = W8_LINE + 4 * ((k + 1) & 1);
    if (ty_next == 0) {
      if constexpr (ROLE == W8_T0) {
        const Fp2 b = fp2_mul(r.t.y, r.t.y), c = fp2_sqr(r.t.z);
        const Fp2 e = fp2_mul(c, fp2_load_const(C_TWIST_3B));
        bx.put(at + 2, NS(223, fp2_sub(b, e)));
        r.a0 = b; r.a1 = e;
        bx.put(W8_TA, b); bx.put(W8_TA + 1, e);
      } else {
        const Fp2 yz = fp2_mul(r.t.y, r.t.z), x2 = fp2_sqr(r.t.x);
        const Fp2 h = NS(207, fp2_dbl(yz));                     // (y + z)^2 - y^2 - z^2
        bx.put(at, fp2_mul(h, in.PAY));
        bx.put(at + 1, fp2_mul(fp2_neg(fp2_add(fp2_dbl(x2), x2)), in.PAX));
        r.a0 = h;
        bx.put(W8_TA + 2, h);
      }
    } else {
      // theta = Y - y_Q Z and mu = X - x_Q Z go out with y_Q / x_Q (mailbox slots of the t1 products, idle until I2): the four
      // products with them that make up the line — l1 = -theta x_A, theta x_Q, l0 = mu y_A, mu y_Q — are computed by the four
      // f waves in the C1 phase of this step (w8_phase_c1), which have nothing else to do there in an addition step's
      // neighbourhood; the twist waves keep the chains theta -> theta^2 -> Z theta^2 and mu -> mu^2 -> mu^3, X mu^2.
      if constexpr (ROLE == W8_T0) {                             // theta side, y_Q
        r.qy = ty_next == -1 ? in.pk_yneg : in.pk.y;
        if (ty_next == 2) r.qy = fp2_mul(fp2_conj(in.pk.y), fp2_load_const(C_TW_FROB_Y1));
        const Fp2 theta = NS(210, fp2_sub(r.t.y, fp2_mul(r.qy, r.t.z)));
        bx.put(W8_P + 12, theta); bx.put(W8_P + 15, r.qy);
        const Fp2 c = fp2_sqr(theta);
        r.a0 = theta;
        bx.put(W8_TA, fp2_mul(r.t.z, c));                        // F = Z theta^2
      } else {                                                   // mu side, x_Q
        r.qx = in.pk.x;
        if (ty_next == 2) r.qx = fp2_mul(fp2_conj(in.pk.x), fp2_load_const(C_TW_FROB_X1));
        if (ty_next == 3) r.qx = fp2_mul(in.pk.x, fp2_load_const(C_TW_FROB_X2));
        const Fp2 mu = NS(211, fp2_sub(r.t.x, fp2_mul(r.qx, r.t.z)));
        bx.put(W8_P + 13, mu); bx.put(W8_P + 14, r.qx);
        const Fp2 d = fp2_sqr(mu);
        const Fp2 e = fp2_mul(mu, d), g = fp2_mul(r.t.x, d);
        r.a0 = mu; r.a1 = e;
        bx.put(W8_TA + 1, e); bx.put(W8_TA + 2, g);
      }
    }
  }
}
// ---- phase C1: coefficients of ab and u (doubling steps), the line product
template <int ROLE, class Box> BN_DEV void w8_phase_c1(W8Regs& r, Box& bx, const W8In& in, int ty, int k, int ty_next) {
  if constexpr (ROLE == W8_A0 || ROLE == W8_A1 || ROLE == W8_B0 || ROLE == W8_B1) {
    if (ty_next != 0 && ty_next != 4) {                          // the next step adds a point: its line from theta, mu (see w8_phase_i1)
      const int at = W8_LINE + 4 * ((k + 1) & 1);
      if constexpr (ROLE == W8_A0) bx.put(at, fp2_mul(bx.get(W8_P + 13), in.PAY));                    // l0 = mu y_A
      else if constexpr (ROLE == W8_A1) bx.put(at + 3, fp2_mul(bx.get(W8_P + 13), bx.get(W8_P + 15)));  // mu y_Q
      else if constexpr (ROLE == W8_B0) bx.put(at + 1, fp2_mul(fp2_neg(bx.get(W8_P + 12)), in.PAX));   // l1 = -theta x_A
      else bx.put(at + 2, fp2_mul(bx.get(W8_P + 12), bx.get(W8_P + 14)));                              // theta x_Q
    }
  }
  if constexpr (ROLE == W8_A0) { if (ty == 0) { r.c = fp6_kfin_coef<406, 0>(r.p0, r.p1, r.p2, bx.get(W8_P + 3)); bx.put(W8_C, r.c); } }
  else if constexpr (ROLE == W8_A1) { if (ty == 0) { r.c = fp6_kfin_coef<406, 1>(bx.get(W8_P), bx.get(W8_P + 1), bx.get(W8_P + 2), r.p1); bx.put(W8_C + 1, r.c); } }
  else if constexpr (ROLE == W8_T0) { if (ty == 0) { r.a2 = fp6_kfin_coef<406, 2>(bx.get(W8_P), bx.get(W8_P + 1), bx.get(W8_P + 2), bx.get(W8_P + 5)); bx.put(W8_C + 2, r.a2); } }
  else if constexpr (ROLE == W8_B0) { if (ty == 0) r.c = fp6_kfin_coef<410, 0>(r.p0, r.p1, r.p2, bx.get(W8_P + 9)); }
  else if constexpr (ROLE == W8_B1) { if (ty == 0) r.c = fp6_kfin_coef<410, 1>(bx.get(W8_P + 6), bx.get(W8_P + 7), bx.get(W8_P + 8), r.p1); }
  else if constexpr (ROLE == W8_T1) { if (ty == 0) r.a2 = fp6_kfin_coef<410, 2>(bx.get(W8_P + 6), bx.get(W8_P + 7), bx.get(W8_P + 8), bx.get(W8_P + 11)); }
  else if constexpr (ROLE == W8_L0) {                            // own: w3p, v0, l1 m0
    Fp2 b00 = NS(224, fp2_add(r.p1, fp2_mul_xi(r.l2)));
    r.b11 = NS(227, fp2_add(r.l0, r.p0));
    if (in.any_skip) { b00 = fp2_select(in.skip_b, r.l0, b00); r.b11 = fp2_select(in.skip_b, r.l2, r.b11); }
    bx.put(W8_LP, b00); bx.put(W8_LP + 4, r.b11);
  } else {                                                       // L1, own: w4p, v1, l0 m1
    Fp2 b01 = r.p1, b02 = NS(225, fp2_add(r.l1, r.p0));
    r.b10 = NS(226, fp2_add(r.p2, bx.get(W8_C + 5)));            // l0 m1 + l1 m0
    if (in.any_skip) {
      b01 = fp2_select(in.skip_b, fp2_zero(), b01); b02 = fp2_select(in.skip_b, fp2_zero(), b02);
      r.b10 = fp2_select(in.skip_b, r.l1, r.b10);
    }
    bx.put(W8_LP + 1, b01); bx.put(W8_LP + 2, b02); bx.put(W8_LP + 3, r.b10);
  }
}
// ---- phase G (doubling steps): g1_k = 2 ab_k, g0_k = u_k - ab_k - (v ab)_k
template <int ROLE, class Box> BN_DEV void w8_phase_g(W8Regs& r, Box& bx) {
  if constexpr (ROLE == W8_A0) bx.put(W8_C + 6, quad_g1_coef<0>(r.c));
  else if constexpr (ROLE == W8_A1) bx.put(W8_C + 7, quad_g1_coef<1>(r.c));
  else if constexpr (ROLE == W8_T0) bx.put(W8_C + 8, quad_g1_coef<2>(r.a2));
  else if constexpr (ROLE == W8_B0) bx.put(W8_C + 3, quad_g0_coef<0>(r.c, bx.get(W8_C), bx.get(W8_C + 2)));
  else if constexpr (ROLE == W8_B1) bx.put(W8_C + 4, quad_g0_coef<1>(r.c, bx.get(W8_C + 1), bx.get(W8_C)));
  else if constexpr (ROLE == W8_T1) bx.put(W8_C + 5, quad_g0_coef<2>(r.a2, bx.get(W8_C + 2), bx.get(W8_C + 1)));
}
// ---- phase I2: products of g * (line product); the twist point's update; the table-line scaling of the next step
template <int ROLE, class Box> BN_DEV void w8_phase_i2(W8Regs& r, Box& bx, const W8In& in, int ty, int k, int ty_next) {
  const int g0_at = ty == 0 ? W8_C + 3 : W8_F0, g1_at = ty == 0 ? W8_C + 6 : W8_F1;
  if constexpr (ROLE == W8_A0 || ROLE == W8_A1) {
    constexpr int H = ROLE == W8_A1 ? 1 : 0;
    Fp6 g0, b0;
    w8_get6(g0, bx, g0_at); w8_get6(b0, bx, W8_LP);
    w8_half_products<H>(r, g0, b0);
    bx.put(W8_P + 3 * H, r.p0); bx.put(W8_P + 3 * H + 1, r.p1); bx.put(W8_P + 3 * H + 2, r.p2);
  } else if constexpr (ROLE == W8_B0 || ROLE == W8_B1) {
    constexpr int H = ROLE == W8_B1 ? 1 : 0;
    Fp12 g;
    TrioLineProduct L;
    w8_get6(g.c0, bx, g0_at); w8_get6(g.c1, bx, g1_at);
    w8_get6(L.b0, bx, W8_LP); L.b10 = bx.get(W8_LP + 3); L.b11 = bx.get(W8_LP + 4);
    TrioLineMul M;
    trio_line_mul_prepare(M, g, L);
    w8_half_products<H>(r, M.sg, M.bs);
    bx.put(W8_P + 6 + 3 * H, r.p0); bx.put(W8_P + 6 + 3 * H + 1, r.p1); bx.put(W8_P + 6 + 3 * H + 2, r.p2);
  } else if constexpr (ROLE == W8_L0) {
    Fp6 g1;
    w8_get6(g1, bx, g1_at);
    const Fp2 b10 = bx.get(W8_LP + 3);
    r.p0 = fp2_mul(g1.c0, b10); r.p1 = fp2_mul(g1.c1, r.b11); r.p2 = fp2_mul(g1.c2, r.b11);
    bx.put(W8_P + 12, r.p0); bx.put(W8_P + 13, r.p1); bx.put(W8_P + 14, r.p2);
  } else if constexpr (ROLE == W8_L1) {
    Fp6 g1;
    w8_get6(g1, bx, g1_at);
    const Fp2 b11 = bx.get(W8_LP + 4);
    r.p0 = fp2_mul(fp2_add(g1.c0, g1.c1), fp2_add(r.b10, b11)); r.p1 = fp2_mul(g1.c2, r.b10);
    bx.put(W8_P + 15, r.p0); bx.put(W8_P + 16, r.p1);
    if (k + 1 < BN_N_FIXED_LINES) r.m = fp2_mul(fp2_load_const(C_NEG_G2_LINES[k + 1][1]), in.PBX);
  } else {                                                       // T0 / T1: update for the step whose line went out in I1
    if constexpr (ROLE == W8_T1) { if (k + 1 < BN_N_FIXED_LINES) bx.put(W8_M0, fp2_mul(fp2_load_const(C_NEG_G2_LINES[k + 1][0]), in.PBY)); }
    if (ty_next == 4) return;
    if (ty_next == 0) {
      if constexpr (ROLE == W8_T0) {                             // y, z of 2T: has b, e; h from T1
        const Fp2 h = bx.get(W8_TA + 2);
        const Fp2 e2 = fp2_sqr(r.a1), f3 = fp2_add(fp2_dbl(r.a1), r.a1);
        const Fp2 bf = NS(204, fp2_add(r.a0, f3));
        const Fp2 oz = fp2_mul(r.a0, h), oy2 = fp2_sqr(bf);
        const Fp2 e2x4 = NS(201, fp2_dbl(fp2_dbl(e2)));
        bx.put(W8_TC + 1, NS(205, fp2_sub(oy2, fp2_add(fp2_dbl(e2x4), e2x4))));
        bx.put(W8_TC + 2, NS(206, fp2_dbl(fp2_dbl(oz))));
      } else {                                                   // x of 2T: b, e from T0
        const Fp2 b = bx.get(W8_TA), e = bx.get(W8_TA + 1);
        const Fp2 f3 = fp2_add(fp2_dbl(e), e);
        const Fp2 xy = fp2_mul(r.t.x, r.t.y);
        bx.put(W8_TC, NS(203, fp2_mul(fp2_dbl(xy), NS(202, fp2_sub(b, f3)))));
      }
    } else {
      const Fp2 ff = bx.get(W8_TA), e = bx.get(W8_TA + 1), g = bx.get(W8_TA + 2);
      const Fp2 h = NS(212, fp2_sub(fp2_sub(fp2_add(e, ff), g), g));
      if constexpr (ROLE == W8_T0) {                             // Y3 = theta (G - H) - E Y
        const Fp2 oy1 = fp2_mul(r.a0, NS(213, fp2_sub(g, h))), oy2 = fp2_mul(e, r.t.y);
        bx.put(W8_TC + 1, NS(214, fp2_sub(oy1, oy2)));
      } else {                                                   // X3 = mu H, Z3 = Z E
        bx.put(W8_TC, fp2_mul(r.a0, h));
        bx.put(W8_TC + 2, fp2_mul(r.t.z, e));
      }
    }
  }
}
// ---- phase C2: coefficients of t0, uu, t1
template <int ROLE, class Box> BN_DEV void w8_phase_c2(W8Regs& r, Box& bx) {
  if constexpr (ROLE == W8_A0) { r.c = fp6_kfin_coef<426, 0>(r.p0, r.p1, r.p2, bx.get(W8_P + 3)); bx.put(W8_C, r.c); }
  else if constexpr (ROLE == W8_A1) { r.c = fp6_kfin_coef<426, 1>(bx.get(W8_P), bx.get(W8_P + 1), bx.get(W8_P + 2), r.p1); bx.put(W8_C + 1, r.c); }
  else if constexpr (ROLE == W8_T0) { r.a2 = fp6_kfin_coef<426, 2>(bx.get(W8_P), bx.get(W8_P + 1), bx.get(W8_P + 2), bx.get(W8_P + 5)); bx.put(W8_C + 2, r.a2); }
  else if constexpr (ROLE == W8_B0) r.c = fp6_kfin_coef<430, 0>(r.p0, r.p1, r.p2, bx.get(W8_P + 9));
  else if constexpr (ROLE == W8_B1) r.c = fp6_kfin_coef<430, 1>(bx.get(W8_P + 6), bx.get(W8_P + 7), bx.get(W8_P + 8), r.p1);
  else if constexpr (ROLE == W8_T1) r.a2 = fp6_kfin_coef<430, 2>(bx.get(W8_P + 6), bx.get(W8_P + 7), bx.get(W8_P + 8), bx.get(W8_P + 11));
  else if constexpr (ROLE == W8_L0) {                            // t1_0 = xi p2 + p0, t1_2 = p4 + p1
    bx.put(W8_C + 3, NS(434, fp2_add(fp2_mul_xi(r.p2), r.p0)));
    bx.put(W8_C + 5, NS(436, fp2_add(bx.get(W8_P + 16), r.p1)));
  } else bx.put(W8_C + 4, NS(435, fp2_sub(fp2_sub(r.p0, bx.get(W8_P + 12)), bx.get(W8_P + 13))));    // L1: t1_1 = p3 - p0 - p1
}
// ---- phase F: f0_k = t0_k + (v t1)_k, f1_k = uu_k - t0_k - t1_k
template <int ROLE, class Box> BN_DEV void w8_phase_f(W8Regs& r, Box& bx) {
  if constexpr (ROLE == W8_A0) bx.put(W8_F0, quad_r0_coef<0>(r.c, bx.get(W8_C + 5)));
  else if constexpr (ROLE == W8_A1) bx.put(W8_F0 + 1, quad_r0_coef<1>(r.c, bx.get(W8_C + 3)));
  else if constexpr (ROLE == W8_T0) bx.put(W8_F0 + 2, quad_r0_coef<2>(r.a2, bx.get(W8_C + 4)));
  else if constexpr (ROLE == W8_B0) bx.put(W8_F1, quad_r1_coef<0>(r.c, bx.get(W8_C), bx.get(W8_C + 3)));
  else if constexpr (ROLE == W8_B1) bx.put(W8_F1 + 1, quad_r1_coef<1>(r.c, bx.get(W8_C + 1), bx.get(W8_C + 4)));
  else if constexpr (ROLE == W8_T1) bx.put(W8_F1 + 2, quad_r1_coef<2>(r.a2, bx.get(W8_C + 2), bx.get(W8_C + 5)));
}
// before the loop: f = 1, T = Q, the table-line scalings of step 0; then the T waves run I1 / I2 once for step 0 (k = -1)
template <int ROLE, class Box> BN_DEV void w8_init(W8Regs& r, Box& bx, const W8In& in) {
  if constexpr (ROLE == W8_A0) { bx.put(W8_F0, fp2_one()); bx.put(W8_F0 + 1, fp2_zero()); bx.put(W8_F0 + 2, fp2_zero()); }
  else if constexpr (ROLE == W8_B0) { bx.put(W8_F1, fp2_zero()); bx.put(W8_F1 + 1, fp2_zero()); bx.put(W8_F1 + 2, fp2_zero()); }
  else if constexpr (ROLE == W8_L1) r.m = fp2_mul(fp2_load_const(C_NEG_G2_LINES[0][1]), in.PBX);
  else if constexpr (ROLE == W8_T0) { bx.put(W8_TC + 1, in.pk.y); bx.put(W8_TC + 2, fp2_one()); }
  else if constexpr (ROLE == W8_T1) bx.put(W8_TC, in.pk.x);
}
#if !defined(BN_TRIO_DEVICE) && !defined(BN_QUAD_DEVICE)
// host model of k_miller_verify_w8: the same role / phase functions over an array mailbox, the eight roles of a phase one
// after the other, the phases in barrier order
struct W8HostBox { Fp2 v[W8_SLOTS]; Fp2 get(int s) const { return v[s]; } void put(int s, const Fp2& x) { v[s] = x; } };
#define W8_ALL_ROLES(CALL) CALL(W8_A0) CALL(W8_A1) CALL(W8_B0) CALL(W8_B1) CALL(W8_L0) CALL(W8_L1) CALL(W8_T0) CALL(W8_T1)
BN_DEVM void miller_verify_w8_model(Fp12& f, const G1Affine& pa, const G2Affine& qa, const G1Affine& pb) {
  W8In in;
  in.skip_a = pa.inf || qa.inf; in.skip_b = pb.inf; in.any_skip = in.skip_a || in.skip_b;
  in.pk = qa; in.pk_yneg = fp2_neg(qa.y);
  in.PAX = fp2_from_fp(pa.x); in.PAY = fp2_from_fp(pa.y); in.PBX = fp2_from_fp(pb.x); in.PBY = fp2_from_fp(pb.y);
  W8HostBox bx;
  for (int i = 0; i < W8_SLOTS; ++i) bx.v[i] = fp2_zero();
  W8Regs r[8];
  for (int w = 0; w < 8; ++w) { r[w].p0 = r[w].p1 = r[w].p2 = r[w].c = r[w].l0 = r[w].l1 = r[w].l2 = r[w].m = r[w].b10 = r[w].b11 = fp2_zero();
    r[w].a0 = r[w].a1 = r[w].a2 = r[w].qx = r[w].qy = fp2_zero(); r[w].t.x = r[w].t.y = r[w].t.z = fp2_zero(); }
#define W8_INIT(R) w8_init<R>(r[R], bx, in);
  W8_ALL_ROLES(W8_INIT)
  QuadSteps s = quad_steps_begin();
  {                                                              // the twist waves' head start: line and update of step 0
    const int ty0 = quad_step_type(s);
    w8_phase_i1<W8_T0>(r[W8_T0], bx, in, 4, -1, ty0); w8_phase_i1<W8_T1>(r[W8_T1], bx, in, 4, -1, ty0);
    w8_phase_i2<W8_T0>(r[W8_T0], bx, in, 4, -1, ty0); w8_phase_i2<W8_T1>(r[W8_T1], bx, in, 4, -1, ty0);
  }
  for (; quad_step_type(s) != 4; quad_step_next(s)) {
    const int ty = quad_step_type(s);
    QuadSteps nx = s; quad_step_next(nx);
    const int ty_next = quad_step_type(nx);
#define W8_I1(R) w8_phase_i1<R>(r[R], bx, in, ty, s.k, ty_next);
#define W8_C1(R) w8_phase_c1<R>(r[R], bx, in, ty, s.k, ty_next);
#define W8_G(R) w8_phase_g<R>(r[R], bx);
#define W8_I2(R) w8_phase_i2<R>(r[R], bx, in, ty, s.k, ty_next);
#define W8_C2(R) w8_phase_c2<R>(r[R], bx);
#define W8_F(R) w8_phase_f<R>(r[R], bx);
    W8_ALL_ROLES(W8_I1)
    W8_ALL_ROLES(W8_C1)
    if (ty == 0) { W8_ALL_ROLES(W8_G) }
    W8_ALL_ROLES(W8_I2)
    W8_ALL_ROLES(W8_C2)
    W8_ALL_ROLES(W8_F)
  }
  f.c0.c0 = bx.v[W8_F0]; f.c0.c1 = bx.v[W8_F0 + 1]; f.c0.c2 = bx.v[W8_F0 + 2];
  f.c1.c0 = bx.v[W8_F1]; f.c1.c1 = bx.v[W8_F1 + 1]; f.c1.c2 = bx.v[W8_F1 + 2];
}
#endif
#endif

// f <- f * lineA(pa) * lineC(pc) for two variable lines: 6 Fq2 products for the line product (Karatsuba over
// the three coefficients) + 17 for f * (5-term element), against 2 x 13 one line at a time.
BN_DEV void mul_by_two_var_lines(Fp12& f, const LineCoef& la, const Fp& pax, const Fp& pay, bool skip_a, const LineCoef& lc, const Fp& pcx,
                                 const Fp& pcy, bool skip_c) {       // sites 230 .. 239
  Fp2 l0 = fp2_mul_fp(la.c0, pay), l1 = fp2_mul_fp(la.c1, pax), l2 = NS(230, la.c2);
  l0 = fp2_select(skip_a, fp2_one(), l0);
  l1 = fp2_select(skip_a, fp2_zero(), l1);
  l2 = fp2_select(skip_a, fp2_zero(), l2);
  Fp2 m0 = fp2_mul_fp(lc.c0, pcy), m1 = fp2_mul_fp(lc.c1, pcx), m2 = NS(231, lc.c2);
  m0 = fp2_select(skip_c, fp2_one(), m0);
  m1 = fp2_select(skip_c, fp2_zero(), m1);
  m2 = fp2_select(skip_c, fp2_zero(), m2);
  Fp2 v0 = fp2_mul(l0, m0), v1 = fp2_mul(l1, m1), v2 = fp2_mul(l2, m2);
  Fp2 x01 = fp2_sub(fp2_sub(fp2_mul(fp2_add(l0, l1), fp2_add(m0, m1)), v0), v1);
  Fp2 x02 = fp2_sub(fp2_sub(fp2_mul(fp2_add(l0, l2), fp2_add(m0, m2)), v0), v2);
  Fp2 x12 = fp2_sub(fp2_sub(fp2_mul(fp2_add(l1, l2), fp2_add(m1, m2)), v1), v2);
  Fp6 b0;
  b0.c0 = NS(232, fp2_add(v0, fp2_mul_xi(v2)));
  b0.c1 = v1;
  b0.c2 = NS(233, x12);
  fp12_mul_line2(f, f, b0, NS(234, x01), NS(235, x02));
}

// Miller loop over two pairs with variable twist points sharing f (randomised batch verification: two
// items per lane).  A pair with an identity member contributes 1.
template <bool F_LDS = false>
BN_DEVM void miller_loop_2var(Fp12& f, const G1Affine& pa, const G2Affine& qa, const G1Affine& pc, const G2Affine& qc) {
  if constexpr (F_LDS) BN_ASSUME_LDS(&f);
  fp12_set_one(f);
  G2Proj ta, tc;
  LineCoef la, lc;
  const bool skip_a = pa.inf || qa.inf, skip_c = pc.inf || qc.inf;
  ta.x = qa.x; ta.y = qa.y; ta.z = fp2_one();
  tc.x = qc.x; tc.y = qc.y; tc.z = fp2_one();
  const Fp2 qa_yneg = fp2_neg(qa.y), qc_yneg = fp2_neg(qc.y);
  for (int d = 0; d < 64; ++d) {
    BN_SET_STEP_PRIORITY(d);
    fp12_sqr(f, f);
    dbl_step(ta, la);
    dbl_step(tc, lc);
    mul_by_two_var_lines(f, la, pa.x, pa.y, skip_a, lc, pc.x, pc.y, skip_c);
    int digit = C_ATE_NAF[d];
    if (digit != 0) {   // wave-uniform
      add_step(ta, la, qa.x, fp2_select(digit > 0, qa.y, qa_yneg));
      add_step(tc, lc, qc.x, fp2_select(digit > 0, qc.y, qc_yneg));
      mul_by_two_var_lines(f, la, pa.x, pa.y, skip_a, lc, pc.x, pc.y, skip_c);
    }
  }
  const Fp2 gx1 = fp2_load_const(C_TW_FROB_X1), gy1 = fp2_load_const(C_TW_FROB_Y1), gx2 = fp2_load_const(C_TW_FROB_X2);
  add_step(ta, la, fp2_mul(fp2_conj(qa.x), gx1), fp2_mul(fp2_conj(qa.y), gy1));
  add_step(tc, lc, fp2_mul(fp2_conj(qc.x), gx1), fp2_mul(fp2_conj(qc.y), gy1));
  mul_by_two_var_lines(f, la, pa.x, pa.y, skip_a, lc, pc.x, pc.y, skip_c);
  add_step(ta, la, fp2_mul(qa.x, gx2), qa.y);
  add_step(tc, lc, fp2_mul(qc.x, gx2), qc.y);
  mul_by_two_var_lines(f, la, pa.x, pa.y, skip_a, lc, pc.x, pc.y, skip_c);
}

// a^u for a in the cyclotomic subgroup (u = 4965661367192848881, 63 bits; a^-1 is the conjugate there): signed digits
// from the set {1, 15, 19} (bn254_constants.h, found by a search over small digit sets): the table costs 4 cyclotomic
// squarings and 2 multiplications (a^16, a^15 = a^16 / a, a^19 = a^15 * a^4), the 12 non-zero digits 11 more —
// 13 multiplications where width-4 signed windows took 16 and plain square-and-multiply 28.
// `acc` is caller-provided working storage (the kernels pass an LDS slot: the accumulator is read and rewritten by every
// squaring).
template <bool ACC_LDS = false>
BN_DEVN void fp12_pow_u(Fp12& r, const Fp12& a, Fp12& acc) {
  if constexpr (ACC_LDS) BN_ASSUME_LDS(&acc);
  Fp12 tab[3], t, a4;                               // a, a^15, a^19
  tab[0] = a;
  fp12_cyclotomic_sqr(t, a);
  fp12_cyclotomic_sqr(a4, t);
  fp12_cyclotomic_sqr(t, a4);
  fp12_cyclotomic_sqr(t, t);                        // a^16
  fp12_conj(tab[1], a);
  fp12_mul(tab[1], t, tab[1]);                      // a^15
  fp12_mul(tab[2], tab[1], a4);                     // a^19
  acc = tab[C_U_W4[0] == 1 ? 0 : C_U_W4[0] == 15 ? 1 : 2];   // leading digit is positive
  for (int i = 1; i < BN_U_W4_LEN; ++i) {           // wave-uniform: u is a public constant
    BN_SET_STEP_PRIORITY(i >> 1);                    // half the rate of the Miller loop's cycle (measured: -1.3 % here)
    fp12_cyclotomic_sqr_hot(acc, acc);
    const int d = C_U_W4[i];
    if (d != 0) {                                    // ONE multiplication site: the inlined body exists once
      const int ad = d < 0 ? -d : d;
      const Fp12* m = &tab[ad == 1 ? 0 : ad == 15 ? 1 : 2];
      if (d < 0) { fp12_conj(t, *m); m = &t; }
      fp12_mul_hot(acc, acc, *m);
    }
  }
  r = acc;
}

// easy part of the final exponentiation: f^((q^6 - 1)(q^2 + 1)), one Fq12 inversion
BN_DEV void final_exp_easy(Fp12& f, const Fp12& fin) {
  Fp12 t, a;
  fp12_inv(t, fin);
  fp12_conj(a, fin);
  fp12_mul(f, a, t);
  fp12_frob(t, f, 2);
  fp12_mul(f, t, f);
}
// f^((q^12-1)/r): easy part (q^6-1)(q^2+1), then the EXACT hard part (q^4-q^2+1)/r =
// q^3 + (6u^2+1) q^2 + (-36u^3-18u^2-12u+1) q + (-36u^3-30u^2-18u-2) by the vectorial
// addition chain y0 * y1^2 * y2^6 * y3^12 * y4^18 * y5^30 * y6^36 (13 multiplications, 4 squarings, 6 Frobenius maps
// beside the three exponentiations by u).  This is the canonical Gt value of the pairing API.
template <bool ACC_LDS = false>
BN_DEVN void final_exponentiation(Fp12& r, const Fp12& fin, Fp12& acc) {
  Fp12 f, a, b;
  final_exp_easy(f, fin);
  Fp12 fu, fu2, fu3, y0, y1, y2, y3, y4, y5, y6;
  fp12_pow_u<ACC_LDS>(fu, f, acc);
  fp12_pow_u<ACC_LDS>(fu2, fu, acc);
  fp12_pow_u<ACC_LDS>(fu3, fu2, acc);
  fp12_frob(a, f, 1); fp12_frob(b, f, 2); fp12_mul(y0, a, b); fp12_frob(a, f, 3); fp12_mul(y0, y0, a);
  fp12_conj(y1, f);
  fp12_frob(y2, fu2, 2);
  fp12_frob(a, fu, 1); fp12_conj(y3, a);
  fp12_frob(a, fu2, 1); fp12_mul(a, a, fu); fp12_conj(y4, a);
  fp12_conj(y5, fu2);
  fp12_frob(a, fu3, 1); fp12_mul(a, a, fu3); fp12_conj(y6, a);
  Fp12 t0, t1;
  BN_SET_STEP_PRIORITY(2);   // the short tail: whoever is still here is behind
  fp12_cyclotomic_sqr(t0, y6); fp12_mul(t0, t0, y4); fp12_mul(t0, t0, y5);
  fp12_mul(t1, y3, y5); fp12_mul(t1, t1, t0);
  fp12_mul(t0, t0, y2);
  fp12_cyclotomic_sqr(t1, t1); fp12_mul(t1, t1, t0); fp12_cyclotomic_sqr(t1, t1);
  fp12_mul(t0, t1, y1); fp12_mul(t1, t1, y0);
  fp12_cyclotomic_sqr(t0, t0);
  fp12_mul(r, t0, t1);
}
// The same test with a cheaper hard part — for the == Gt::one() comparison of ECDSA::verify / check_public_keys
// (/root/reference/src/ecdsa.rs:59, :88) ONLY: f^(m (q^12-1)/r) with m = 2u(6u^2+3u+1) by the chain of Fuentes-Castaneda,
// Knapp, Rodriguez-Henriquez ("Faster hashing to G2", 2011): 10 multiplications, 3 squarings and 3 Frobenius maps beside
// the three exponentiations by u.  m < r and r is prime, so the result is one exactly when the exact value is one; the
// VALUE differs from the canonical Gt, which is why the pairing API keeps final_exponentiation above.
//   lambda = (12u^3+12u^2+6u+1) + (12u^3+6u^2+4u) q + (12u^3+6u^2+6u) q^2 + (12u^3+6u^2+4u-1) q^3
template <bool ACC_LDS = false>
BN_DEVN void final_exponentiation_check(Fp12& r, const Fp12& fin, Fp12& acc) {
  Fp12 f, t, y1, y3, y4, a, y8, y9, y11;
  final_exp_easy(f, fin);
  fp12_pow_u<ACC_LDS>(t, f, acc);
  fp12_conj(t, t);                                  // f^-u
  fp12_cyclotomic_sqr(y1, t);                       // f^-2u
  fp12_cyclotomic_sqr(t, y1);                       // f^-4u
  fp12_mul(y3, t, y1);                              // f^-6u
  fp12_pow_u<ACC_LDS>(t, y3, acc);
  fp12_conj(y4, t);                                 // f^(6u^2)
  fp12_cyclotomic_sqr(t, y4);                       // f^(12u^2)
  fp12_pow_u<ACC_LDS>(a, t, acc);                   // f^(12u^3)
  fp12_conj(y3, y3);                                // f^(6u)
  BN_SET_STEP_PRIORITY(2);   // the short tail: whoever is still here is behind
  fp12_mul(t, a, y4);                               // f^(12u^3+6u^2)
  fp12_mul(y8, t, y3);                              // f^(12u^3+6u^2+6u)
  fp12_mul(y9, y8, y1);                             // f^(12u^3+6u^2+4u)
  fp12_mul(t, y8, y4);                              // f^(12u^3+12u^2+6u)
  fp12_mul(y11, t, f);                              // f^(12u^3+12u^2+6u+1)
  fp12_frob(t, y9, 1);
  fp12_mul(y11, t, y11);
  fp12_frob(t, y8, 2);
  fp12_mul(y11, t, y11);
  fp12_conj(t, f);
  fp12_mul(t, t, y9);                               // f^(12u^3+6u^2+4u-1)
  fp12_frob(a, t, 3);
  fp12_mul(r, a, y11);
}

// ---- the final exponentiation as an ACCUMULATOR MACHINE (the lane-pair kernels, bn254_pair.hip) ---------------------------
// The chains above call ~26 Fq12 routines outside the loop of fp12_pow_u; as real functions each of them saves and restores
// the callee-saved VGPRs it touches (fp12_mul: all 112 -> 224 private-segment dwords per call) and takes its Fq12 arguments
// through memory.  Here the same chains are PROGRAMS (bn254_constants.h: C_FE_CHECK, C_FE_EXACT, written and proved by
// gen_constants.py, which tracks the exponent every value carries) for a machine with ONE Fq12 accumulator — an LDS slot
// in the kernels — and a file of Fq12 slots in the lane's private segment (coalesced dword-interleaved like any private
// array).  The interpreter loop has one switch; every Fq12 routine is inlined at its ONE case, no value lives across
// iterations, and the only memory traffic left is what the chain itself needs: a slot read per multiplication (54 words per
// lane), a slot written per stored intermediate.  Instruction = (opcode, slot or Frobenius power), wave-uniform.
#ifndef BN_FE_PRIO_SHIFT
#define BN_FE_PRIO_SHIFT 2     // the wave-priority cycle advances every 2^shift program steps (x the kernel's own BN_PRIO_SHIFT); 0..5 measured: profiles/r03_x_ab_fe_priority_period.log
#endif
// Round 6: in the translation units that run this machine on the LANE-PAIR layout with the accumulator in LDS (bn254_fe.hip,
// bn254_probe.hip: they define BN_ASM_CSQR_UNIT unless built with -DBN_NO_ASM_CSQR) the CSQR opcode — 189 of the 326 steps of a verify's
// program — is the generated straight-line assembly block of gen_step_asm.py (bn254_csqr_asm.h): explicit VGPR + LDS map, the nine squaring
// leaves inlined with the registers their operands already sit in, no v_mov between operations, nothing through the private segment; the
// same formulas and carry sites as fp12_cyclotomic_sqr_body<170> (so the same int32 limb values and the same bound proof), executed by
// the generator's own four-lane simulator against a big-integer model before it is assembled (tests/test_abi.py).  Same box, alternating:
// final exponentiation 3.98-4.02 -> 3.91-3.92 ms per 65 536 (profiles/r06_d_ab_asm_csqr.log).
#if defined(BN_ASM_CSQR_UNIT) && defined(__HIP_DEVICE_COMPILE__)
#include "bn254_csqr_asm.h"
#define BN_FE_CSQR(acc)                                                                                              \
  do {                                                                                                               \
    const uint32_t lds_addr_ = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)&(acc);                 \
    asm volatile(BN_CSQR_ASM_TEXT : : "v"(lds_addr_) : BN_CSQR_ASM_CLOBBERS);                                        \
  } while (0)
#else
#define BN_FE_CSQR(acc) fp12_cyclotomic_sqr_body<170>(acc, acc)
#endif
// ... and the MUL opcode (51 steps) as the generated block of gen_step_asm.py header_mul (bn254_mul_asm.h; -DBN_NO_ASM_MUL restores the compiled
// routine): same formulas and sites as fp12_mul_body, the eighteen dual products ONE subroutine inside the block (s_call_b64), operands built in
// its input registers by the additions that form them, the slot read with scratch_load from an SGPR address, results at rest in input sets that
// are dead by then (15 sets of nine registers + the leaf's).  Same box, alternating (profiles/r06_m_ab_asm_mul*.log): final exponentiation
// 3.88-4.03 -> 3.82-3.94 ms per 65 536 (-1.6 ... -2.4 %); the chip is power-limited under this mix (1.25-1.28 kW of its 1.4 kW package limit,
// tools/power_sample.sh), and a denser instruction stream pays part of its gain back in clock (Miller kernel +0.4 ... +0.9 % beside it).
#if defined(BN_ASM_CSQR_UNIT) && defined(BN_ASM_MUL) && defined(__HIP_DEVICE_COMPILE__)
#include "bn254_mul_asm.h"
#define BN_FE_MUL(acc, b)                                                                                                                   \
  do {                                                                                                                                      \
    const uint32_t lds_addr_ = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)&(acc);                                        \
    const uint32_t prv_addr_ = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(5))) const void*)&(b));   \
    asm volatile(BN_MUL_ASM_TEXT : : "v"(lds_addr_), "s"(prv_addr_) : BN_MUL_ASM_CLOBBERS);                                                 \
  } while (0)
#else
#define BN_FE_MUL(acc, b) fp12_mul_body(acc, acc, b)
#endif
enum FeOpcode : int { FE_END = 0, FE_LOAD = 1, FE_STORE = 2, FE_CSQR = 3, FE_MUL = 4, FE_CONJ = 5, FE_FROB = 6, FE_INV = 7 };
template <int NSLOTS>
BN_DEV void fe_machine(Fp12& acc, Fp12 (&slot)[NSLOTS], const unsigned char (*prog)[2]) {
#if defined(__HIPCC__)
#pragma clang loop unroll(disable)
#endif
  for (int pc = 0;; ++pc) {
    const int op = prog[pc][0], arg = prog[pc][1];
    if (op == FE_END) break;
    BN_SET_STEP_PRIORITY(pc >> BN_FE_PRIO_SHIFT);
    switch (op) {
      case FE_LOAD: acc = slot[arg]; break;
      case FE_STORE: slot[arg] = acc; break;
      case FE_CSQR: BN_FE_CSQR(acc); break;
      case FE_MUL: BN_FE_MUL(acc, slot[arg]); break;
      case FE_CONJ: fp6_neg(acc.c1, acc.c1); break;        // balanced digits stay balanced: no carry
      case FE_FROB: fp12_frob_body(acc, acc, arg); break;
      default: fp12_inv(acc, acc); break;                   // FE_INV: once per program, a real call
    }
  }
}
// f^(m (q^12-1)/r), one exactly when the pairing product is one: what final_exponentiation_check computes
BN_DEV void fe_machine_check(Fp12& acc) { Fp12 slot[BN_FE_CHECK_SLOTS]; fe_machine(acc, slot, C_FE_CHECK); }
// the canonical Gt value f^((q^12-1)/r): what final_exponentiation computes
BN_DEV void fe_machine_exact(Fp12& acc) { Fp12 slot[BN_FE_EXACT_SLOTS]; fe_machine(acc, slot, C_FE_EXACT); }

}  // namespace bn254
