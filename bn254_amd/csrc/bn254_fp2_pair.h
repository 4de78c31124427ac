// Fq2 in the "pair" layout: the two coefficients of every Fq2 element live in two ADJACENT LANES (even lane:
// real part, odd lane: imaginary part), so a verify is carried by a lane pair.  Every tower level above
// (Fq6, Fq12, twist-point steps, Miller loop, final exponentiation — bn254_field.h / bn254_pairing.h) is
// written against the fp2_* interface only and compiles unchanged for this layout.  Per lane that halves the
// multiplications, the live registers (an Fq12 is 60 words instead of 120) and the LDS footprint of the
// accumulators, which is what lets TWO waves share a SIMD (256 registers each): a lone wave can only issue a
// v_mad_u64_u32 every 8 cycles, two co-resident waves overlap (profiles/r01_issue_mix_microbench.jsonl).
//   product:  re lane  a0*b0 + (-a1)*b1      im lane  a1*b0 + a0*b1     one dual-accumulated Montgomery
//             product each (300 multiply instructions), operands of the partner fetched by DPP quad_perm
//   square:   re lane  (a0+a1)(a0-a1)        im lane  2*a0*a1           one product each
// On the host (tests/hostsim, bound tracker) an element keeps both coefficients and every primitive runs the
// two roles in sequence through the same per-role code.
#pragma once

namespace bn254 {

#if defined(__HIPCC__)
#define BN_FOR_ROLES(k) for (int k = 0; k < 1; ++k)
BN_DEV bool bn_role_im(int) { return (threadIdx.x & 1u) != 0; }
BN_DEV int bn_role_index(int) { return (int)(threadIdx.x & 1u); }
// the same word of the partner lane (quad_perm [1,0,3,2]); both lanes of a pair are always active together
// (measured and rejected: the same exchange as ds_swizzle_b32 on the LDS crossbar — it frees 20 VALU issue slots per
// product in a VALU-bound kernel, but the kernels run 5 % slower with the swizzles placed by the compiler and 12 %
// slower with all of them hoisted to the top of the product routine)
BN_DEV int32_t bn_partner_word(int32_t v) { return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true); }
// the word of the pair's real-part (even) lane / imaginary-part (odd) lane, in both lanes of the pair:
// quad_perm [0,0,2,2] and [1,1,3,3].  A broadcast needs no role select, and as the single user's DPP operand it
// folds into that add / sub / and.
BN_DEV int32_t bn_pair_re_word(int32_t v) { return __builtin_amdgcn_update_dpp(0, v, 0xA0, 0xF, 0xF, true); }
BN_DEV int32_t bn_pair_im_word(int32_t v) { return __builtin_amdgcn_update_dpp(0, v, 0xF5, 0xF, 0xF, true); }
BN_DEV Fp bn_partner(const Fp2& a, int) {
  Fp r;
#pragma unroll
  for (int i = 0; i < BN_LIMBS; ++i) r.v[i] = bn_partner_word(a.c[0].v[i]);
  return r;
}
BN_DEV bool bn_pair_and(bool x) { return x && (bn_partner_word(x ? 1 : 0) != 0); }
#else
#define BN_FOR_ROLES(k) for (int k = 0; k < 2; ++k)
BN_DEV bool bn_role_im(int k) { return k == 1; }
BN_DEV int bn_role_index(int k) { return k; }
BN_DEV Fp bn_partner(const Fp2& a, int k) { return a.c[k ^ 1]; }
BN_DEV bool bn_pair_and(bool x) { return x; }
#endif

// r = Montgomery-reduce(x0*y0 + x1*y1): the per-lane half of an Fq2 product
#if defined(BN_TRACK_BOUNDS) && !defined(__HIPCC__)
static inline void bn_trk_dual(Fp& r, const Fp& x0, const Fp& y0, const Fp& x1, const Fp& y1) {
  double col = bn_col_ab(x0, y0) + bn_col_ab(x1, y1) + BN_COL_EXTRA;
  if (col >= 9223372036854775808.0) {
    if (!bn_bound_soft) fprintf(stderr, "  limbs %g x %g + %g x %g (units of 2^28)\n", bn_absmax(x0) / BN_T, bn_absmax(y0) / BN_T, bn_absmax(x1) / BN_T, bn_absmax(y1) / BN_T);
    bn_bound_fail("pair product column overflow", col);
  }
  auto prod = [](const Fp& x, const Fp& y, double& lo, double& hi) {
    double c[4] = {x.bd.vlo * y.bd.vlo, x.bd.vlo * y.bd.vhi, x.bd.vhi * y.bd.vlo, x.bd.vhi * y.bd.vhi};
    lo = std::fmin(std::fmin(c[0], c[1]), std::fmin(c[2], c[3])) / BN_R_OVER_Q;
    hi = std::fmax(std::fmax(c[0], c[1]), std::fmax(c[2], c[3])) / BN_R_OVER_Q;
  };
  double l0, h0, l1, h1;
  prod(x0, y0, l0, h0); prod(x1, y1, l1, h1);
  if (std::fmax(std::fabs(l0 + l1), std::fabs(h0 + h1)) > BN_VALUE_CAP) {
    if (!bn_bound_soft) fprintf(stderr, "  values %g x %g + %g x %g q\n", bn_vabs(x0), bn_vabs(y0), bn_vabs(x1), bn_vabs(y1));
    bn_bound_fail("pair product value bound", h0 + h1);
  }
  bn_set_tight(r, l0 + l1 - 0.501, h0 + h1 + 0.501);
}
#endif
BN_DEVN BN_VEC10 fp_dual_impl(BN_VEC10 x0, BN_VEC10 y0, BN_VEC10 x1, BN_VEC10 y1) {
  BN_COUNT_MUL();   // a lane pair spends 2 x 1.5 = the 3 algorithmic products of a Karatsuba Fq2 multiplication;
                    // counted per lane as 1 (host emulation: 2 per Fq2 product)
  BN_COUNT_DUAL();
  int32_t a[BN_LIMBS], b[BN_LIMBS], c[BN_LIMBS], d[BN_LIMBS], r[BN_LIMBS];
#pragma unroll
  for (int i = 0; i < BN_LIMBS; ++i) { a[i] = x0[i]; b[i] = y0[i]; c[i] = x1[i]; d[i] = y1[i]; }
  fp_dual_mul_reduce(r, a, b, c, d);
  BN_VEC10 z;
#pragma unroll
  for (int i = 0; i < BN_LIMBS; ++i) z[i] = r[i];
  return z;
}
BN_DEV Fp fp_dual(const Fp& x0, const Fp& y0, const Fp& x1, const Fp& y1) {
  BN_VEC10 a, b, c, d;
#pragma unroll
  for (int i = 0; i < BN_LIMBS; ++i) { a[i] = x0.v[i]; b[i] = y0.v[i]; c[i] = x1.v[i]; d[i] = y1.v[i]; }
  BN_VEC10 z = fp_dual_impl(a, b, c, d);
  Fp r;
#pragma unroll
  for (int i = 0; i < BN_LIMBS; ++i) r.v[i] = z[i];
  BN_TRK(bn_trk_dual(r, x0, y0, x1, y1));
  return r;
}

BN_DEV Fp2 fp2_zero() { Fp2 r; BN_FOR_ROLES(k) r.c[k] = fp_zero(); return r; }
BN_DEV Fp2 fp2_one() { Fp2 r; BN_FOR_ROLES(k) r.c[k] = fp_select(bn_role_im(k), fp_zero(), fp_one()); return r; }
BN_DEV Fp2 fp2_load_const(const int32_t (*c)[BN_LIMBS]) { Fp2 r; BN_FOR_ROLES(k) r.c[k] = fp_load_const(c[bn_role_index(k)]); return r; }
BN_DEV Fp2 fp2_add(const Fp2& a, const Fp2& b) { Fp2 r; BN_FOR_ROLES(k) r.c[k] = fp_add(a.c[k], b.c[k]); return r; }
BN_DEV Fp2 fp2_sub(const Fp2& a, const Fp2& b) { Fp2 r; BN_FOR_ROLES(k) r.c[k] = fp_sub(a.c[k], b.c[k]); return r; }
BN_DEV Fp2 fp2_neg(const Fp2& a) { Fp2 r; BN_FOR_ROLES(k) r.c[k] = fp_neg(a.c[k]); return r; }
BN_DEV Fp2 fp2_dbl(const Fp2& a) { return fp2_add(a, a); }
BN_DEV Fp2 fp2_conj(const Fp2& a) { Fp2 r; BN_FOR_ROLES(k) r.c[k] = fp_select(bn_role_im(k), fp_neg(a.c[k]), a.c[k]); return r; }
BN_DEV Fp2 fp2_norm(const Fp2& a) { Fp2 r; BN_FOR_ROLES(k) r.c[k] = fp_norm(a.c[k]); return r; }
BN_DEV Fp2 fp2_reduce_weak(const Fp2& a) { Fp2 r; BN_FOR_ROLES(k) r.c[k] = fp_reduce_weak(a.c[k]); return r; }
BN_DEV Fp2 fp2_lin2_reduce(const Fp2& x, int32_t cx, const Fp2& y, int32_t cy) {
  Fp2 r; BN_FOR_ROLES(k) r.c[k] = fp_lin2_reduce(x.c[k], cx, y.c[k], cy); return r;
}
BN_DEV Fp2 fp2_lin4_reduce(const Fp2& a, int32_t ca, const Fp2& b, int32_t cb, const Fp2& c, int32_t cc, const Fp2& d, int32_t cd) {
  Fp2 r; BN_FOR_ROLES(k) r.c[k] = fp_lin4_reduce(a.c[k], ca, b.c[k], cb, c.c[k], cc, d.c[k], cd); return r;
}
BN_DEV bool fp2_is_zero(const Fp2& a) { bool z = true; BN_FOR_ROLES(k) z = fp_is_zero(a.c[k]) && z; return bn_pair_and(z); }
BN_DEV bool fp2_eq(const Fp2& a, const Fp2& b) { return fp2_is_zero(fp2_sub(a, b)); }
BN_DEV Fp2 fp2_select(bool c, const Fp2& a, const Fp2& b) { Fp2 r; BN_FOR_ROLES(k) r.c[k] = fp_select(c, a.c[k], b.c[k]); return r; }
BN_DEV Fp2 fp2_select_pos(bool c, const Fp2& a, const Fp2& b) { Fp2 r; BN_FOR_ROLES(k) r.c[k] = fp_select_pos(c, a.c[k], b.c[k]); return r; }   // c = a constant of the lane's position (bn254_field.h)

#if defined(__HIPCC__)
// Device form of fp2_mul: the callee fetches the partner's operands itself (DPP) and picks its role's operand
// pairing, so a call passes 20 words in registers (a 40-word call spills 8 argument words to the stack) and the
// exchange / select code exists once instead of at every call site.
BN_DEVN BN_VEC10 fp_pair_mul_impl(BN_VEC10 a, BN_VEC10 b) {
  // own * b0 + partner * (+-b1):   re lane  a0*b0 + a1*(-b1)     im lane  a1*b0 + a0*b1
  // b0 and b1 are broadcasts within the pair
  // -b1 in the real-part lanes: (p ^ -1) + 1, the fetch folded into the XOR (v_xor_b32_dpp) and the +1 / +0 a plain add —
  // two 32-bit VALU operations where a multiplication by the lane's -1 / +1 cost a v_mul_lo_u32 (issue cost of a multiply-add)
  const int32_t one = 1 - (int32_t)(threadIdx.x & 1u), mask = -one;
  int32_t ao[BN_LIMBS], ap[BN_LIMBS], x[BN_LIMBS], y[BN_LIMBS], r[BN_LIMBS];
#pragma unroll
  for (int i = 0; i < BN_LIMBS; ++i) {
    ao[i] = a[i];
    ap[i] = bn_partner_word(a[i]);
    x[i] = bn_pair_re_word(b[i]);
    y[i] = (bn_pair_im_word(b[i]) ^ mask) + one;
  }
  BN_MONT_DUAL_BODY(ao, x, ap, y, r);
#if defined(BN_PROBE_EXTRA_ADDS) || defined(BN_PROBE_EXTRA_MACS)
  // Measurement builds only (tools/build_variant.sh … -DBN_PROBE_EXTRA_ADDS=k / -DBN_PROBE_EXTRA_MACS=k): 4 k extra 32-bit additions, or 4 k extra
  // multiply-adds, per dual product on registers of their own (four independent chains each) — what ONE more instruction of either class costs
  // the kernels in time at the clock the part grants the mix (profiles/r06_z_marginal_instruction_cost.log).
#define BN_REP1(s) s
#define BN_REP2(s) s s
#define BN_REP3(s) s s s
#define BN_REP4(s) s s s s
#define BN_REP6(s) s s s s s s
#define BN_REP8(s) s s s s s s s s
#define BN_REP12(s) BN_REP6(s) BN_REP6(s)
#define BN_REPX(k, s) BN_REP##k(s)
#define BN_REP(k, s) BN_REPX(k, s)
  {
#if defined(BN_PROBE_EXTRA_ADDS)
    int32_t d0 = ao[0], d1 = ao[1], d2 = ao[2], d3 = ao[3];
    asm volatile(BN_REP(BN_PROBE_EXTRA_ADDS, "v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n")
                 : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(x[0]));
    asm volatile("" ::"v"(d0), "v"(d1), "v"(d2), "v"(d3));
#endif
#if defined(BN_PROBE_EXTRA_MACS)
    int64_t e0 = ao[0], e1 = ao[1], e2 = ao[2], e3 = ao[3];
    uint64_t k0, k1, k2, k3;
    asm volatile(BN_REP(BN_PROBE_EXTRA_MACS, "v_mad_i64_i32 %0, %4, %8, %9, %0\n v_mad_i64_i32 %1, %5, %8, %9, %1\n v_mad_i64_i32 %2, %6, %8, %9, %2\n v_mad_i64_i32 %3, %7, %8, %9, %3\n")
                 : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3), "=&s"(k0), "=&s"(k1), "=&s"(k2), "=&s"(k3) : "v"(x[0]), "v"(y[0]));
    asm volatile("" ::"v"(e0), "v"(e1), "v"(e2), "v"(e3));
#endif
  }
#endif
  BN_VEC10 z;
#pragma unroll
  for (int i = 0; i < BN_LIMBS; ++i) z[i] = r[i];
  return z;
}
#endif
BN_DEV Fp2 fp2_mul(const Fp2& a, const Fp2& b) {   // outputs are tight
  Fp2 r;
#if defined(__HIPCC__)
  BN_VEC10 x, y;
#pragma unroll
  for (int i = 0; i < BN_LIMBS; ++i) { x[i] = a.c[0].v[i]; y[i] = b.c[0].v[i]; }
  BN_VEC10 z = fp_pair_mul_impl(x, y);
#pragma unroll
  for (int i = 0; i < BN_LIMBS; ++i) r.c[0].v[i] = z[i];
  return r;
#endif
  BN_FOR_ROLES(k) {
    const bool im = bn_role_im(k);
    const Fp ap = bn_partner(a, k), bp = bn_partner(b, k);
    // re: a0*b0 + a1*(-b1)   im: a1*b0 + a0*b1   (own = a.c[k], partner = ap)
    r.c[k] = fp_dual(a.c[k], fp_select(im, bp, b.c[k]), ap, fp_select(im, b.c[k], fp_neg(bp)));
  }
  return r;
}
// ---- sum of N Fq2 products with ONE reduction per coefficient (the Fq6-level lazy reduction, bn254_field.h: fp6_mul_lazy) --------------
// r = a[0] b[0] + ... + a[N-1] b[N-1]: per lane 2N limb products share the columns (BN_MONT_MULTI_BODY).  Operand contract (checked by the
// bound tracker): sum over the 2N limb products of A_t B_t <= 14 units — tight operands, one side of each product may be a lazy sum of two.
#if defined(BN_TRACK_BOUNDS) && !defined(__HIPCC__)
static inline void bn_trk_multi(Fp& r, int n, const Fp* x, const Fp* y) {
  double col = BN_COL_EXTRA, lo = 0, hi = 0;
  for (int t = 0; t < n; ++t) {
    col += bn_col_ab(x[t], y[t]);
    double c[4] = {x[t].bd.vlo * y[t].bd.vlo, x[t].bd.vlo * y[t].bd.vhi, x[t].bd.vhi * y[t].bd.vlo, x[t].bd.vhi * y[t].bd.vhi};
    lo += std::fmin(std::fmin(c[0], c[1]), std::fmin(c[2], c[3])) / BN_R_OVER_Q;
    hi += std::fmax(std::fmax(c[0], c[1]), std::fmax(c[2], c[3])) / BN_R_OVER_Q;
  }
  if (col >= 9223372036854775808.0) bn_bound_fail("multi product column overflow", col);
  if (std::fmax(std::fabs(lo), std::fabs(hi)) > BN_VALUE_CAP) bn_bound_fail("multi product value bound", hi);
  bn_set_tight(r, lo - 0.501, hi + 0.501);
}
#endif
template <int N> BN_DEV Fp2 fp2_mul_sum(const Fp2* const (&a)[N], const Fp2* const (&b)[N]) {
  Fp2 r;
#if defined(__HIPCC__)
  // own * bcast_re(b) + partner * (+-bcast_im(b)) for every product: the exchanges of fp_pair_mul_impl, once per operand
  const int32_t one = 1 - (int32_t)(threadIdx.x & 1u), mask = -one;
  int32_t x[2 * N][BN_LIMBS], y[2 * N][BN_LIMBS], z[BN_LIMBS];
#pragma unroll
  for (int t = 0; t < N; ++t)
#pragma unroll
    for (int i = 0; i < BN_LIMBS; ++i) {
      x[2 * t][i] = a[t]->c[0].v[i];
      x[2 * t + 1][i] = bn_partner_word(a[t]->c[0].v[i]);
      y[2 * t][i] = bn_pair_re_word(b[t]->c[0].v[i]);
      y[2 * t + 1][i] = (bn_pair_im_word(b[t]->c[0].v[i]) ^ mask) + one;
    }
  BN_MONT_MULTI_BODY(2 * N, x, y, z);
#pragma unroll
  for (int i = 0; i < BN_LIMBS; ++i) r.c[0].v[i] = z[i];
#else
  BN_FOR_ROLES(k) {
    const bool im = bn_role_im(k);
    Fp xs[2 * N], ys[2 * N];
    int32_t x[2 * N][BN_LIMBS], y[2 * N][BN_LIMBS], z[BN_LIMBS];
    for (int t = 0; t < N; ++t) {
      const Fp ap = bn_partner(*a[t], k), bp = bn_partner(*b[t], k);
      xs[2 * t] = a[t]->c[k]; ys[2 * t] = fp_select(im, bp, b[t]->c[k]);
      xs[2 * t + 1] = ap; ys[2 * t + 1] = fp_select(im, b[t]->c[k], fp_neg(bp));
      BN_COUNT_MUL(); BN_COUNT_DUAL();
    }
    for (int t = 0; t < 2 * N; ++t)
      for (int i = 0; i < BN_LIMBS; ++i) { x[t][i] = xs[t].v[i]; y[t][i] = ys[t].v[i]; }
    BN_MONT_MULTI_BODY(2 * N, x, y, z);
    for (int i = 0; i < BN_LIMBS; ++i) r.c[k].v[i] = z[i];
    BN_TRK(bn_trk_multi(r.c[k], 2 * N, xs, ys));
  }
#endif
  return r;
}

#if defined(__HIPCC__)
// Device form of fp2_sqr, same idea: re (a0 + a1)(a0 - a1), im 2 * (a1 * a0) — one product per lane
// A leaf: the product body is inlined (no call frame, no saved return address in scratch).
BN_DEVN BN_VEC10 fp_pair_sqr_impl(BN_VEC10 a) {
#include "bn254_pair_sqr_body.inc"
}
#endif
BN_DEV Fp2 fp2_sqr(const Fp2& a) {
  Fp2 r;
#if defined(__HIPCC__)
  BN_VEC10 x;
#pragma unroll
  for (int i = 0; i < BN_LIMBS; ++i) x[i] = a.c[0].v[i];
  BN_VEC10 z = fp_pair_sqr_impl(x);
#pragma unroll
  for (int i = 0; i < BN_LIMBS; ++i) r.c[0].v[i] = z[i];
  return r;
#endif
  BN_FOR_ROLES(k) {
    const bool im = bn_role_im(k);
    const Fp ap = bn_partner(a, k);
    // re: (a0 + a1)(a0 - a1)   im: (2 a1) * a0
    r.c[k] = fp_mul(fp_select(im, fp_dbl(a.c[k]), fp_add(a.c[k], ap)), fp_select(im, ap, fp_sub(a.c[k], ap)));
  }
  return r;
}
BN_DEV Fp2 fp2_mul_fp(const Fp2& a, const Fp& s) { Fp2 r; BN_FOR_ROLES(k) r.c[k] = fp_mul(a.c[k], s); return r; }
BN_DEV Fp2 fp2_mul_xi(const Fp2& a) {              // (9 + i) * a; 8 * own by fp_mul8_spread: limbs <= 2^28 + |own_i| + |partner_i|
  Fp2 r;
#if defined(__HIPCC__)
  // re: 9 a0 - a1   im: 9 a1 + a0.   -p = (p ^ -1) + 1: the partner fetch folds into the XOR with this lane's mask
  // (-1 in a real-part lane, 0 in an imaginary-part lane).  8 * own crosses the limb boundary (fp_mul8_spread: own =
  // h * 2^26 + l, 8 l stays in limb i, h moves up) so that no limb grows eight-fold; the +1 of the negation rides in
  // the rounding constant of h:  h' = (own + 2^25 + one * 2^26) >> 26 = h + one.       6 instructions per limb.
  const int32_t one = 1 - (int32_t)(threadIdx.x & 1u), mask = -one;
  const uint32_t hround = (1u << (BN_SPREAD - 1)) + ((uint32_t)one << BN_SPREAD);
  int32_t h = one;
#pragma unroll
  for (int i = 0; i < BN_LIMBS; ++i) {
    const int32_t own = a.c[0].v[i];
    const int32_t t = (bn_partner_word(own) ^ mask) + own + h;        // own -+ partner, + what limb i-1 carried up
    if (i < BN_LIMBS - 1) {
      r.c[0].v[i] = bn_spread_lo(own) * 8 + t;
      h = (int32_t)((uint32_t)own + hround) >> BN_SPREAD;
    } else {
      r.c[0].v[i] = own * 8 + t;
    }
  }
  return r;
#endif
  BN_FOR_ROLES(k) {
    const Fp ap = bn_partner(a, k);
    // re: 9 a0 - a1   im: 9 a1 + a0
    r.c[k] = fp_add(fp_add(fp_mul8_spread(a.c[k]), a.c[k]), fp_select(bn_role_im(k), ap, fp_neg(ap)));
  }
  return r;
}
BN_DEV Fp2 fp2_mul_xi_n(const Fp2& a) { return fp2_mul_xi(fp2_norm(a)); }
BN_DEV Fp2 fp2_mul8(const Fp2& a) { Fp2 r; BN_FOR_ROLES(k) r.c[k] = fp_mul8_spread(a.c[k]); return r; }
BN_DEV Fp2 fp2_inv(const Fp2& a) {
  Fp2 t, r;
  BN_FOR_ROLES(k) t.c[k] = fp_sqr(a.c[k]);
  BN_FOR_ROLES(k) {
    Fp n = fp_inv(fp_add(t.c[k], bn_partner(t, k)));   // both lanes invert the norm (the one duplicated step)
    Fp m = fp_mul(a.c[k], n);
    r.c[k] = fp_select(bn_role_im(k), fp_neg(m), m);
  }
  return r;
}

// i * a = (-a1, a0); the element re + im i (every lane holds both integers and keeps its role's); u512 order
BN_DEV Fp2 fp2_mul_i(const Fp2& a) {
  Fp2 r;
  BN_FOR_ROLES(k) { const Fp ap = bn_partner(a, k); r.c[k] = fp_select(bn_role_im(k), ap, fp_neg(ap)); }
  return r;
}
BN_DEV Fp2 fp2_make(const Fp& re, const Fp& im) { Fp2 r; BN_FOR_ROLES(k) r.c[k] = fp_select(bn_role_im(k), im, re); return r; }
// canonical "u512(c) = c.im * q + c.re" order == lexicographic (im, re): each role compares its coefficient, the
// imaginary role decides unless its coefficients are equal
BN_DEV bool fp2_u512_greater(const Fp2& a, const Fp2& b) {
  bool eq[BN_PAIR_ROLES], gt[BN_PAIR_ROLES];
  BN_FOR_ROLES(k) {
    U256 x = fp_to_u256(a.c[k]), y = fp_to_u256(b.c[k]);
    bool e = true;
    for (int i = 0; i < 8; ++i) e = e && x.w[i] == y.w[i];
    eq[k] = e; gt[k] = !e && u256_geq(x.w, y.w);
  }
#if defined(__HIPCC__)
  const bool im = (threadIdx.x & 1u) != 0;
  const bool p_eq = bn_partner_word(eq[0] ? 1 : 0) != 0, p_gt = bn_partner_word(gt[0] ? 1 : 0) != 0;
  const bool im_eq = im ? eq[0] : p_eq, im_gt = im ? gt[0] : p_gt, re_gt = im ? p_gt : gt[0];
  return im_eq ? re_gt : im_gt;
#else
  return eq[1] ? gt[0] : gt[1];
#endif
}

}  // namespace bn254
