// BN254 base field Fq and the tower Fq2 / Fq6 / Fq12 for gfx950 lanes.
//
// One field element per lane: 10 signed limbs of 27 bits in VGPRs, Montgomery form (R = 2^270).
// The hot primitive is a product-scanning Montgomery product built from v_mad_i64_i32 /
// v_mad_u64_u32 (measured on MI355X: ~5 cycles per wave64 instruction per SIMD — the same issue
// cost as v_mul_lo/hi_u32, v_add_co/v_addc or v_fma_f64, so what matters is the instruction COUNT:
// unsaturated limbs let a 64-bit column accumulator absorb every carry; no MFMA — integer work).
//
// Tower: Fq2 = Fq[i]/(i^2+1), Fq6 = Fq2[v]/(v^3 - xi), Fq12 = Fq6[w]/(w^2 - v), xi = 9+i
// (SURVEY.md Appendix A.1).  This replaces, for the hot path only, the arithmetic the
// reference gets from `bn::{Fq,Fq2,Fq12,...}` (zeropool-bn 0.5.11, /root/reference/Cargo.toml:24;
// call sites /root/reference/src/ecdsa.rs:57, /root/reference/src/utils.rs:111-125).
//
// The code is plain C++ (no HIP intrinsics) so that tests can also compile this exact source
// for the host and check the algorithm against the oracle without a GPU (tests/hostsim/).
// Inlining policy: Fq mul/sqr and everything from Fq6 upwards are real (non-inlined) device
// functions — a fully inlined Fq12 tower is hundreds of KB of ISA against a 64 KB I-cache.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define BN_DEV __device__ __forceinline__
#define BN_DEVN __device__ __noinline__
#define BN_CONST __device__ __constant__ const
// BN_DEVH: the two Fq12 routines of the Miller loop body (fp12_sqr, fp12_mul_line2).  As real functions each call
// saves and restores the ~87 callee-saved VGPRs it needs for values that live across its product calls
// (154 calls x 174 scratch dwords per Miller loop = 14 GB of private-segment traffic per 65 536 verifies);
// a translation unit that defines BN_INLINE_FP12_HOT inlines them into the loop, whose own prologue then saves
// those registers once.
#if defined(BN_INLINE_FP12_HOT)
#define BN_DEVH BN_DEV
#else
#define BN_DEVH BN_DEVN
#endif
#else
#define BN_DEV static inline __attribute__((always_inline))
#define BN_DEVN static __attribute__((noinline))
#define BN_CONST static const
#define BN_DEVH BN_DEVN
#endif

#include "bn254_constants.h"

// host-only instrumentation (tests/hostsim): exact count of Montgomery products per kernel stage,
// the algorithmic-work figure behind bench.py's roofline (1 product = 136 MAC32)
#if defined(BN_COUNT_FP_MUL) && !defined(__HIPCC__)
extern "C" unsigned long long bn_fp_mul_counter;
extern "C" unsigned long long bn_fp_dual_counter;   // of those, dual-accumulated products (pair layout: x0*y0 + x1*y1, one reduction)
#define BN_COUNT_MUL() (++bn_fp_mul_counter)
#define BN_COUNT_DUAL() (++bn_fp_dual_counter)
#else
#define BN_COUNT_MUL()
#define BN_COUNT_DUAL()
#endif

// Hook for kernels that keep two waves on a SIMD (bn254_pair.hip): wave priority cycling 3,2,1,0 with the step
// count of the long loops.  The issue arbiter otherwise favours the older wave of a SIMD and the pair drifts 3 ms
// apart; with the cycle, whichever wave falls a few steps behind is in a higher-priority part of the cycle and
// catches up.  A no-op everywhere else.
#ifndef BN_SET_STEP_PRIORITY
#define BN_SET_STEP_PRIORITY(step) do { } while (0)
#endif

namespace bn254 {

// ------------------------------------------------------------------------------------------
// Fq: unsaturated signed limbs.
//
// An element is 10 int32 limbs of nominally 27 bits, value = sum v[i] * 2^(27 i), Montgomery form
// with R = 2^270.  Limbs are SIGNED and may temporarily exceed 27 bits:
//   * add / sub / neg / dbl are 10 plain v_add/v_sub (no carry chain, no modular correction);
//   * mul is a product-scanning Montgomery product: each of the 19 columns accumulates its
//     a_i*b_j and m_i*q_j terms in one 64-bit register with v_mad_i64_i32 / v_mad_u64_u32 — no
//     carry handling at all — followed by one shift per column (~280 instructions instead of
//     ~575 for saturated 8x32-bit CIOS);
//   * norm() propagates carries so that limbs 0..8 are back in [0, 2^27) (the top limb absorbs);
//   * canon() produces the unique representative in [0, q) (only for comparisons and output).
// Safety conditions (machine-checked by the bound-tracking host build, tests/test_bounds.py):
//   mul(a,b): 10 * max|a_i| * max|b_j| + 10 * 2^54 + 2^37 < 2^63   and   |a||b| < q*R/2^6
//   every limb always fits int32.
// mul outputs are "tight": limbs 0..8 in [0, 2^27), |value| < q * (1 + |a||b|/(qR)).
// ------------------------------------------------------------------------------------------
#define BN_QL_ARRAY {BN_QL0, BN_QL1, BN_QL2, BN_QL3, BN_QL4, BN_QL5, BN_QL6, BN_QL7, BN_QL8, BN_QL9}

#if defined(BN_TRACK_BOUNDS) && !defined(__HIPCC__)
// Host-only interval bookkeeping: [lo,hi] bounds every limb, [vlo,vhi] bounds value/q.  The
// bounds depend only on the sequence of operations, not on the data, so one pass of the test
// vectors through this build proves the safety conditions for every formula on the path.
}  // namespace bn254
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <execinfo.h>
namespace bn254 {
struct FpBounds { double lo, hi, top, vlo, vhi; };   // limbs 0..8 in [lo,hi], |limb 9| <= top, value/q in [vlo,vhi]
#define BN_BOUNDS_MEMBER FpBounds bd;
#define BN_T 134217728.0 /* 2^27 */
static inline void bn_bound_fail(const char* what, double x) {
  fprintf(stderr, "BOUND VIOLATION: %s (%g)\n", what, x);
  void* bt[24];
  int n = backtrace(bt, 24);
  backtrace_symbols_fd(bt, n, 2);
  abort();
}
#else
#define BN_BOUNDS_MEMBER
#endif

struct Fp { int32_t v[BN_LIMBS]; BN_BOUNDS_MEMBER };
#if defined(BN_SPLIT_FP2)
// "pair" layout (bn254_fp2_pair.h): on the device a lane holds ONE coefficient of every Fq2 element and the
// adjacent lane the other; the host build keeps both and runs the two roles one after the other.
#if defined(__HIPCC__)
#define BN_PAIR_ROLES 1
#else
#define BN_PAIR_ROLES 2
#endif
struct Fp2 { Fp c[BN_PAIR_ROLES]; };
#else
struct Fp2 { Fp c0, c1; };
#endif
struct Fp6 { Fp2 c0, c1, c2; };
struct Fp12 { Fp6 c0, c1; };
struct U256 { uint32_t w[8]; };   // plain 256-bit integer

#if defined(BN_TRACK_BOUNDS) && !defined(__HIPCC__)
static inline double bn_absmax(const Fp& a) { return std::fmax(std::fmax(std::fabs(a.bd.lo), std::fabs(a.bd.hi)), a.bd.top); }
static inline double bn_vabs(const Fp& a) { return std::fmax(std::fabs(a.bd.vlo), std::fabs(a.bd.vhi)); }
// "tight": limbs 0..8 in [0, 2^27); the top limb is then exactly floor(value / 2^243), |top| <= |value/q| * 1549 + 1
static inline void bn_set_tight(Fp& r, double vlo, double vhi) {
  r.bd.lo = 0; r.bd.hi = BN_T; r.bd.vlo = vlo; r.bd.vhi = vhi;
  r.bd.top = std::fmax(std::fabs(vlo), std::fabs(vhi)) * 1549.0 + 2.0;
}
static inline void bn_chk_i32(const Fp& r) { if (bn_absmax(r) >= 2147483648.0) bn_bound_fail("limb exceeds int32", bn_absmax(r)); }
#define BN_TRK(stmt) do { stmt; } while (0)
#else
#define BN_TRK(stmt) do { } while (0)
#endif

BN_DEV Fp fp_load_const(const int32_t* c) {
  Fp r;
#pragma unroll
  for (int i = 0; i < BN_LIMBS; ++i) r.v[i] = c[i];
  BN_TRK(bn_set_tight(r, 0, 1));
  return r;
}
BN_DEV Fp fp_zero() {
  Fp r;
#pragma unroll
  for (int i = 0; i < BN_LIMBS; ++i) r.v[i] = 0;
  BN_TRK(r.bd = FpBounds({0, 0, 0, 0, 0}));
  return r;
}
BN_DEV Fp fp_one() { return fp_load_const(C_ONE); }
// r = c ? a : b
BN_DEV Fp fp_select(bool c, const Fp& a, const Fp& b) {
  Fp r;
#pragma unroll
  for (int i = 0; i < BN_LIMBS; ++i) r.v[i] = c ? a.v[i] : b.v[i];
  BN_TRK(r.bd = FpBounds({std::fmin(a.bd.lo, b.bd.lo), std::fmax(a.bd.hi, b.bd.hi), std::fmax(a.bd.top, b.bd.top), std::fmin(a.bd.vlo, b.bd.vlo), std::fmax(a.bd.vhi, b.bd.vhi)}));
  return r;
}
// a >= b as 256-bit integers
BN_DEV bool u256_geq(const uint32_t* a, const uint32_t* b) {
  uint32_t bw = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    uint64_t x = (uint64_t)a[i] - b[i] - bw;
    bw = (uint32_t)(x >> 63);
  }
  return bw == 0;
}
BN_DEV Fp fp_add(const Fp& a, const Fp& b) {
  Fp r;
#pragma unroll
  for (int i = 0; i < BN_LIMBS; ++i) r.v[i] = a.v[i] + b.v[i];
  BN_TRK(r.bd = FpBounds({a.bd.lo + b.bd.lo, a.bd.hi + b.bd.hi, a.bd.top + b.bd.top, a.bd.vlo + b.bd.vlo, a.bd.vhi + b.bd.vhi}); bn_chk_i32(r));
  return r;
}
BN_DEV Fp fp_sub(const Fp& a, const Fp& b) {
  Fp r;
#pragma unroll
  for (int i = 0; i < BN_LIMBS; ++i) r.v[i] = a.v[i] - b.v[i];
  BN_TRK(r.bd = FpBounds({a.bd.lo - b.bd.hi, a.bd.hi - b.bd.lo, a.bd.top + b.bd.top, a.bd.vlo - b.bd.vhi, a.bd.vhi - b.bd.vlo}); bn_chk_i32(r));
  return r;
}
BN_DEV Fp fp_neg(const Fp& a) {
  Fp r;
#pragma unroll
  for (int i = 0; i < BN_LIMBS; ++i) r.v[i] = -a.v[i];
  BN_TRK(r.bd = FpBounds({-a.bd.hi, -a.bd.lo, a.bd.top, -a.bd.vhi, -a.bd.vlo}));
  return r;
}
BN_DEV Fp fp_dbl(const Fp& a) { return fp_add(a, a); }
// carry propagation: limbs 0..8 -> [0, 2^27), the top limb absorbs; the value is unchanged
BN_DEV Fp fp_norm(const Fp& a) {
  Fp r;
  int32_t c = 0;
#pragma unroll
  for (int i = 0; i < BN_LIMBS - 1; ++i) {
    int32_t x = a.v[i] + c;
    r.v[i] = (int32_t)((uint32_t)x & BN_MASK);
    c = x >> BN_W;
  }
  r.v[BN_LIMBS - 1] = a.v[BN_LIMBS - 1] + c;
  BN_TRK(if (bn_absmax(a) + 64 >= 2147483648.0) bn_bound_fail("norm input", bn_absmax(a)); bn_set_tight(r, a.bd.vlo, a.bd.vhi));
  return r;
}

// Weak modular reduction of a tight element: subtract k*q with k ~ value/q estimated from the top
// limb (k = floor(top * 21 / 2^15), 21/2^15 = 0.9924 * 2^243/q).  The residue is unchanged and the
// value drops to within [-0.0076|V| - 0.01, 0.0076|V| + 1.02] * q.  Used where an output is LINEAR in
// an input (cyclotomic squaring) so that values cannot build up across iterations.
BN_DEV Fp fp_reduce_weak(const Fp& a) {
  const int32_t q[BN_LIMBS] = BN_QL_ARRAY;
  int32_t k = (a.v[BN_LIMBS - 1] * 21) >> 15;
  int32_t carry = 0;
  Fp r;
#pragma unroll
  for (int i = 0; i < BN_LIMBS; ++i) {
    int64_t acc = (int64_t)(a.v[i] + carry) - (int64_t)k * q[i];
    if (i < BN_LIMBS - 1) {
      r.v[i] = (int32_t)((uint32_t)acc & BN_MASK);
      carry = (int32_t)(acc >> BN_W);
    } else {
      r.v[i] = (int32_t)acc;
    }
  }
  BN_TRK(if (a.bd.lo < 0 || a.bd.hi > BN_T || a.bd.top * 21.0 >= 2147483648.0) bn_bound_fail("reduce_weak needs a tight input", a.bd.hi);
         double va_ = bn_vabs(a); bn_set_tight(r, -0.0076 * va_ - 0.01, 0.0076 * va_ + 1.02));
  return r;
}

// Montgomery product a*b*R^-1 (mod q), product scanning.  Columns are accumulated in a signed
// 64-bit register; m_k = (column * -q^-1) mod 2^27 makes each column divisible by 2^27.
#if defined(__HIPCC__)
typedef int32_t bn_i32x16 __attribute__((vector_size(64)));   // 10 limbs travel in VGPRs across the call
#define BN_LIMB_VEC bn_i32x16
#else
struct bn_limbvec { int32_t e[16]; int32_t& operator[](int i) { return e[i]; } const int32_t& operator[](int i) const { return e[i]; } };
#define BN_LIMB_VEC bn_limbvec
#endif

// acc >>= 27 (arithmetic): one v_ashrrev_i64 — measured at the issue cost of a 32-bit VALU op on gfx950
// (profiles/r01_issue_mix_microbench.jsonl), cheaper than an alignbit + ashr pair
#define BN_COLUMN_SHIFT(acc) do { (acc) >>= BN_W; } while (0)

// acc += x * y: one v_mad_i64_i32 per limb product, left to the compiler.  Measured and rejected: spelling the
// instruction out with the (unused) carry-out alternating between two SGPR pairs.  In a synthetic stream that
// lifts a lone wave from 4.0 to 2.2 ns per multiply-add (profiles/r01_mad_latency_microbench.jsonl), but in the
// real routines, where every multiply-add reads four fresh VGPR words, it changes nothing (793 vs 774 ns per
// product) and the asm statements cost scheduling freedom.
#define BN_MAC(acc, x, y) do { (acc) += (int64_t)(x) * (y); } while (0)

// The body is a macro so that both users contain the loops themselves: the same code reached through an inlined
// helper compiles to 18 more instructions (a second accumulator chain merged by a v_lshl_add_u64 per column).
#define BN_MONT_PRODUCT_BODY(a, b, r)                                                    \
  do {                                                                                   \
    const int32_t q_[BN_LIMBS] = BN_QL_ARRAY;                                            \
    int64_t acc_ = 0;                                                                    \
    int32_t m_[BN_LIMBS];                                                                \
    _Pragma("unroll") for (int k_ = 0; k_ < 2 * BN_LIMBS - 1; ++k_) {                    \
      _Pragma("unroll") for (int i_ = 0; i_ < BN_LIMBS; ++i_) {                          \
        int j_ = k_ - i_;                                                                \
        if (j_ < 0 || j_ >= BN_LIMBS) continue;                                          \
        BN_MAC(acc_, (a)[i_], (b)[j_]);                                                  \
      }                                                                                  \
      _Pragma("unroll") for (int i_ = 0; i_ < BN_LIMBS; ++i_) {                          \
        int j_ = k_ - i_;                                                                \
        if (j_ < 0 || j_ >= BN_LIMBS) continue;                                          \
        if (k_ < BN_LIMBS && i_ >= k_) continue; /* m_k itself is added below */         \
        BN_MAC(acc_, m_[i_], q_[j_]);                                                    \
      }                                                                                  \
      if (k_ < BN_LIMBS) {                                                               \
        m_[k_] = (int32_t)(((uint32_t)acc_ * BN_N0) & BN_MASK);                          \
        BN_MAC(acc_, m_[k_], q_[0]);                                                     \
      } else {                                                                           \
        (r)[k_ - BN_LIMBS] = (int32_t)((uint32_t)acc_ & BN_MASK);                        \
      }                                                                                  \
      BN_COLUMN_SHIFT(acc_);                                                             \
    }                                                                                    \
    (r)[BN_LIMBS - 1] = (int32_t)acc_;                                                   \
  } while (0)
BN_DEVN BN_LIMB_VEC fp_mul_impl(BN_LIMB_VEC a, BN_LIMB_VEC b) {
  BN_COUNT_MUL();
  BN_LIMB_VEC r;
  BN_MONT_PRODUCT_BODY(a, b, r);
  return r;
}
// a^2: 55 limb products instead of 100 (cross terms through the doubled operand)
BN_DEVN BN_LIMB_VEC fp_sqr_impl(BN_LIMB_VEC a) {
  BN_COUNT_MUL();
  const int32_t q[BN_LIMBS] = BN_QL_ARRAY;
  int32_t a2[BN_LIMBS];
#pragma unroll
  for (int i = 0; i < BN_LIMBS; ++i) a2[i] = a[i] + a[i];
  int64_t acc = 0;
  int32_t m[BN_LIMBS];
  BN_LIMB_VEC r;
#pragma unroll
  for (int k = 0; k < 2 * BN_LIMBS - 1; ++k) {
#pragma unroll
    for (int i = 0; i < BN_LIMBS; ++i) {
      int j = k - i;
      if (j < 0 || j >= BN_LIMBS || i > j) continue;
      if (i == j) BN_MAC(acc, a[i], a[i]); else BN_MAC(acc, a2[i], a[j]);
    }
#pragma unroll
    for (int i = 0; i < BN_LIMBS; ++i) {
      int j = k - i;
      if (j < 0 || j >= BN_LIMBS) continue;
      if (k < BN_LIMBS && i >= k) continue;
      BN_MAC(acc, m[i], q[j]);
    }
    if (k < BN_LIMBS) {
      m[k] = (int32_t)(((uint32_t)acc * BN_N0) & BN_MASK);
      BN_MAC(acc, m[k], q[0]);
    } else {
      r[k - BN_LIMBS] = (int32_t)((uint32_t)acc & BN_MASK);
    }
    BN_COLUMN_SHIFT(acc);
  }
  r[BN_LIMBS - 1] = (int32_t)acc;
  return r;
}

#if defined(BN_TRACK_BOUNDS) && !defined(__HIPCC__)
static inline void bn_trk_mul(Fp& r, const Fp& a, const Fp& b) {
  double A = bn_absmax(a), B = bn_absmax(b);
  double col = 10.0 * A * B + 10.0 * 18014398509481984.0 /* 2^54 */ + 137438953472.0 /* 2^37 */;
  if (col >= 9223372036854775808.0) bn_bound_fail("mul column overflow: 10*A*B", col);
  double vv = bn_vabs(a) * bn_vabs(b) / 86000.0;   // |a||b| / (q R) in units of q  (q/R = 2^-16.4 < 1/86000)
  if (vv > 64.0) { fprintf(stderr, "  |a| < %g q, |b| < %g q, limbs %g %g\n", bn_vabs(a), bn_vabs(b), A / BN_T, B / BN_T); bn_bound_fail("mul value bound |a||b|/(qR)", vv); }
  // value = (ab + mq)/R with 0 <= m < R
  double plo = std::fmin(std::fmin(a.bd.vlo * b.bd.vlo, a.bd.vlo * b.bd.vhi), std::fmin(a.bd.vhi * b.bd.vlo, a.bd.vhi * b.bd.vhi)) / 86000.0;
  double phi = std::fmax(std::fmax(a.bd.vlo * b.bd.vlo, a.bd.vlo * b.bd.vhi), std::fmax(a.bd.vhi * b.bd.vlo, a.bd.vhi * b.bd.vhi)) / 86000.0;
  bn_set_tight(r, std::fmin(plo, 0.0), 1.0 + std::fmax(phi, 0.0));
}
#endif

BN_DEV Fp fp_mul(const Fp& a, const Fp& b) {
  BN_LIMB_VEC x, y;
#pragma unroll
  for (int i = 0; i < BN_LIMBS; ++i) { x[i] = a.v[i]; y[i] = b.v[i]; }
  BN_LIMB_VEC z = fp_mul_impl(x, y);
  Fp r;
#pragma unroll
  for (int i = 0; i < BN_LIMBS; ++i) r.v[i] = z[i];
  BN_TRK(bn_trk_mul(r, a, b));
  return r;
}
BN_DEV Fp fp_sqr(const Fp& a) {
  BN_LIMB_VEC x;
#pragma unroll
  for (int i = 0; i < BN_LIMBS; ++i) x[i] = a.v[i];
  BN_LIMB_VEC z = fp_sqr_impl(x);
  Fp r;
#pragma unroll
  for (int i = 0; i < BN_LIMBS; ++i) r.v[i] = z[i];
  BN_TRK(bn_trk_mul(r, a, a));
  return r;
}

// ---- lazy-reduction Fq2 product -------------------------------------------------------------------
// (a0 + a1 i)(b0 + b1 i): both limb products of each output coefficient are accumulated into the same
// 64-bit columns (re: a0*b0 + (-a1)*b1, im: a0*b1 + a1*b0) and reduced ONCE — 600 multiplies like a
// 3-product Karatsuba, but no Karatsuba additions, no separate carry normalisation (outputs are tight)
// and one call instead of three.  Column bound: 10*(A0*B0 + A1*B1) + 10*2^54 + 2^37 < 2^63.
#if defined(__HIPCC__)
typedef int32_t bn_i32x10 __attribute__((ext_vector_type(10)));
typedef int32_t bn_i32x20 __attribute__((ext_vector_type(20)));
#define BN_VEC10 bn_i32x10
#define BN_VEC20 bn_i32x20
#else
struct bn_vec10 { int32_t e[10]; int32_t& operator[](int i) { return e[i]; } const int32_t& operator[](int i) const { return e[i]; } };
struct bn_vec20 { int32_t e[20]; int32_t& operator[](int i) { return e[i]; } const int32_t& operator[](int i) const { return e[i]; } };
#define BN_VEC10 bn_vec10
#define BN_VEC20 bn_vec20
#endif
// r = Montgomery-reduce(x0*y0 + x1*y1), limbs as plain arrays (a macro for the same reason as BN_MONT_PRODUCT_BODY)
#define BN_MONT_DUAL_BODY(x0, y0, x1, y1, r)                                             \
  do {                                                                                   \
    const int32_t q_[BN_LIMBS] = BN_QL_ARRAY;                                            \
    int64_t acc_ = 0;                                                                    \
    int32_t m_[BN_LIMBS];                                                                \
    _Pragma("unroll") for (int k_ = 0; k_ < 2 * BN_LIMBS - 1; ++k_) {                    \
      _Pragma("unroll") for (int i_ = 0; i_ < BN_LIMBS; ++i_) {                          \
        int j_ = k_ - i_;                                                                \
        if (j_ < 0 || j_ >= BN_LIMBS) continue;                                          \
        BN_MAC(acc_, (x0)[i_], (y0)[j_]);                                                \
        BN_MAC(acc_, (x1)[i_], (y1)[j_]);                                                \
      }                                                                                  \
      _Pragma("unroll") for (int i_ = 0; i_ < BN_LIMBS; ++i_) {                          \
        int j_ = k_ - i_;                                                                \
        if (j_ < 0 || j_ >= BN_LIMBS) continue;                                          \
        if (k_ < BN_LIMBS && i_ >= k_) continue;                                         \
        BN_MAC(acc_, m_[i_], q_[j_]);                                                    \
      }                                                                                  \
      if (k_ < BN_LIMBS) {                                                               \
        m_[k_] = (int32_t)(((uint32_t)acc_ * BN_N0) & BN_MASK);                          \
        BN_MAC(acc_, m_[k_], q_[0]);                                                     \
      } else {                                                                           \
        (r)[k_ - BN_LIMBS] = (int32_t)((uint32_t)acc_ & BN_MASK);                        \
      }                                                                                  \
      BN_COLUMN_SHIFT(acc_);                                                             \
    }                                                                                    \
    (r)[BN_LIMBS - 1] = (int32_t)acc_;                                                   \
  } while (0)
BN_DEV void fp_dual_mul_reduce(int32_t* r, const int32_t* x0, const int32_t* y0, const int32_t* x1, const int32_t* y1) {
  BN_MONT_DUAL_BODY(x0, y0, x1, y1, r);
}
#if !defined(BN_SPLIT_FP2)
BN_DEVN BN_VEC20 fp2_mul_impl(BN_VEC10 a0, BN_VEC10 a1, BN_VEC10 b0, BN_VEC10 b1) {
  BN_COUNT_MUL(); BN_COUNT_MUL(); BN_COUNT_MUL();   // algorithmic cost: a 3-product Karatsuba Fq2 multiplication
  int32_t x0[BN_LIMBS], x1[BN_LIMBS], y0[BN_LIMBS], y1[BN_LIMBS], n1[BN_LIMBS], re[BN_LIMBS], im[BN_LIMBS];
#pragma unroll
  for (int i = 0; i < BN_LIMBS; ++i) { x0[i] = a0[i]; x1[i] = a1[i]; y0[i] = b0[i]; y1[i] = b1[i]; n1[i] = -a1[i]; }
  BN_MONT_DUAL_BODY(x0, y0, n1, y1, re);
  BN_MONT_DUAL_BODY(x0, y1, x1, y0, im);
  BN_VEC20 r;
#pragma unroll
  for (int i = 0; i < BN_LIMBS; ++i) { r[i] = re[i]; r[BN_LIMBS + i] = im[i]; }
  return r;
}
#endif

// the unique representative in [0, q) with canonical limbs (same Montgomery residue).
// One product by the Montgomery one brings |value| into (-eps q, (1+eps) q); then at most one
// correction by q either way.
BN_DEVN Fp fp_canon(Fp a) {
  Fp t = fp_mul(a, fp_one());
  Fp ql = fp_load_const(C_QL);
  Fp up = fp_norm(fp_add(t, ql));
  t = fp_select(t.v[BN_LIMBS - 1] < 0, up, t);
  Fp dn = fp_norm(fp_sub(t, ql));
  t = fp_select(dn.v[BN_LIMBS - 1] >= 0, dn, t);
  BN_TRK(bn_set_tight(t, 0, 1));
  return t;
}
BN_DEV bool fp_limbs_all_zero(const Fp& a) {
  int32_t o = 0;
#pragma unroll
  for (int i = 0; i < BN_LIMBS; ++i) o |= a.v[i];
  return o == 0;
}
BN_DEV bool fp_is_zero(const Fp& a) { return fp_limbs_all_zero(fp_canon(a)); }
BN_DEV bool fp_eq(const Fp& a, const Fp& b) { return fp_is_zero(fp_sub(a, b)); }

// plain U256 (< 2^256) -> limbs (not yet Montgomery)
BN_DEV Fp fp_from_u256_plain(const U256& x) {
  Fp r;
#pragma unroll
  for (int i = 0; i < BN_LIMBS; ++i) {
    int bit = BN_W * i, w = bit >> 5, s = bit & 31;
    uint32_t v = x.w[w] >> s;
    if (s + BN_W > 32 && w + 1 < 8) v |= x.w[w + 1] << (32 - s);
    r.v[i] = (int32_t)(v & BN_MASK);
  }
  BN_TRK(bn_set_tight(r, 0, 5.3));   // any 256-bit integer is < 5.3 q
  return r;
}
// integer x (any U256) -> Montgomery form of x mod q
BN_DEV Fp fp_from_u256(const U256& x) { return fp_mul(fp_from_u256_plain(x), fp_load_const(C_R2)); }
// Montgomery element -> canonical integer in [0, q)
BN_DEVN U256 fp_to_u256(Fp a) {
  Fp one = fp_zero();
  one.v[0] = 1;
  BN_TRK(bn_set_tight(one, 0, 1));
  // a * 1 / R is the plain residue; canonicalise it with the same +-q correction as fp_canon
  Fp t = fp_mul(a, one);
  Fp ql = fp_load_const(C_QL);
  Fp up = fp_norm(fp_add(t, ql));
  t = fp_select(t.v[BN_LIMBS - 1] < 0, up, t);
  Fp dn = fp_norm(fp_sub(t, ql));
  t = fp_select(dn.v[BN_LIMBS - 1] >= 0, dn, t);
  U256 r;
#pragma unroll
  for (int w = 0; w < 8; ++w) r.w[w] = 0;
#pragma unroll
  for (int i = 0; i < BN_LIMBS; ++i) {
    int bit = BN_W * i, w = bit >> 5, s = bit & 31;
    r.w[w] |= (uint32_t)t.v[i] << s;
    if (s + BN_W > 32 && w + 1 < 8) r.w[w + 1] |= (uint32_t)t.v[i] >> (32 - s);
  }
  return r;
}
// a^e for a fixed public exponent given as a width-4 sliding-window schedule (bn254_constants.h): the odd
// powers a, a^3, .., a^15, then {squarings, multiplier} steps — ~250 squarings + ~48 multiplications + 8 for
// the table instead of 256 + ~110 for plain square-and-multiply.  Wave-uniform control flow.
BN_DEVN Fp fp_pow_sched(Fp a, const unsigned char (*sched)[2], int n_steps) {
  Fp odd[8];
  a = fp_norm(a);
  odd[0] = a;
  Fp a2 = fp_sqr(a);
  for (int i = 1; i < 8; ++i) odd[i] = fp_mul(odd[i - 1], a2);
  Fp acc = odd[sched[0][1] >> 1];
  for (int s = 1; s < n_steps; ++s) {
    BN_SET_STEP_PRIORITY(s);
    for (int k = 0; k < sched[s][0]; ++k) acc = fp_sqr(acc);
    if (sched[s][1]) acc = fp_mul(acc, odd[sched[s][1] >> 1]);
  }
  return acc;
}
BN_DEV Fp fp_inv(const Fp& a) { return fp_pow_sched(a, C_SCHED_QM2, BN_SCHED_QM2_LEN); }   // Fermat; inv(0) = 0
// y = a^((q+1)/4) (q = 3 mod 4); returns true iff y^2 == a
BN_DEV bool fp_sqrt(Fp& y, const Fp& a) {
  y = fp_pow_sched(a, C_SCHED_QP1D4, BN_SCHED_QP1D4_LEN);
  return fp_eq(fp_sqr(y), a);
}

#if defined(BN_SPLIT_FP2)
}  // namespace bn254
#include "bn254_fp2_pair.h"
namespace bn254 {
#else
// ------------------------------------------------------------------------------------------
// Fq2.  Inline helpers are LAZY: results of add/sub/mul_xi/mul carry whatever limb bounds their
// inputs imply; callers place fp2_norm where the next product needs it (tests/test_bounds.py).
// ------------------------------------------------------------------------------------------
BN_DEV Fp2 fp2_zero() { Fp2 r; r.c0 = fp_zero(); r.c1 = fp_zero(); return r; }
BN_DEV Fp2 fp2_one() { Fp2 r; r.c0 = fp_one(); r.c1 = fp_zero(); return r; }
BN_DEV Fp2 fp2_load_const(const int32_t (*c)[BN_LIMBS]) { Fp2 r; r.c0 = fp_load_const(c[0]); r.c1 = fp_load_const(c[1]); return r; }
BN_DEV Fp2 fp2_add(const Fp2& a, const Fp2& b) { Fp2 r; r.c0 = fp_add(a.c0, b.c0); r.c1 = fp_add(a.c1, b.c1); return r; }
BN_DEV Fp2 fp2_sub(const Fp2& a, const Fp2& b) { Fp2 r; r.c0 = fp_sub(a.c0, b.c0); r.c1 = fp_sub(a.c1, b.c1); return r; }
BN_DEV Fp2 fp2_neg(const Fp2& a) { Fp2 r; r.c0 = fp_neg(a.c0); r.c1 = fp_neg(a.c1); return r; }
BN_DEV Fp2 fp2_dbl(const Fp2& a) { return fp2_add(a, a); }
BN_DEV Fp2 fp2_conj(const Fp2& a) { Fp2 r; r.c0 = a.c0; r.c1 = fp_neg(a.c1); return r; }
BN_DEV Fp2 fp2_norm(const Fp2& a) { Fp2 r; r.c0 = fp_norm(a.c0); r.c1 = fp_norm(a.c1); return r; }
BN_DEV Fp2 fp2_reduce_weak(const Fp2& a) { Fp2 r; r.c0 = fp_reduce_weak(a.c0); r.c1 = fp_reduce_weak(a.c1); return r; }
BN_DEV bool fp2_is_zero(const Fp2& a) { return fp_is_zero(a.c0) && fp_is_zero(a.c1); }
BN_DEV bool fp2_eq(const Fp2& a, const Fp2& b) { return fp_eq(a.c0, b.c0) && fp_eq(a.c1, b.c1); }
BN_DEV Fp2 fp2_select(bool c, const Fp2& a, const Fp2& b) { Fp2 r; r.c0 = fp_select(c, a.c0, b.c0); r.c1 = fp_select(c, a.c1, b.c1); return r; }
#if defined(BN_TRACK_BOUNDS) && !defined(__HIPCC__)
static inline void bn_trk_fp2mul(Fp2& r, const Fp2& a, const Fp2& b) {
  double A0 = bn_absmax(a.c0), A1 = bn_absmax(a.c1), B0 = bn_absmax(b.c0), B1 = bn_absmax(b.c1);
  double extra = 10.0 * 18014398509481984.0 + 137438953472.0;
  double col_re = 10.0 * (A0 * B0 + A1 * B1) + extra, col_im = 10.0 * (A0 * B1 + A1 * B0) + extra;
  if (col_re >= 9223372036854775808.0 || col_im >= 9223372036854775808.0) bn_bound_fail("fp2_mul column overflow", std::fmax(col_re, col_im));
  auto prod = [](const Fp& x, const Fp& y, double& lo, double& hi) {
    double c[4] = {x.bd.vlo * y.bd.vlo, x.bd.vlo * y.bd.vhi, x.bd.vhi * y.bd.vlo, x.bd.vhi * y.bd.vhi};
    lo = std::fmin(std::fmin(c[0], c[1]), std::fmin(c[2], c[3])) / 86000.0;
    hi = std::fmax(std::fmax(c[0], c[1]), std::fmax(c[2], c[3])) / 86000.0;
  };
  double l00, h00, l11, h11, l01, h01, l10, h10;
  prod(a.c0, b.c0, l00, h00); prod(a.c1, b.c1, l11, h11); prod(a.c0, b.c1, l01, h01); prod(a.c1, b.c0, l10, h10);
  double re_lo = l00 - h11, re_hi = h00 - l11, im_lo = l01 + l10, im_hi = h01 + h10;
  if (std::fmax(std::fmax(std::fabs(re_lo), std::fabs(re_hi)), std::fmax(std::fabs(im_lo), std::fabs(im_hi))) > 64.0) bn_bound_fail("fp2_mul value bound", re_hi);
  bn_set_tight(r.c0, re_lo, re_hi + 1.0);
  bn_set_tight(r.c1, im_lo, im_hi + 1.0);
}
#endif
BN_DEV Fp2 fp2_mul(const Fp2& a, const Fp2& b) {   // lazy-reduction product; outputs are tight
  BN_VEC10 a0, a1, b0, b1;
#pragma unroll
  for (int i = 0; i < BN_LIMBS; ++i) { a0[i] = a.c0.v[i]; a1[i] = a.c1.v[i]; b0[i] = b.c0.v[i]; b1[i] = b.c1.v[i]; }
  BN_VEC20 z = fp2_mul_impl(a0, a1, b0, b1);
  Fp2 r;
#pragma unroll
  for (int i = 0; i < BN_LIMBS; ++i) { r.c0.v[i] = z[i]; r.c1.v[i] = z[BN_LIMBS + i]; }
  BN_TRK(bn_trk_fp2mul(r, a, b));
  return r;
}
BN_DEV Fp2 fp2_sqr(const Fp2& a) {                 // 2 Fq products
  Fp2 r;
  r.c0 = fp_mul(fp_add(a.c0, a.c1), fp_sub(a.c0, a.c1));
  r.c1 = fp_mul(fp_dbl(a.c0), a.c1);                // doubled before the product: the output is tight, as in the pair layout
  return r;
}
BN_DEV Fp2 fp2_mul_fp(const Fp2& a, const Fp& k) { Fp2 r; r.c0 = fp_mul(a.c0, k); r.c1 = fp_mul(a.c1, k); return r; }
BN_DEV Fp2 fp2_mul_xi(const Fp2& a) {              // (9 + i) * a; limb bounds grow 10x: input must be (near) tight
  Fp a2 = fp_dbl(a.c0), a4 = fp_dbl(a2), a8 = fp_dbl(a4);
  Fp b2 = fp_dbl(a.c1), b4 = fp_dbl(b2), b8 = fp_dbl(b4);
  Fp2 r;
  r.c0 = fp_sub(fp_add(a8, a.c0), a.c1);
  r.c1 = fp_add(fp_add(b8, a.c1), a.c0);
  return r;
}
BN_DEV Fp2 fp2_mul_xi_n(const Fp2& a) { return fp2_mul_xi(fp2_norm(a)); }
BN_DEV Fp2 fp2_inv(const Fp2& a) {
  Fp n = fp_inv(fp_add(fp_sqr(a.c0), fp_sqr(a.c1)));
  Fp2 r;
  r.c0 = fp_mul(a.c0, n);
  r.c1 = fp_neg(fp_mul(a.c1, n));
  return r;
}

// i * a = (-a1, a0); the element re + im i; u512 order of utils.rs:40-45 — the three primitives besides the ring
// operations that the layout-independent code below (fp2_sqrt, the compressed G2 codec) needs
BN_DEV Fp2 fp2_mul_i(const Fp2& a) { Fp2 r; r.c0 = fp_neg(a.c1); r.c1 = a.c0; return r; }
BN_DEV Fp2 fp2_make(const Fp& re, const Fp& im) { Fp2 r; r.c0 = re; r.c1 = im; return r; }
// canonical "u512(c) = c.im * q + c.re" order == lexicographic (im, re)
BN_DEV bool fp2_u512_greater(const Fp2& a, const Fp2& b) {
  U256 ai = fp_to_u256(a.c1), bi = fp_to_u256(b.c1), ar = fp_to_u256(a.c0), br = fp_to_u256(b.c0);
  bool im_eq = true;
  for (int i = 0; i < 8; ++i) im_eq = im_eq && ai.w[i] == bi.w[i];
  if (!im_eq) return u256_geq(ai.w, bi.w);
  bool re_eq = true;
  for (int i = 0; i < 8; ++i) re_eq = re_eq && ar.w[i] == br.w[i];
  return !re_eq && u256_geq(ar.w, br.w);
}

#endif  // BN_SPLIT_FP2

// ---- layout-independent Fq2 code (classic and pair layout alike) ---------------------------------------------
// a^e in Fq2 for a fixed public exponent (plain U256 words); wave-uniform control flow
BN_DEVN Fp2 fp2_pow_sched(Fp2 a, const unsigned char (*sched)[2], int n_steps) {   // as fp_pow_sched
  Fp2 odd[8];
  a = fp2_norm(a);
  odd[0] = a;
  Fp2 a2 = fp2_norm(fp2_sqr(a));
  for (int i = 1; i < 8; ++i) odd[i] = fp2_mul(odd[i - 1], a2);
  Fp2 acc = odd[sched[0][1] >> 1];
  for (int s = 1; s < n_steps; ++s) {
    BN_SET_STEP_PRIORITY(s);
    for (int k = 0; k < sched[s][0]; ++k) acc = fp2_norm(fp2_sqr(acc));
    if (sched[s][1]) acc = fp2_mul(acc, odd[sched[s][1] >> 1]);
  }
  return acc;
}
// square root in Fq2 for q = 3 mod 4 (complex method, Adj & Rodriguez-Henriquez alg. 9):
// a1 = a^((q-3)/4), alpha = a1^2 a, a0 = alpha^(q+1); a0 == -1 -> no root; x0 = a1 a;
// alpha == -1 -> x = i x0, else x = (1+alpha)^((q-1)/2) x0.  Returns true iff x^2 == a.
BN_DEVN bool fp2_sqrt(Fp2& x, const Fp2& a_in) {
  Fp2 a = fp2_norm(a_in);
  Fp2 a1 = fp2_pow_sched(a, C_SCHED_QM3D4, BN_SCHED_QM3D4_LEN);
  Fp2 alpha = fp2_mul(fp2_norm(fp2_sqr(a1)), a);
  Fp2 x0 = fp2_mul(a1, a);
  Fp2 minus_one = fp2_norm(fp2_neg(fp2_one()));
  bool alpha_is_m1 = fp2_eq(alpha, minus_one);
  Fp2 b = fp2_pow_sched(fp2_add(fp2_one(), alpha), C_SCHED_QM1D2, BN_SCHED_QM1D2_LEN);
  Fp2 xb = fp2_mul(b, x0);
  Fp2 xi_ = fp2_norm(fp2_mul_i(x0));
  x = fp2_select(alpha_is_m1, xi_, xb);
  return fp2_eq(fp2_sqr(x), a);
}



// ------------------------------------------------------------------------------------------
// Fq6, Fq12.  Fq12-level operations are real (non-inlined) functions on the per-lane private segment;
// the Fq6 layer is inlined into them so that all intermediates of one Fq12 operation live in VGPRs and
// each operand crosses memory once (rocprofv3 showed the earlier call-per-Fq6-product structure moving
// ~1.5 MB of private-segment traffic per verify: HBM-bound instead of VALU-bound).
// Contract: inputs with |limb| <= 2^27 ("tight"), outputs tight again (norm at the end).
// ------------------------------------------------------------------------------------------
BN_DEV void fp6_add(Fp6& r, const Fp6& a, const Fp6& b) { r.c0 = fp2_add(a.c0, b.c0); r.c1 = fp2_add(a.c1, b.c1); r.c2 = fp2_add(a.c2, b.c2); }
BN_DEV void fp6_sub(Fp6& r, const Fp6& a, const Fp6& b) { r.c0 = fp2_sub(a.c0, b.c0); r.c1 = fp2_sub(a.c1, b.c1); r.c2 = fp2_sub(a.c2, b.c2); }
BN_DEV void fp6_neg(Fp6& r, const Fp6& a) { r.c0 = fp2_neg(a.c0); r.c1 = fp2_neg(a.c1); r.c2 = fp2_neg(a.c2); }
BN_DEV void fp6_norm(Fp6& r, const Fp6& a) { r.c0 = fp2_norm(a.c0); r.c1 = fp2_norm(a.c1); r.c2 = fp2_norm(a.c2); }
BN_DEV void fp6_mul_v(Fp6& r, const Fp6& a) { Fp2 t = fp2_mul_xi(a.c2); r.c2 = a.c1; r.c1 = a.c0; r.c0 = t; }   // a.c2 tight

BN_DEV void fp6_mul(Fp6& r, const Fp6& a, const Fp6& b) {
  Fp2 v0 = fp2_mul(a.c0, b.c0), v1 = fp2_mul(a.c1, b.c1), v2 = fp2_mul(a.c2, b.c2);
  Fp2 c0 = fp2_add(fp2_mul_xi_n(fp2_sub(fp2_sub(fp2_mul(fp2_add(a.c1, a.c2), fp2_add(b.c1, b.c2)), v1), v2)), v0);
  Fp2 c1 = fp2_add(fp2_sub(fp2_sub(fp2_mul(fp2_add(a.c0, a.c1), fp2_add(b.c0, b.c1)), v0), v1), fp2_mul_xi(v2));
  Fp2 c2 = fp2_add(fp2_sub(fp2_sub(fp2_mul(fp2_add(a.c0, a.c2), fp2_add(b.c0, b.c2)), v0), v2), v1);
  r.c0 = fp2_norm(c0); r.c1 = fp2_norm(c1); r.c2 = fp2_norm(c2);
}
BN_DEV void fp6_mul_fp2(Fp6& r, const Fp6& a, const Fp2& k) {
  Fp2 c0 = fp2_mul(a.c0, k), c1 = fp2_mul(a.c1, k), c2 = fp2_mul(a.c2, k);
  r.c0 = c0; r.c1 = c1; r.c2 = c2;
}
// a * (b0 + b1 v)
BN_DEV void fp6_mul_01(Fp6& r, const Fp6& a, const Fp2& b0, const Fp2& b1) {
  Fp2 v0 = fp2_mul(a.c0, b0), v1 = fp2_mul(a.c1, b1);
  Fp2 c0 = fp2_add(fp2_mul_xi(fp2_mul(a.c2, b1)), v0);
  Fp2 c1 = fp2_sub(fp2_sub(fp2_mul(fp2_add(a.c0, a.c1), fp2_add(b0, b1)), v0), v1);
  Fp2 c2 = fp2_add(fp2_mul(a.c2, b0), v1);
  r.c0 = c0; r.c1 = c1; r.c2 = fp2_norm(c2);
}
BN_DEVN void fp6_inv(Fp6& r, const Fp6& a) {
  Fp2 t0 = fp2_norm(fp2_sub(fp2_sqr(a.c0), fp2_mul_xi(fp2_mul(a.c1, a.c2))));
  Fp2 t1 = fp2_norm(fp2_sub(fp2_mul_xi_n(fp2_sqr(a.c2)), fp2_mul(a.c0, a.c1)));
  Fp2 t2 = fp2_norm(fp2_sub(fp2_sqr(a.c1), fp2_mul(a.c0, a.c2)));
  Fp2 d = fp2_add(fp2_mul_xi_n(fp2_add(fp2_mul(a.c2, t1), fp2_mul(a.c1, t2))), fp2_mul(a.c0, t0));
  d = fp2_norm(fp2_inv(fp2_norm(d)));
  r.c0 = fp2_mul(t0, d); r.c1 = fp2_mul(t1, d); r.c2 = fp2_mul(t2, d);
}

BN_DEV void fp12_set_one(Fp12& r) {
  r.c0.c0 = fp2_one(); r.c0.c1 = fp2_zero(); r.c0.c2 = fp2_zero();
  r.c1.c0 = fp2_zero(); r.c1.c1 = fp2_zero(); r.c1.c2 = fp2_zero();
}
BN_DEV bool fp12_is_one(const Fp12& a) {
  return fp2_eq(a.c0.c0, fp2_one()) && fp2_is_zero(a.c0.c1) && fp2_is_zero(a.c0.c2) && fp2_is_zero(a.c1.c0) &&
         fp2_is_zero(a.c1.c1) && fp2_is_zero(a.c1.c2);
}
BN_DEVN void fp12_mul(Fp12& r, const Fp12& a, const Fp12& b) {
  Fp6 t0, t1, s, t, u;
  fp6_mul(t0, a.c0, b.c0);
  fp6_mul(t1, a.c1, b.c1);
  fp6_add(s, a.c0, a.c1); fp6_norm(s, s);
  fp6_add(t, b.c0, b.c1); fp6_norm(t, t);
  fp6_mul(u, s, t);
  fp6_sub(u, u, t0);
  fp6_sub(u, u, t1);
  fp6_mul_v(s, t1);
  fp6_add(s, t0, s);
  fp6_norm(r.c0, s);
  fp6_norm(r.c1, u);
}
BN_DEVH void fp12_sqr(Fp12& r, const Fp12& a) {
  Fp6 ab, s, t, u;
  fp6_mul(ab, a.c0, a.c1);
  fp6_add(s, a.c0, a.c1);
  fp6_mul_v(t, a.c1);
  fp6_add(t, t, a.c0); fp6_norm(t, t);
  fp6_mul(u, s, t);
  fp6_sub(u, u, ab);
  fp6_mul_v(s, ab);
  fp6_sub(u, u, s);
  fp6_norm(r.c0, u);
  fp6_add(s, ab, ab);
  r.c1 = s;
}
BN_DEV void fp12_conj(Fp12& r, const Fp12& a) { r.c0 = a.c0; fp6_neg(r.c1, a.c1); fp6_norm(r.c1, r.c1); }
BN_DEVN void fp12_inv(Fp12& r, const Fp12& a) {
  Fp6 t0, t1, d;
  fp6_mul(t0, a.c0, a.c0);
  fp6_mul(t1, a.c1, a.c1);
  fp6_mul_v(t1, t1);
  fp6_sub(d, t0, t1); fp6_norm(d, d);
  fp6_inv(d, d);
  fp6_mul(t0, a.c1, d);
  fp6_mul(r.c0, a.c0, d);
  fp6_neg(t0, t0); fp6_norm(r.c1, t0);
}
// f * (l0 + (l1 + l2 v) w): the sparse shape of a D-twist line (l0 at w^0, l1 at w^1, l2 at w^3)
BN_DEVN void fp12_mul_line(Fp12& r, const Fp12& f, const Fp2& l0, const Fp2& l1, const Fp2& l2) {
  Fp6 t0, t1, s, u;
  fp6_mul_fp2(t0, f.c0, l0);
  fp6_mul_01(t1, f.c1, l1, l2);
  fp6_add(s, f.c0, f.c1); 
  fp6_mul_01(u, s, fp2_add(l0, l1), l2);
  fp6_sub(u, u, t0);
  fp6_sub(u, u, t1);
  fp6_mul_v(s, t1);
  fp6_add(s, t0, s);
  fp6_norm(r.c0, s);
  fp6_norm(r.c1, u);
}
// f * (b0 + b1 w): b0 a full Fq6, b1 = b10 + b11 v — the shape of a product of two lines
BN_DEVH void fp12_mul_line2(Fp12& r, const Fp12& f, const Fp6& b0, const Fp2& b10, const Fp2& b11) {
  Fp6 t0, t1, s, u, bs;
  fp6_mul(t0, f.c0, b0);
  fp6_mul_01(t1, f.c1, b10, b11);
  fp6_add(s, f.c0, f.c1);
  bs.c0 = fp2_add(b0.c0, b10); bs.c1 = fp2_add(b0.c1, b11); bs.c2 = b0.c2;
  fp6_mul(u, s, bs);
  fp6_sub(u, u, t0);
  fp6_sub(u, u, t1);
  fp6_mul_v(s, t1);
  fp6_add(s, t0, s);
  fp6_norm(r.c0, s);
  fp6_norm(r.c1, u);
}
// coefficient k of w^k in the polynomial basis: c[2i] = c0.c_i, c[2i+1] = c1.c_i
BN_DEV Fp2& fp12_coef(Fp12& a, int k) {
  Fp6& h = (k & 1) ? a.c1 : a.c0;
  return (k >> 1) == 0 ? h.c0 : (k >> 1) == 1 ? h.c1 : h.c2;
}
// q^power Frobenius, power in {1,2,3}
BN_DEVN void fp12_frob(Fp12& r, const Fp12& a, int power) {
  Fp12 t = a;
  for (int k = 0; k < 6; ++k) {
    Fp2& c = fp12_coef(t, k);
    Fp2 x = (power & 1) ? fp2_conj(c) : c;
    const int32_t (*g)[BN_LIMBS] = power == 1 ? C_FROB1[k] : power == 2 ? C_FROB2[k] : C_FROB3[k];
    c = fp2_mul(x, fp2_load_const(g));
  }
  r = t;
}
// (a + b s)^2 in Fq4 = Fq2[s]/(s^2 - xi): r0 = a^2 + xi b^2, r1 = 2ab
BN_DEV void fp4_sqr(Fp2& r0, Fp2& r1, const Fp2& a, const Fp2& b) {
  Fp2 a2 = fp2_sqr(a), b2 = fp2_sqr(b);
  r1 = fp2_norm(fp2_sub(fp2_sub(fp2_sqr(fp2_add(a, b)), a2), b2));
  r0 = fp2_norm(fp2_add(a2, fp2_mul_xi(b2)));
}
// Granger-Scott squaring for the cyclotomic subgroup (after the easy part of the final exp.)
// The outputs 3t -+ 2a are linear in a, so across a run of squarings the value doubles each time unless the 2a term
// uses a weakly reduced copy of a (fp_reduce_weak).  `reduce` = do that in this call; callers in a long run may skip
// it three times out of four (values stay below ~700 q, products tolerate |a||b| < 64 * 86 000 q^2; the bound
// tracker follows the actual sequences).  The branch is wave-uniform.
BN_DEVN void fp12_cyclotomic_sqr(Fp12& r, const Fp12& a, bool reduce = true) {
  Fp2 t0, t1, t2, t3, t4, t5;
  fp4_sqr(t0, t1, a.c0.c0, a.c1.c1);
  fp4_sqr(t2, t3, a.c1.c0, a.c0.c2);
  fp4_sqr(t4, t5, a.c0.c1, a.c1.c2);
  Fp2 a00 = a.c0.c0, a11 = a.c1.c1, a10 = a.c1.c0, a02 = a.c0.c2, a01 = a.c0.c1, a12 = a.c1.c2;
  if (reduce) {
    a00 = fp2_reduce_weak(a00); a11 = fp2_reduce_weak(a11); a10 = fp2_reduce_weak(a10);
    a02 = fp2_reduce_weak(a02); a01 = fp2_reduce_weak(a01); a12 = fp2_reduce_weak(a12);
  }
  Fp12 o;
  o.c0.c0 = fp2_norm(fp2_add(fp2_dbl(fp2_sub(t0, a00)), t0));
  o.c1.c1 = fp2_norm(fp2_add(fp2_dbl(fp2_add(t1, a11)), t1));
  t5 = fp2_norm(fp2_mul_xi(t5));
  o.c1.c0 = fp2_norm(fp2_add(fp2_dbl(fp2_add(t5, a10)), t5));
  o.c0.c2 = fp2_norm(fp2_add(fp2_dbl(fp2_sub(t4, a02)), t4));
  o.c0.c1 = fp2_norm(fp2_add(fp2_dbl(fp2_sub(t2, a01)), t2));
  o.c1.c2 = fp2_norm(fp2_add(fp2_dbl(fp2_add(t3, a12)), t3));
  r = o;
}

}  // namespace bn254
