// BN254 base field Fq and the tower Fq2 / Fq6 / Fq12 for gfx950 lanes.
//
// One field element per lane: 9 signed BALANCED limbs of 29 bits in VGPRs, Montgomery form (R = 2^261).
// The hot primitive is a product-scanning Montgomery product built from v_mad_i64_i32 (measured on
// MI355X: ~4.3-5.3 cycles per wave64 instruction per SIMD, twice a plain 32-bit VALU operation, the same
// as v_mul_lo/hi_u32 or v_fma_f64 — the multiply-add COUNT is what a product costs: unsaturated limbs let
// a 64-bit column accumulator absorb every carry, and 9 limbs need 81 + 81 multiply-adds where 10 limbs of
// 27 bits needed 100 + 100; no MFMA — integer carry-chain work).
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
// BN_DEVF: the two Fq12 routines of the loop of fp12_pow_u (fp12_cyclotomic_sqr, fp12_mul), 254 of the 270 Fq12-level
// calls of a final exponentiation.  A real function that keeps values alive across its product calls uses the
// callee-saved VGPRs (fp12_mul: all 112 of them) and saves / restores every one it touches on EACH call although its
// caller holds nothing in them: 224 scratch dwords per fp12_mul call.  A translation unit that defines
// BN_INLINE_FE_HOT gets *_hot variants inlined into that loop (one call site each), whose enclosing function then
// saves the registers once; everywhere else the ordinary functions are called.
#if defined(BN_INLINE_FE_HOT)
#define BN_DEVF BN_DEV
#else
#define BN_DEVF BN_DEVN
#endif
// BN_DEVM: the Miller loops as real functions (default) or inlined into their kernels (BN_INLINE_MILLER; measured and
// rejected: Miller launch 5.8 -> 8.3 ms per 65 536 — the kernel-level register allocation spills the twist point and the
// affine inputs around every step).  The accumulator f lives in LDS either way; BN_ASSUME_LDS tells the compiler so inside
// the function, which turns the flat_load / flat_store through the generic reference (flat aperture path, both vmcnt
// and lgkmcnt held) into ds_read / ds_write.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BN_NO_ASSUME_LDS)
#define BN_ASSUME_LDS(ptr) __builtin_assume(__builtin_amdgcn_is_shared((const void*)(ptr)))
#else
#define BN_ASSUME_LDS(ptr) do { } while (0)        /* host pass of hipcc */
#endif
#if defined(BN_INLINE_MILLER)
#define BN_DEVM BN_DEV
#else
#define BN_DEVM BN_DEVN
#endif
#else
#define BN_DEV static inline __attribute__((always_inline))
#define BN_DEVN static __attribute__((noinline))
#define BN_CONST static const
#define BN_DEVH BN_DEVN
#define BN_DEVF BN_DEVN
#define BN_DEVM BN_DEVN
#define BN_ASSUME_LDS(ptr) do { } while (0)
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
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__) && defined(BN_USE_RELOAD_FENCE)
#define BN_RELOAD_FENCE() asm volatile("" ::: "memory")
#else
#define BN_RELOAD_FENCE() do { } while (0)
#endif
#ifndef BN_SET_STEP_PRIORITY
#define BN_SET_STEP_PRIORITY(step) do { } while (0)
#endif

namespace bn254 {

// ------------------------------------------------------------------------------------------
// Fq: unsaturated signed limbs.
//
// An element is 9 int32 limbs of nominally 29 bits, value = sum v[i] * 2^(29 i), Montgomery form
// with R = 2^261.  Limbs are SIGNED and BALANCED: a "tight" element has limbs 0..7 in [-2^28, 2^28)
// and the top limb absorbs the rest.  Balanced digits are what makes 29 bits fit: a product of two
// tight limbs is < 2^56, so a 64-bit column holds 9 (a*b) + 9 (m*q) terms with room for operands that
// are lazy sums of a few tight values — with digits in [0, 2^29) a column of a dual product would
// overflow as soon as one operand is a sum of two.
//   * add / sub / neg / dbl are 9 plain v_add/v_sub (no carry chain, no modular correction);
//   * mul is a product-scanning Montgomery product: each of the 17 columns accumulates its
//     a_i*b_j and m_i*q_j terms in one 64-bit register with v_mad_i64_i32 — no carry handling at
//     all; m_k and the output digits are the sign-extended low 29 bits of the column (v_bfe_i32),
//     q itself is held in balanced digits, and the balanced m makes the result SYMMETRIC:
//     |value| <= |a||b| / R + q/2;
//   * norm() propagates carries so that limbs 0..7 are back in [-2^28, 2^28) (the top limb absorbs);
//   * norm_floor() / canon() produce digits in [0, 2^29) / the unique representative in [0, q) (only for
//     comparisons and output).
// Safety conditions (machine-checked by the bound-tracking host build, tests/test_bounds.py):
//   mul(a,b): 9 * max|a_i| * max|b_j| + 9 * 2^56 + 2^37 < 2^63   and   |a||b| / (qR) small enough for the top limb
//   every limb always fits int32.
// R / q = 169: a product shrinks values much less than the 86 000 of the earlier 10 x 27-bit layout (R = 2^270), so
// the formulas keep operands within a few q (the tracker follows the actual sequences) and the one place where an
// output is linear in an input (cyclotomic squaring) reduces that term weakly (fp_reduce_weak).
// ------------------------------------------------------------------------------------------

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
struct FpBounds { double lo, hi, top, vlo, vhi; };   // limbs 0..7 in [lo,hi], |limb 8| <= top, value/q in [vlo,vhi]
#define BN_BOUNDS_MEMBER FpBounds bd;
#define BN_T ((double)BN_HALF) /* 2^28: magnitude of a balanced digit */
extern "C" int bn_bound_soft;     // 1: record the violation in bn_bound_failed and go on (tests/norm_site_search.py)
extern "C" int bn_bound_failed;
static inline void bn_bound_fail(const char* what, double x) {
  if (bn_bound_soft) { bn_bound_failed = 1; return; }
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
// "tight": limbs 0..7 in [-2^28, 2^28); the top limb is then round(value / 2^232), |top| <= |value/q| * (q / 2^232) + 1
static inline void bn_set_tight(Fp& r, double vlo, double vhi) {
  r.bd.lo = -BN_T; r.bd.hi = BN_T; r.bd.vlo = vlo; r.bd.vhi = vhi;
  r.bd.top = std::fmax(std::fabs(vlo), std::fabs(vhi)) * BN_TOP_PER_Q + 2.0;
}
// "floor-tight": limbs 0..7 in [0, 2^29) (fp_norm_floor, fp_from_u256_plain)
static inline void bn_set_floor_tight(Fp& r, double vlo, double vhi) {
  bn_set_tight(r, vlo, vhi);
  r.bd.lo = 0; r.bd.hi = 2.0 * BN_T;
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
// r = c ? a : b where c is a constant of the lane's POSITION in a distributed layout (which coefficient / product / output its lane pair
// handles: bn254_nonet.h), not data.  Same instructions as fp_select; the difference is the proof: fp_select is followed by the bound
// tracker with the UNION of both operands' bounds (one pass covers every value of a data-dependent condition), this one with the bounds of
// the operand the position takes — the host emulations run EVERY position, so every lane class is proven with its own bounds.
BN_DEV Fp fp_select_pos(bool c, const Fp& a, const Fp& b) {
  Fp r = fp_select(c, a, b);
  BN_TRK(r.bd = c ? a.bd : b.bd);
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
// 8 * a WITHOUT growing the limbs eight-fold (8 x 2^28 would leave int32): the shift crosses the limb boundary with
// a BALANCED split  a_i = h_i * 2^26 + l_i,  l_i in [-2^25, 2^25):
//   limb i = 8 * l_i + h_{i-1}      (top limb: 8 * a_top + h_{top-1}; it is small)
// so the result's limbs stay within +-(2^28 + 2^6) for any int32 input.  This is what lets xi = 9 + i be applied
// to a tight (or slightly lazy) element without a carry pass: 9 a -+ b has limbs <= 2^28 + |a_i| + |b_i| (fp2_mul_xi).
#define BN_SPREAD 26
BN_DEV int32_t bn_spread_lo(int32_t a) { return (int32_t)((uint32_t)a << (32 - BN_SPREAD)) >> (32 - BN_SPREAD); }   // v_bfe_i32
BN_DEV int32_t bn_spread_hi(int32_t a) { return (int32_t)((uint32_t)a + (1u << (BN_SPREAD - 1))) >> BN_SPREAD; }
BN_DEV Fp fp_mul8_spread(const Fp& a) {
  Fp r;
  int32_t h = 0;
#pragma unroll
  for (int i = 0; i < BN_LIMBS - 1; ++i) {
    r.v[i] = bn_spread_lo(a.v[i]) * 8 + h;
    h = bn_spread_hi(a.v[i]);
  }
  r.v[BN_LIMBS - 1] = a.v[BN_LIMBS - 1] * 8 + h;
  BN_TRK(double am_ = std::fmax(std::fabs(a.bd.lo), std::fabs(a.bd.hi));
         if (am_ + 33554432.0 + 64.0 >= 2147483648.0 || 8.0 * a.bd.top + 64.0 >= 2147483648.0) bn_bound_fail("mul8_spread input", am_);
         r.bd = FpBounds({-BN_T - am_ / 67108864.0 - 1.0, BN_T + am_ / 67108864.0 + 1.0, 8.0 * a.bd.top + am_ / 67108864.0 + 1.0,
                          8.0 * a.bd.vlo, 8.0 * a.bd.vhi}));
  return r;
}
// the sign-extended low BN_W bits of x: the balanced digit of x, one v_bfe_i32
BN_DEV int32_t bn_digit(uint32_t x) { return (int32_t)(x << (32 - BN_W)) >> (32 - BN_W); }
// carry propagation: limbs 0..7 -> [-2^28, 2^28), the top limb absorbs; the value is unchanged
BN_DEV Fp fp_norm(const Fp& a) {
  Fp r;
  int32_t c = 0;
#pragma unroll
  for (int i = 0; i < BN_LIMBS - 1; ++i) {
    int32_t x = a.v[i] + c;
    r.v[i] = bn_digit((uint32_t)x);
    c = (int32_t)((uint32_t)x + (uint32_t)BN_HALF) >> BN_W;       // (x - digit) / 2^W
  }
  r.v[BN_LIMBS - 1] = a.v[BN_LIMBS - 1] + c;
  BN_TRK(if (bn_absmax(a) + BN_T + 64 >= 2147483648.0) bn_bound_fail("norm input", bn_absmax(a)); bn_set_tight(r, a.bd.vlo, a.bd.vhi));
  return r;
}
// carry propagation to digits in [0, 2^29) (floor): the top limb is then floor(value / 2^232), so its sign is the
// sign of the value — what fp_canon / fp_to_u256 need; not used on the arithmetic paths
BN_DEV Fp fp_norm_floor(const Fp& a) {
  Fp r;
  int32_t c = 0;
#pragma unroll
  for (int i = 0; i < BN_LIMBS - 1; ++i) {
    int32_t x = a.v[i] + c;
    r.v[i] = (int32_t)((uint32_t)x & BN_MASK);
    c = x >> BN_W;
  }
  r.v[BN_LIMBS - 1] = a.v[BN_LIMBS - 1] + c;
  BN_TRK(if (bn_absmax(a) + 64 >= 2147483648.0) bn_bound_fail("norm_floor input", bn_absmax(a)); bn_set_floor_tight(r, a.bd.vlo, a.bd.vhi));
  return r;
}

// Weak modular reduction (with carry propagation; the input may be lazy): subtract k*q with k = round(value / q)
// estimated from the top limb: k = mulhi(top + (q/2^232)/2, round(2^32 / (q/2^232))) (one v_mul_hi_i32; the multiplier
// is 0.99985 of the exact ratio; lazy lower limbs shift the estimate by < 10^-5).  The residue is unchanged, the limbs
// come out tight and the value within +-(0.51 + 0.0003 |V|) * q.  Used on the outputs of the Fq12-level operations:
// with R / q = 169 the products alone do not keep the values of a long chain bounded.
BN_DEV Fp fp_reduce_weak(const Fp& a) {
  const int32_t q[BN_LIMBS] = BN_QL_ARRAY;
  int32_t k = (int32_t)(((int64_t)(a.v[BN_LIMBS - 1] + BN_WEAK_HALF) * BN_WEAK_KMUL) >> 32);
  int32_t carry = 0;
  Fp r;
#pragma unroll
  for (int i = 0; i < BN_LIMBS; ++i) {
    int64_t acc = (int64_t)a.v[i] + carry - (int64_t)k * q[i];
    if (i < BN_LIMBS - 1) {
      r.v[i] = bn_digit((uint32_t)acc);
      carry = (int32_t)((acc + BN_HALF) >> BN_W);
    } else {
      r.v[i] = (int32_t)acc;
    }
  }
  BN_TRK(if (bn_absmax(a) + 4194304.0 >= 2147483648.0) bn_bound_fail("reduce_weak input limbs", bn_absmax(a));
         double va_ = bn_vabs(a); if (va_ > 600.0) bn_bound_fail("reduce_weak input value", va_);
         bn_set_tight(r, -0.0003 * va_ - 0.51, 0.0003 * va_ + 0.51));
  return r;
}

// weak_reduce(cx * x + cy * y) for small integer factors, in one carry pass with 64-bit limb accumulation: x and y may be
// lazy and the combination need not fit int32 limb-wise (3 * (a 4-fold lazy value) does not).  Three multiply-adds per
// limb; replaces carry(x), the 32-bit combination and fp_reduce_weak (Granger-Scott squaring: 3 t -+ 2 a).
BN_DEV Fp fp_lin2_reduce(const Fp& x, int32_t cx, const Fp& y, int32_t cy) {
  const int32_t q[BN_LIMBS] = BN_QL_ARRAY;
  const int32_t top = x.v[BN_LIMBS - 1] * cx + y.v[BN_LIMBS - 1] * cy;
  const int32_t k = (int32_t)(((int64_t)(top + BN_WEAK_HALF) * BN_WEAK_KMUL) >> 32);
  int64_t carry = 0;
  Fp r;
#pragma unroll
  for (int i = 0; i < BN_LIMBS; ++i) {
    int64_t acc = carry + (int64_t)x.v[i] * cx + (int64_t)y.v[i] * cy - (int64_t)k * q[i];
    if (i < BN_LIMBS - 1) {
      r.v[i] = bn_digit((uint32_t)acc);
      carry = (acc + BN_HALF) >> BN_W;
    } else {
      r.v[i] = (int32_t)acc;
    }
  }
  BN_TRK(double vv_ = std::fabs((double)cx) * bn_vabs(x) + std::fabs((double)cy) * bn_vabs(y);
         if (std::fabs((double)cx) * x.bd.top + std::fabs((double)cy) * y.bd.top + 4194304.0 >= 2147483648.0 || vv_ > 600.0) bn_bound_fail("lin2_reduce input", vv_);
         bn_set_tight(r, -0.0003 * vv_ - 0.51, 0.0003 * vv_ + 0.51));
  return r;
}
// ... and of FOUR terms with per-lane factors (the linear stage of the lane machine, bn254_lmachine.h: every linear output of a level is
// weak_reduce(sum k_j x_j), so it is tight and within +-(0.51 + 0.0003 sum |k_j| |x_j|) q whatever went in)
BN_DEV Fp fp_lin4_reduce(const Fp& a, int32_t ca, const Fp& b, int32_t cb, const Fp& c, int32_t cc, const Fp& d, int32_t cd) {
  const int32_t q[BN_LIMBS] = BN_QL_ARRAY;
  const int32_t top = a.v[BN_LIMBS - 1] * ca + b.v[BN_LIMBS - 1] * cb + c.v[BN_LIMBS - 1] * cc + d.v[BN_LIMBS - 1] * cd;
  const int32_t k = (int32_t)(((int64_t)(top + BN_WEAK_HALF) * BN_WEAK_KMUL) >> 32);
  int64_t carry = 0;
  Fp r;
#pragma unroll
  for (int i = 0; i < BN_LIMBS; ++i) {
    int64_t acc = carry + (int64_t)a.v[i] * ca + (int64_t)b.v[i] * cb + (int64_t)c.v[i] * cc + (int64_t)d.v[i] * cd - (int64_t)k * q[i];
    if (i < BN_LIMBS - 1) {
      r.v[i] = bn_digit((uint32_t)acc);
      carry = (acc + BN_HALF) >> BN_W;
    } else {
      r.v[i] = (int32_t)acc;
    }
  }
  BN_TRK(double fa_ = std::fabs((double)ca); double fb_ = std::fabs((double)cb); double fc_ = std::fabs((double)cc); double fd_ = std::fabs((double)cd);
         double vv_ = fa_ * bn_vabs(a) + fb_ * bn_vabs(b) + fc_ * bn_vabs(c) + fd_ * bn_vabs(d);
         if (fa_ * a.bd.top + fb_ * b.bd.top + fc_ * c.bd.top + fd_ * d.bd.top + 4194304.0 >= 2147483648.0 || vv_ > 600.0) bn_bound_fail("lin4_reduce input", vv_);
         if (fa_ + fb_ + fc_ + fd_ > 64.0) bn_bound_fail("lin4_reduce factors", fa_ + fb_ + fc_ + fd_);
         bn_set_tight(r, -0.0003 * vv_ - 0.51, 0.0003 * vv_ + 0.51));
  return r;
}

// Montgomery product a*b*R^-1 (mod q), product scanning.  Columns are accumulated in a signed
// 64-bit register; m_k = the balanced digit of (column * -q^-1) makes each column divisible by 2^29.
#if defined(__HIPCC__)
// the nine limbs of an operand travel in VGPRs across the call.  (Until round 5 this was a 16-wide vector_size type, a leftover of the
// 10-limb layout: two of them are 32 dwords, one more than the calling convention has argument registers, so every call of a
// single-lane product stored a — never read — dword of padding to the stack: 4 B x every Fq product of the hash, G1 and scaling paths.)
typedef int32_t bn_i32xl __attribute__((ext_vector_type(BN_LIMBS)));
#define BN_LIMB_VEC bn_i32xl
#else
struct bn_limbvec { int32_t e[16]; int32_t& operator[](int i) { return e[i]; } const int32_t& operator[](int i) const { return e[i]; } };
#define BN_LIMB_VEC bn_limbvec
#endif

// acc >>= 29 (arithmetic): one v_ashrrev_i64 — measured at the issue cost of a 32-bit VALU op on gfx950
// (profiles/r01_issue_mix_microbench.jsonl), cheaper than an alignbit + ashr pair
#define BN_COLUMN_SHIFT(acc) do { (acc) >>= BN_W; } while (0)
// an output column: digit = balanced low 29 bits (v_bfe_i32), carry = (acc - digit) / 2^29 = (acc + 2^28) >> 29
// (one 64-bit add of a constant + the shift)
#define BN_COLUMN_OUT(acc, digit) do { (digit) = bn_digit((uint32_t)(acc)); (acc) = ((acc) + (int64_t)BN_HALF) >> BN_W; } while (0)

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
        m_[k_] = bn_digit((uint32_t)acc_ * BN_N0);                                       \
        BN_MAC(acc_, m_[k_], q_[0]);                                                     \
        BN_COLUMN_SHIFT(acc_);                                                           \
      } else {                                                                           \
        BN_COLUMN_OUT(acc_, (r)[k_ - BN_LIMBS]);                                         \
      }                                                                                  \
    }                                                                                    \
    (r)[BN_LIMBS - 1] = (int32_t)acc_;                                                   \
  } while (0)
BN_DEVN BN_LIMB_VEC fp_mul_impl(BN_LIMB_VEC a, BN_LIMB_VEC b) {
  BN_COUNT_MUL();
  BN_LIMB_VEC r;
  BN_MONT_PRODUCT_BODY(a, b, r);
  return r;
}
// a^2: 45 limb products instead of 81 (cross terms through the doubled operand)
BN_DEV BN_LIMB_VEC fp_sqr_body(BN_LIMB_VEC a) {
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
      m[k] = bn_digit((uint32_t)acc * BN_N0);
      BN_MAC(acc, m[k], q[0]);
      BN_COLUMN_SHIFT(acc);
    } else {
      BN_COLUMN_OUT(acc, r[k - BN_LIMBS]);
    }
  }
  r[BN_LIMBS - 1] = (int32_t)acc;
  return r;
}
BN_DEVN BN_LIMB_VEC fp_sqr_impl(BN_LIMB_VEC a) { return fp_sqr_body(a); }

#if defined(BN_TRACK_BOUNDS) && !defined(__HIPCC__)
// the m*q part of a column (9 balanced x balanced digit products) plus the carry from the column below
#define BN_COL_EXTRA ((double)BN_LIMBS * BN_T * BN_T + 137438953472.0 /* 2^37 */)
#define BN_VALUE_CAP 500.0  /* |value| / q allowed for a product output: its top limb (value x 3.17e6) must stay inside int32; the column bound sees that limb too */
// Largest a*b column of a 9 x 9 limb product, with the top limbs bounded separately: columns 0..7 hold at most 8
// products of ordinary limbs; column 8 has 7 of those plus a_0 b_8 and a_8 b_0; columns 9..15 fewer of each; column 16
// is a_8 b_8 alone.  (The top limb carries the VALUE — |top| ~ |v/q| * 3.17e6 — so a large value costs two terms of a
// column, not nine: this is what lets operands of a few hundred q through without a weak reduction.)
static inline double bn_limb_abs(const Fp& a) { return std::fmax(std::fabs(a.bd.lo), std::fabs(a.bd.hi)); }
static inline double bn_col_ab(const Fp& a, const Fp& b) {
  double A = bn_limb_abs(a), B = bn_limb_abs(b), At = a.bd.top, Bt = b.bd.top;
  return std::fmax(std::fmax(8.0 * A * B, 7.0 * A * B + A * Bt + At * B), At * Bt);
}
static inline void bn_trk_mul(Fp& r, const Fp& a, const Fp& b) {
  double A = bn_absmax(a), B = bn_absmax(b);
  double col = bn_col_ab(a, b) + BN_COL_EXTRA;
  if (col >= 9223372036854775808.0) { if (!bn_bound_soft) fprintf(stderr, "  limbs %g x %g (units of 2^28)\n", A / BN_T, B / BN_T); bn_bound_fail("mul column overflow: 9*A*B", col); }
  double vv = bn_vabs(a) * bn_vabs(b) / BN_R_OVER_Q;   // |a||b| / (q R) in units of q
  if (vv > BN_VALUE_CAP) { if (!bn_bound_soft) fprintf(stderr, "  |a| < %g q, |b| < %g q, limbs %g %g\n", bn_vabs(a), bn_vabs(b), A / BN_T, B / BN_T); bn_bound_fail("mul value bound |a||b|/(qR)", vv); }
  // value = (ab + mq)/R with |m| <= R/2 (balanced digits)
  double plo = std::fmin(std::fmin(a.bd.vlo * b.bd.vlo, a.bd.vlo * b.bd.vhi), std::fmin(a.bd.vhi * b.bd.vlo, a.bd.vhi * b.bd.vhi)) / BN_R_OVER_Q;
  double phi = std::fmax(std::fmax(a.bd.vlo * b.bd.vlo, a.bd.vlo * b.bd.vhi), std::fmax(a.bd.vhi * b.bd.vlo, a.bd.vhi * b.bd.vhi)) / BN_R_OVER_Q;
  bn_set_tight(r, plo - 0.501, phi + 0.501);
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
// 64-bit columns (re: a0*b0 + (-a1)*b1, im: a0*b1 + a1*b0) and reduced ONCE — 6 x 81 multiply-adds like a
// 3-product Karatsuba, but no Karatsuba additions, no separate carry normalisation (outputs are tight)
// and one call instead of three.  Column bound: 9*(A0*B0 + A1*B1) + 9*2^56 + 2^37 < 2^63.
#if defined(__HIPCC__)
typedef int32_t bn_i32x10 __attribute__((ext_vector_type(BN_LIMBS)));        // "VEC10": one element's limbs (the name predates 9 limbs)
typedef int32_t bn_i32x20 __attribute__((ext_vector_type(2 * BN_LIMBS)));
#define BN_VEC10 bn_i32x10
#define BN_VEC20 bn_i32x20
#else
struct bn_vec10 { int32_t e[BN_LIMBS]; int32_t& operator[](int i) { return e[i]; } const int32_t& operator[](int i) const { return e[i]; } };
struct bn_vec20 { int32_t e[2 * BN_LIMBS]; int32_t& operator[](int i) { return e[i]; } const int32_t& operator[](int i) const { return e[i]; } };
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
        m_[k_] = bn_digit((uint32_t)acc_ * BN_N0);                                       \
        BN_MAC(acc_, m_[k_], q_[0]);                                                     \
        BN_COLUMN_SHIFT(acc_);                                                           \
      } else {                                                                           \
        BN_COLUMN_OUT(acc_, (r)[k_ - BN_LIMBS]);                                         \
      }                                                                                  \
    }                                                                                    \
    (r)[BN_LIMBS - 1] = (int32_t)acc_;                                                   \
  } while (0)
BN_DEV void fp_dual_mul_reduce(int32_t* r, const int32_t* x0, const int32_t* y0, const int32_t* x1, const int32_t* y1) {
  BN_MONT_DUAL_BODY(x0, y0, x1, y1, r);
}
// r = Montgomery-reduce(sum_{t < N} x[t] * y[t]): N limb products share the 64-bit columns and ONE reduction (x, y: int32_t [N][BN_LIMBS]).
// The Fq6-level lazy reduction (fp6_mul_lazy below): N = 6 is an output coefficient of a schoolbook Fq6 product in the pair layout — three
// Fq2 products, two limb products each.  Column bound: 8 * sum A_t B_t + 9 * 2^56 + 2^37 < 2^63, i.e. sum A_t B_t <= 14 units of (2^28)^2:
// six products of tight operands use 6 of them, so one operand of each may be a lazy sum of two.
#define BN_MONT_MULTI_BODY(N, x, y, r)                                                   \
  do {                                                                                   \
    const int32_t q_[BN_LIMBS] = BN_QL_ARRAY;                                            \
    int64_t acc_ = 0;                                                                    \
    int32_t m_[BN_LIMBS];                                                                \
    _Pragma("unroll") for (int k_ = 0; k_ < 2 * BN_LIMBS - 1; ++k_) {                    \
      _Pragma("unroll") for (int i_ = 0; i_ < BN_LIMBS; ++i_) {                          \
        int j_ = k_ - i_;                                                                \
        if (j_ < 0 || j_ >= BN_LIMBS) continue;                                          \
        _Pragma("unroll") for (int t_ = 0; t_ < (N); ++t_) BN_MAC(acc_, (x)[t_][i_], (y)[t_][j_]); \
      }                                                                                  \
      _Pragma("unroll") for (int i_ = 0; i_ < BN_LIMBS; ++i_) {                          \
        int j_ = k_ - i_;                                                                \
        if (j_ < 0 || j_ >= BN_LIMBS) continue;                                          \
        if (k_ < BN_LIMBS && i_ >= k_) continue;                                         \
        BN_MAC(acc_, m_[i_], q_[j_]);                                                    \
      }                                                                                  \
      if (k_ < BN_LIMBS) {                                                               \
        m_[k_] = bn_digit((uint32_t)acc_ * BN_N0);                                       \
        BN_MAC(acc_, m_[k_], q_[0]);                                                     \
        BN_COLUMN_SHIFT(acc_);                                                           \
      } else {                                                                           \
        BN_COLUMN_OUT(acc_, (r)[k_ - BN_LIMBS]);                                         \
      }                                                                                  \
    }                                                                                    \
    (r)[BN_LIMBS - 1] = (int32_t)acc_;                                                   \
  } while (0)
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
  Fp t = fp_norm_floor(fp_mul(a, fp_one()));         // |value| < 0.51 q + |a| / 169; floor digits: sign(top) = sign(value)
  Fp ql = fp_load_const(C_QL);
  Fp up = fp_norm_floor(fp_add(t, ql));
  t = fp_select(t.v[BN_LIMBS - 1] < 0, up, t);
  Fp dn = fp_norm_floor(fp_sub(t, ql));
  t = fp_select(dn.v[BN_LIMBS - 1] >= 0, dn, t);
  BN_TRK(if (bn_vabs(a) > 80.0) bn_bound_fail("canon input value", bn_vabs(a)); bn_set_floor_tight(t, 0, 1));
  return t;
}
BN_DEV bool fp_limbs_all_zero(const Fp& a) {
  int32_t o = 0;
#pragma unroll
  for (int i = 0; i < BN_LIMBS; ++i) o |= a.v[i];
  return o == 0;
}
// a == 0 (mod q).  Round 6: by the WEAK reduction instead of fp_canon.  fp_reduce_weak leaves the value inside (-0.7 q, 0.7 q) for any input of
// up to 600 q (its contract, which the bound tracker enforces at every call) with tight limbs — and in that interval the only multiple of q is 0,
// whose balanced-digit representation is unique (a digit in [-2^28, 2^28) that is 0 mod 2^29 is 0, limb by limb): all limbs zero.  ~60 32-bit
// instructions (one v_mul_hi, nine multiply-adds) where fp_canon — a Montgomery product by one and two conditional corrections — took ~400: the
// mixed additions of k_aggregate_pair test h and z per addition (two of these against their 11 products), the complete additions of the
// ladders four.  Same truth value on every input fp_canon accepted (|value| <= 80 q).
BN_DEV bool fp_is_zero(const Fp& a) { return fp_limbs_all_zero(fp_reduce_weak(a)); }
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
  BN_TRK(bn_set_floor_tight(r, 0, 5.3));   // any 256-bit integer is < 5.3 q
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
  Fp t = fp_norm_floor(fp_mul(a, one));
  Fp ql = fp_load_const(C_QL);
  Fp up = fp_norm_floor(fp_add(t, ql));
  t = fp_select(t.v[BN_LIMBS - 1] < 0, up, t);
  Fp dn = fp_norm_floor(fp_sub(t, ql));
  t = fp_select(dn.v[BN_LIMBS - 1] >= 0, dn, t);
  BN_TRK(if (bn_vabs(a) > 80.0) bn_bound_fail("to_u256 input value", bn_vabs(a)));
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
    for (int k = 0; k < sched[s][0]; ++k) acc = fp_sqr(acc);      // (the leaf inlined here: 33.5 against 32.2 ms per 16 Mi, profiles/r04_h_ab_inline_pow_leaves.log)
    if (sched[s][1]) acc = fp_mul(acc, odd[sched[s][1] >> 1]);
  }
  return acc;
}
BN_DEV Fp fp_inv_fermat(const Fp& a) { return fp_pow_sched(a, C_SCHED_QM2, BN_SCHED_QM2_LEN); }   // a^(q-2): 306 products; inv(0) = 0

// ---- modular inverse by division steps ("safegcd", Bernstein & Yang 2019; the batched form of Wuille's modinv) -------
// 21 batches of 29 division steps (609 >= the 590 that 256-bit inputs need) on the low limb of (f, g) = (q, x), each batch
// producing a 2x2 transition matrix that is then applied to the full f, g and to the Bezout pair d, e (mod q, with the
// 2^-29 of the batch folded in).  Constant control flow, no data-dependent branch: every lane of a wave walks the same
// path.  ~17 k mostly 32-bit instructions where the Fermat exponentiation a^(q-2) takes 306 Montgomery products (~79 k,
// multiply-adds).  Input: the canonical digits of an integer in [0, q).  Output: digits of a value congruent to the
// inverse, in (-2q, 2q), limbs in (-2^29, 2^29); 0 for input 0.
struct BnTrans { int32_t u, v, q, r; };
BN_DEV int32_t bn_divsteps(int32_t zeta, uint32_t f0, uint32_t g0, BnTrans& t) {
  uint32_t u = 1, v = 0, q = 0, r = 1, f = f0, g = g0;
#pragma unroll 1
  for (int i = 0; i < BN_W; ++i) {
    uint32_t c1 = (uint32_t)(zeta >> 31);                 // all ones if zeta < 0
    const uint32_t c2 = 0u - (g & 1u);                    // all ones if g is odd
    const uint32_t x = (f ^ c1) - c1, y = (u ^ c1) - c1, z = (v ^ c1) - c1;   // f, u, v negated if zeta < 0
    g += x & c2; q += y & c2; r += z & c2;
    c1 &= c2;
    zeta = (int32_t)((uint32_t)zeta ^ c1) - 1;            // -zeta - 2 if (zeta < 0 and g odd), else zeta - 1
    f += g & c1; u += q & c1; v += r & c1;
    g >>= 1; u <<= 1; v <<= 1;
  }
  t.u = (int32_t)u; t.v = (int32_t)v; t.q = (int32_t)q; t.r = (int32_t)r;
  return zeta;
}
// (f, g) <- t * (f, g) / 2^29   (exact)
BN_DEV void bn_update_fg(int32_t* f, int32_t* g, const BnTrans& t) {
  int64_t cf = (int64_t)t.u * f[0] + (int64_t)t.v * g[0], cg = (int64_t)t.q * f[0] + (int64_t)t.r * g[0];
  cf >>= BN_W; cg >>= BN_W;
#pragma unroll
  for (int i = 1; i < BN_LIMBS; ++i) {
    cf += (int64_t)t.u * f[i] + (int64_t)t.v * g[i];
    cg += (int64_t)t.q * f[i] + (int64_t)t.r * g[i];
    f[i - 1] = (int32_t)((uint32_t)cf & BN_MASK); cf >>= BN_W;
    g[i - 1] = (int32_t)((uint32_t)cg & BN_MASK); cg >>= BN_W;
  }
  f[BN_LIMBS - 1] = (int32_t)cf; g[BN_LIMBS - 1] = (int32_t)cg;
}
// (d, e) <- t * (d, e) / 2^29 mod q, both kept in (-2q, q)
BN_DEV void bn_update_de(int32_t* d, int32_t* e, const BnTrans& t) {
  const int32_t qf[BN_LIMBS] = {C_QF_0, C_QF_1, C_QF_2, C_QF_3, C_QF_4, C_QF_5, C_QF_6, C_QF_7, C_QF_8};
  const int32_t sd = d[BN_LIMBS - 1] >> 31, se = e[BN_LIMBS - 1] >> 31;
  int32_t md = (t.u & sd) + (t.v & se), me = (t.q & sd) + (t.r & se);
  int64_t cd = (int64_t)t.u * d[0] + (int64_t)t.v * e[0], ce = (int64_t)t.q * d[0] + (int64_t)t.r * e[0];
  md -= (int32_t)((BN_QINV * (uint32_t)cd + (uint32_t)md) & BN_MASK);
  me -= (int32_t)((BN_QINV * (uint32_t)ce + (uint32_t)me) & BN_MASK);
  cd += (int64_t)qf[0] * md; ce += (int64_t)qf[0] * me;
  cd >>= BN_W; ce >>= BN_W;
#pragma unroll
  for (int i = 1; i < BN_LIMBS; ++i) {
    cd += (int64_t)t.u * d[i] + (int64_t)t.v * e[i] + (int64_t)qf[i] * md;
    ce += (int64_t)t.q * d[i] + (int64_t)t.r * e[i] + (int64_t)qf[i] * me;
    d[i - 1] = (int32_t)((uint32_t)cd & BN_MASK); cd >>= BN_W;
    e[i - 1] = (int32_t)((uint32_t)ce & BN_MASK); ce >>= BN_W;
  }
  d[BN_LIMBS - 1] = (int32_t)cd; e[BN_LIMBS - 1] = (int32_t)ce;
}
#define BN_INV_BATCHES 21
BN_DEVN Fp fp_inv_plain_divsteps(Fp x) {                  // x: canonical digits of an integer in [0, q)
  int32_t f[BN_LIMBS] = {C_QF_0, C_QF_1, C_QF_2, C_QF_3, C_QF_4, C_QF_5, C_QF_6, C_QF_7, C_QF_8};
  int32_t g[BN_LIMBS], d[BN_LIMBS], e[BN_LIMBS];
#pragma unroll
  for (int i = 0; i < BN_LIMBS; ++i) { g[i] = x.v[i]; d[i] = 0; e[i] = 0; }
  e[0] = 1;
  int32_t zeta = -1;
#pragma unroll 1
  for (int b = 0; b < BN_INV_BATCHES; ++b) {
    BnTrans t;
    zeta = bn_divsteps(zeta, (uint32_t)f[0], (uint32_t)g[0], t);
    bn_update_de(d, e, t);
    bn_update_fg(f, g, t);
  }
  // g = 0 and f = +-1 now (f = q for x = 0, where d = 0): the inverse is d * f
  const int32_t s = f[BN_LIMBS - 1] >> 31;
  Fp r;
#pragma unroll
  for (int i = 0; i < BN_LIMBS; ++i) r.v[i] = (d[i] ^ s) - s;
  BN_TRK(r.bd = FpBounds({-2.0 * BN_T, 2.0 * BN_T, 2.0 * BN_TOP_PER_Q + 4.0, -2.0, 2.0}));
  return r;
}
// Montgomery inverse: a = x R  ->  x^-1 R = (plain inverse of the integer a) * R^2, i.e. one product with R^3; inv(0) = 0
BN_DEV Fp fp_inv(const Fp& a) { return fp_mul(fp_inv_plain_divsteps(fp_canon(a)), fp_load_const(C_R3)); }
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
BN_DEV Fp2 fp2_lin2_reduce(const Fp2& x, int32_t cx, const Fp2& y, int32_t cy) {
  Fp2 r; r.c0 = fp_lin2_reduce(x.c0, cx, y.c0, cy); r.c1 = fp_lin2_reduce(x.c1, cx, y.c1, cy); return r;
}
BN_DEV bool fp2_is_zero(const Fp2& a) { return fp_is_zero(a.c0) && fp_is_zero(a.c1); }
BN_DEV bool fp2_eq(const Fp2& a, const Fp2& b) { return fp_eq(a.c0, b.c0) && fp_eq(a.c1, b.c1); }
BN_DEV Fp2 fp2_select(bool c, const Fp2& a, const Fp2& b) { Fp2 r; r.c0 = fp_select(c, a.c0, b.c0); r.c1 = fp_select(c, a.c1, b.c1); return r; }
#if defined(BN_TRACK_BOUNDS) && !defined(__HIPCC__)
static inline void bn_trk_fp2mul(Fp2& r, const Fp2& a, const Fp2& b) {
  double extra = BN_COL_EXTRA;
  double col_re = bn_col_ab(a.c0, b.c0) + bn_col_ab(a.c1, b.c1) + extra, col_im = bn_col_ab(a.c0, b.c1) + bn_col_ab(a.c1, b.c0) + extra;
  if (col_re >= 9223372036854775808.0 || col_im >= 9223372036854775808.0) bn_bound_fail("fp2_mul column overflow", std::fmax(col_re, col_im));
  auto prod = [](const Fp& x, const Fp& y, double& lo, double& hi) {
    double c[4] = {x.bd.vlo * y.bd.vlo, x.bd.vlo * y.bd.vhi, x.bd.vhi * y.bd.vlo, x.bd.vhi * y.bd.vhi};
    lo = std::fmin(std::fmin(c[0], c[1]), std::fmin(c[2], c[3])) / BN_R_OVER_Q;
    hi = std::fmax(std::fmax(c[0], c[1]), std::fmax(c[2], c[3])) / BN_R_OVER_Q;
  };
  double l00, h00, l11, h11, l01, h01, l10, h10;
  prod(a.c0, b.c0, l00, h00); prod(a.c1, b.c1, l11, h11); prod(a.c0, b.c1, l01, h01); prod(a.c1, b.c0, l10, h10);
  double re_lo = l00 - h11, re_hi = h00 - l11, im_lo = l01 + l10, im_hi = h01 + h10;
  if (std::fmax(std::fmax(std::fabs(re_lo), std::fabs(re_hi)), std::fmax(std::fabs(im_lo), std::fabs(im_hi))) > BN_VALUE_CAP) bn_bound_fail("fp2_mul value bound", re_hi);
  bn_set_tight(r.c0, re_lo - 0.501, re_hi + 0.501);
  bn_set_tight(r.c1, im_lo - 0.501, im_hi + 0.501);
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
BN_DEV Fp2 fp2_mul_xi(const Fp2& a) {              // (9 + i) * a through fp_mul8_spread: limbs <= 2^28 + |a0_i| + |a1_i|
  Fp a8 = fp_mul8_spread(a.c0), b8 = fp_mul8_spread(a.c1);
  Fp2 r;
  r.c0 = fp_sub(fp_add(a8, a.c0), a.c1);
  r.c1 = fp_add(fp_add(b8, a.c1), a.c0);
  return r;
}
BN_DEV Fp2 fp2_mul_xi_n(const Fp2& a) { return fp2_mul_xi(fp2_norm(a)); }
BN_DEV Fp2 fp2_mul8(const Fp2& a) { Fp2 r; r.c0 = fp_mul8_spread(a.c0); r.c1 = fp_mul8_spread(a.c1); return r; }
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
  Fp2 a2 = fp2_sqr(a);
  for (int i = 1; i < 8; ++i) odd[i] = fp2_mul(odd[i - 1], a2);
  Fp2 acc = odd[sched[0][1] >> 1];
  for (int s = 1; s < n_steps; ++s) {
    BN_SET_STEP_PRIORITY(s);
    for (int k = 0; k < sched[s][0]; ++k) acc = fp2_sqr(acc);
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
  Fp2 alpha = fp2_mul(fp2_sqr(a1), a);
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
//
// Limb / value management with R = 2^261 (bn254_field.h header): a product column holds ~13 units of (2^28)^2, so
//   fp2_mul(a, b) needs A * B <= 6 (A, B = limb magnitudes in units of 2^28), fp2_sqr(a) needs A <= 1.8,
// and a product only shrinks values by R / q = 169 while xi = 9 + i and the Karatsuba subtractions expand them
// ~30-fold per Fq6 level.  Contract of the Fq12-level operations: inputs tight and of small value (|V| < ~1 q),
// outputs the same — every output coefficient passes through a SITE that carries (fp2_norm) or carries and weakly
// reduces (fp2_reduce_weak) it.  Which sites are needed at all, and in which mode, is decided by a search under the
// bound tracker (tests/norm_site_search.py) and recorded in bn254_norm_sites.h; the source states the safe default.
// ------------------------------------------------------------------------------------------
}  // namespace bn254
#include "bn254_norm_sites.h"                  // constexpr int bn_site_override(int id): -1 or the mode found by the search
namespace bn254 {
#if defined(BN_TRACK_BOUNDS) && !defined(__HIPCC__)
extern "C" signed char bn_site_mode[1024];     // starts as bn_site_override(id); the search rewrites entries at run time
extern "C" unsigned int bn_site_hits[1024];    // how often a flow passed through each site (the search's cost weights)
extern "C" signed char bn_site_dflt[1024];     // the default mode the source states for the site
#define BN_SITE_MODE(id, dflt) (++bn_site_hits[id], bn_site_dflt[id] = (dflt), bn_site_mode[id] < 0 ? (dflt) : (int)bn_site_mode[id])
#else
#define BN_SITE_MODE(id, dflt) (bn_site_override(id) < 0 ? (dflt) : bn_site_override(id))
#endif
BN_DEV Fp2 fp2_site(const Fp2& x, int mode) { return mode == 2 ? fp2_reduce_weak(x) : mode == 1 ? fp2_norm(x) : x; }
#define NS(id, x) fp2_site((x), BN_SITE_MODE(id, 1))   /* default: carry */
#define NR(id, x) fp2_site((x), BN_SITE_MODE(id, 2))   /* default: carry + weak reduction */

BN_DEV void fp6_add(Fp6& r, const Fp6& a, const Fp6& b) { r.c0 = fp2_add(a.c0, b.c0); r.c1 = fp2_add(a.c1, b.c1); r.c2 = fp2_add(a.c2, b.c2); }
BN_DEV void fp6_sub(Fp6& r, const Fp6& a, const Fp6& b) { r.c0 = fp2_sub(a.c0, b.c0); r.c1 = fp2_sub(a.c1, b.c1); r.c2 = fp2_sub(a.c2, b.c2); }
BN_DEV void fp6_neg(Fp6& r, const Fp6& a) { r.c0 = fp2_neg(a.c0); r.c1 = fp2_neg(a.c1); r.c2 = fp2_neg(a.c2); }
BN_DEV void fp6_norm(Fp6& r, const Fp6& a) { r.c0 = fp2_norm(a.c0); r.c1 = fp2_norm(a.c1); r.c2 = fp2_norm(a.c2); }
BN_DEV void fp6_mul_v(Fp6& r, const Fp6& a) { Fp2 t = fp2_mul_xi(a.c2); r.c2 = a.c1; r.c1 = a.c0; r.c0 = t; }
template <int S> BN_DEV void fp6_site_n(Fp6& r, const Fp6& a) { r.c0 = NS(S, a.c0); r.c1 = NS(S + 1, a.c1); r.c2 = NS(S + 2, a.c2); }
template <int S> BN_DEV void fp6_site_r(Fp6& r, const Fp6& a) { r.c0 = NR(S, a.c0); r.c1 = NR(S + 1, a.c1); r.c2 = NR(S + 2, a.c2); }

// a * b, Karatsuba: 6 Fq2 products.  Sites S .. S+3.
template <int S> BN_DEV void fp6_mul(Fp6& r, const Fp6& a, const Fp6& b) {
  Fp2 v0 = fp2_mul(a.c0, b.c0), v1 = fp2_mul(a.c1, b.c1), v2 = fp2_mul(a.c2, b.c2);
  Fp2 c0 = fp2_add(fp2_mul_xi(NS(S, fp2_sub(fp2_sub(fp2_mul(fp2_add(a.c1, a.c2), fp2_add(b.c1, b.c2)), v1), v2))), v0);
  Fp2 c1 = fp2_add(fp2_sub(fp2_sub(fp2_mul(fp2_add(a.c0, a.c1), fp2_add(b.c0, b.c1)), v0), v1), fp2_mul_xi(v2));
  Fp2 c2 = fp2_add(fp2_sub(fp2_sub(fp2_mul(fp2_add(a.c0, a.c2), fp2_add(b.c0, b.c2)), v0), v2), v1);
  r.c0 = NS(S + 1, c0); r.c1 = NS(S + 2, c1); r.c2 = NS(S + 3, c2);
}
BN_DEV void fp6_mul_fp2(Fp6& r, const Fp6& a, const Fp2& k) {
  Fp2 c0 = fp2_mul(a.c0, k), c1 = fp2_mul(a.c1, k), c2 = fp2_mul(a.c2, k);
  r.c0 = c0; r.c1 = c1; r.c2 = c2;
}
// a * (b0 + b1 v): 5 Fq2 products.  Sites S .. S+2.
template <int S> BN_DEV void fp6_mul_01(Fp6& r, const Fp6& a, const Fp2& b0, const Fp2& b1) {
  Fp2 v0 = fp2_mul(a.c0, b0), v1 = fp2_mul(a.c1, b1);
  Fp2 c0 = fp2_add(fp2_mul_xi(fp2_mul(a.c2, b1)), v0);
  Fp2 c1 = fp2_sub(fp2_sub(fp2_mul(fp2_add(a.c0, a.c1), fp2_add(b0, b1)), v0), v1);
  Fp2 c2 = fp2_add(fp2_mul(a.c2, b0), v1);
  r.c0 = NS(S, c0); r.c1 = NS(S + 1, c1); r.c2 = NS(S + 2, c2);
}
#if defined(BN_SPLIT_FP2) && defined(BN_FP6_LAZY)
// ---- Fq6-level lazy reduction (pair layout) ----------------------------------------------------------------------------
// a * b as the SCHOOLBOOK product with one reduction per output coefficient: xi is applied to the operands a1, a2 (carried, so they are
// tight again), and every coefficient is a sum of three Fq2 products accumulated in the same 64-bit columns (fp2_mul_sum):
//   c0 = a0 b0 + (xi a1) b2 + (xi a2) b1     c1 = a0 b1 + a1 b0 + (xi a2) b2     c2 = a0 b2 + a1 b1 + a2 b0
// 18 limb products + 3 reductions per lane against the Karatsuba form's 12 + 6 — the same 486 + ... multiply-adds per three outputs, but
// none of its 15 additions, four carry sites, two late xi and six calls (each with its own partner exchanges and argument moves).
// Contract: b tight; a tight or a lazy sum of two tight values; outputs tight.
BN_DEV void fp6_mul_lazy(Fp6& r, const Fp6& a, const Fp6& b) {
  const Fp2 xa1 = fp2_norm(fp2_mul_xi(a.c1)), xa2 = fp2_norm(fp2_mul_xi(a.c2));
  const Fp2* const x0[3] = {&a.c0, &xa1, &xa2};
  const Fp2* const y0[3] = {&b.c0, &b.c2, &b.c1};
  const Fp2* const x1[3] = {&a.c0, &a.c1, &xa2};
  const Fp2* const y1[3] = {&b.c1, &b.c0, &b.c2};
  const Fp2* const x2[3] = {&a.c0, &a.c1, &a.c2};
  const Fp2* const y2[3] = {&b.c2, &b.c1, &b.c0};
  const Fp2 c0 = fp2_mul_sum<3>(x0, y0), c1 = fp2_mul_sum<3>(x1, y1), c2 = fp2_mul_sum<3>(x2, y2);
  r.c0 = c0; r.c1 = c1; r.c2 = c2;
}
// a * (b0 + b1 v): c0 = a0 b0 + (xi a2) b1, c1 = a0 b1 + a1 b0, c2 = a1 b1 + a2 b0
BN_DEV void fp6_mul_01_lazy(Fp6& r, const Fp6& a, const Fp2& b0, const Fp2& b1) {
  const Fp2 xa2 = fp2_norm(fp2_mul_xi(a.c2));
  const Fp2* const x0[2] = {&a.c0, &xa2};
  const Fp2* const y0[2] = {&b0, &b1};
  const Fp2* const x1[2] = {&a.c0, &a.c1};
  const Fp2* const y1[2] = {&b1, &b0};
  const Fp2* const x2[2] = {&a.c1, &a.c2};
  const Fp2* const y2[2] = {&b1, &b0};
  const Fp2 c0 = fp2_mul_sum<2>(x0, y0), c1 = fp2_mul_sum<2>(x1, y1), c2 = fp2_mul_sum<2>(x2, y2);
  r.c0 = c0; r.c1 = c1; r.c2 = c2;
}
#endif
BN_DEVN void fp6_inv(Fp6& r, const Fp6& a) {      // sites 10 .. 15
  Fp2 t0 = NS(10, fp2_sub(fp2_sqr(a.c0), fp2_mul_xi(fp2_mul(a.c1, a.c2))));
  Fp2 t1 = NS(11, fp2_sub(fp2_mul_xi(fp2_sqr(a.c2)), fp2_mul(a.c0, a.c1)));
  Fp2 t2 = NS(12, fp2_sub(fp2_sqr(a.c1), fp2_mul(a.c0, a.c2)));
  Fp2 d = fp2_add(fp2_mul_xi(NS(13, fp2_add(fp2_mul(a.c2, t1), fp2_mul(a.c1, t2)))), fp2_mul(a.c0, t0));
  d = NS(15, fp2_inv(NR(14, d)));
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
#if !defined(BN_TRIO_FORMULAS)
BN_DEV void fp12_mul_body(Fp12& r, const Fp12& a, const Fp12& b) {  // sites 20 .. 49
  // Order chosen for register pressure (values that live across the product calls must sit in the ~110 callee-saved
  // VGPRs): the Karatsuba product of the sums first, while nothing else is alive, then one Fq6 product at a time.
  // t0 = a0 b0 is parked in r.c0 (memory) while a1 b1 is computed: r may be a or b themselves (acc = acc * x), and by
  // then a.c0 / b.c0 have had their last use.
  // BN_RELOAD_FENCE between the three Fq6 products (opt-in, -DBN_USE_RELOAD_FENCE): a and b are memory operands (LDS
  // accumulator, private-segment slot); without the fence the compiler keeps the halves it loaded for the sums alive across
  // the first product and spills 28 words (reloading 42); with it each product re-reads the two halves it needs and the
  // multiplication has no spill at all — but the re-reads sit right in front of their use: measured 4.08 against 4.05 ms
  // per 65 536 (profiles/r03_b_ab_fe_machine.log), so it is off.
  Fp6 s, t, u;
  fp6_add(s, a.c0, a.c1); fp6_site_n<28>(s, s);
  fp6_add(t, b.c0, b.c1); fp6_site_n<31>(t, t);
  fp6_mul<34>(u, s, t);
  BN_RELOAD_FENCE();
  fp6_mul<20>(s, a.c0, b.c0);                      // t0
  fp6_sub(u, u, s);
  r.c0 = s;
  BN_RELOAD_FENCE();
  fp6_mul<24>(t, a.c1, b.c1);                      // t1
  fp6_sub(u, u, t);
  fp6_site_r<41>(r.c1, u);
  fp6_mul_v(s, t);
  fp6_add(s, r.c0, s);
  fp6_site_r<38>(r.c0, s);
}
BN_DEVN void fp12_mul(Fp12& r, const Fp12& a, const Fp12& b) { fp12_mul_body(r, a, b); }
BN_DEVF void fp12_mul_hot(Fp12& r, const Fp12& a, const Fp12& b) { fp12_mul_body(r, a, b); }     // the loop of fp12_pow_u
#endif
// ---- the "octet" layout for small batches: bn254_trio.hip ------------------------------------------------------------
// One verify is carried by the eight lanes (four lane pairs) of an octet.  A wave that has its SIMD to itself issues a
// multiply-add only every ~4.4 ns, so the latency of a small batch is instructions per LANE; the octet spreads them:
//   * an Fq12 product as its four Fq6 products, one per pair, recombined in two exchanges (fp12_kmul4);
//   * a cyclotomic squaring as three Fq4 squarings, each pair also forming the two outputs that come from its square;
//   * the Miller loop of a verify as rounds of four independent Fq2 products, one per pair (trio4; bn254_pairing.h:
//     miller_verify_rounds), everything linear replicated in the pairs.
// Values are exchanged through LDS.  BN_TRIO_FORMULAS selects the FORMULAS of that layout in any build — the host
// emulations use it to run the same arithmetic, bound tracker included, with the products of a group computed one after
// the other; BN_TRIO_DEVICE adds the lane-group machinery.
#if defined(BN_QUAD_DEVICE)
// ---- wave roles (bn254_quad.hip): the four lane pairs of a verify are the four WAVES of a workgroup; Fq2 values cross
// waves through LDS mailboxes between workgroup barriers.  A mailbox: [slot][lane of the wave][12 words] — nine limbs in
// three 16-byte-aligned accesses (b128, b128, b32); the stride of 12 words keeps the 16 lanes of a b128 pass on distinct banks.
}  // namespace bn254
extern __shared__ int32_t bn_trio_lds[];
namespace bn254 {
#define BN_QUAD_WG 256
#define BN_QUAD_STRIDE 12
#define BN_QUAD_SLOT_WORDS (BN_QUAD_STRIDE * 64)
static_assert(BN_LIMBS == 9, "mailbox accessors move 4 + 4 + 1 limbs");
typedef int32_t bn_i4 __attribute__((ext_vector_type(4)));
BN_DEV int quad_wave() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }     // the wave's role, 0..3 (scalar)
BN_DEV void qbox_put(int slot, const Fp2& x) {
  int32_t* at = bn_trio_lds + ((unsigned)slot * 64 + (threadIdx.x & 63u)) * BN_QUAD_STRIDE;
  const int32_t* v = x.c[0].v;
  *(bn_i4*)at = bn_i4{v[0], v[1], v[2], v[3]};
  *(bn_i4*)(at + 4) = bn_i4{v[4], v[5], v[6], v[7]};
  at[8] = v[8];
}
BN_DEV Fp2 qbox_get(int slot) {
  const int32_t* at = bn_trio_lds + ((unsigned)slot * 64 + (threadIdx.x & 63u)) * BN_QUAD_STRIDE;
  const bn_i4 lo = *(const bn_i4*)at, hi = *(const bn_i4*)(at + 4);
  Fp2 x;
  int32_t* v = x.c[0].v;
  v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w; v[8] = at[8];
  return x;
}
BN_DEV void qbox_put6(int slot, const Fp6& x) { qbox_put(slot, x.c0); qbox_put(slot + 1, x.c1); qbox_put(slot + 2, x.c2); }
BN_DEV void qbox_get6(Fp6& x, int slot) { x.c0 = qbox_get(slot); x.c1 = qbox_get(slot + 1); x.c2 = qbox_get(slot + 2); }
// workgroup barrier for the mailboxes: only LDS traffic has to be complete (lgkmcnt) — __syncthreads() also waits for every
// outstanding scratch / global store (vmcnt(0)), ~0.5 us per barrier here, several hundred barriers per verify
#define QUAD_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#endif
#if defined(BN_TRIO_DEVICE)
}  // namespace bn254
extern __shared__ int32_t bn_trio_lds[];          // dynamic LDS of the octet kernels: [accumulator slots | Fq6 exchange | Fq2 exchange]
namespace bn254 {
#define BN_TRIO_WG 256
#define BN_TRIO_F_WORDS (6 * BN_LIMBS + 1)                         // accumulator slot per lane (odd stride)
#define BN_TRIO_X6_OFF (BN_TRIO_WG * BN_TRIO_F_WORDS)              // Fq6 exchange: [octet][lane of the octet][27], octet stride 216
#define BN_TRIO_X6_STRIDE (8 * 3 * BN_LIMBS)
#define BN_TRIO_X2_OFF (BN_TRIO_X6_OFF + (BN_TRIO_WG / 8) * BN_TRIO_X6_STRIDE)   // Fq2 exchange: [octet][lane][9], octet stride 72
#define BN_TRIO_X2_STRIDE (8 * BN_LIMBS)
#define BN_TRIO_LDS_WORDS (BN_TRIO_X2_OFF + (BN_TRIO_WG / 8) * BN_TRIO_X2_STRIDE)
BN_DEV int trio_pair() { return (int)((threadIdx.x >> 1) & 3u); }                 // lane pair within the octet, 0..3
BN_DEV int trio_g3() { const int g = trio_pair(); return g == 3 ? 0 : g; }        // Fq6-level group: pair 3 shadows pair 0
// Exchanges: all eight lanes of an octet are in one wave and a wave's LDS instructions execute in order, so no barrier is
// involved; the wavefront-scope fences keep the compiler from moving the reads above the writes (or the next
// exchange's writes above these reads).  The array is indexed directly so that the accesses stay LDS instructions
// (ds_write / ds_read), batched under one wait.
#define BN_TRIO_FENCE() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
// every pair publishes one Fq2 value and reads those of the four pairs (same role)
BN_DEV void trio_share2x4(const Fp2& p, Fp2& r0, Fp2& r1, Fp2& r2, Fp2& r3) {
  const unsigned base = BN_TRIO_X2_OFF + (threadIdx.x >> 3) * BN_TRIO_X2_STRIDE, mine = base + (threadIdx.x & 7u) * BN_LIMBS;
  BN_TRIO_FENCE();
#pragma unroll
  for (int i = 0; i < BN_LIMBS; ++i) bn_trio_lds[mine + i] = p.c[0].v[i];
  BN_TRIO_FENCE();
  const unsigned role = threadIdx.x & 1u;
  Fp2* dst[4] = {&r0, &r1, &r2, &r3};
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const unsigned from = base + (2 * g + role) * BN_LIMBS;
#pragma unroll
    for (int i = 0; i < BN_LIMBS; ++i) dst[g]->c[0].v[i] = bn_trio_lds[from + i];
  }
  BN_TRIO_FENCE();
}
#endif
// One ROUND of the octet layout: four independent Fq2 products, one per lane pair, results exchanged so that every pair
// has all four afterwards (the twist-point steps and the line preparation are written as such rounds,
// bn254_pairing.h: miller_verify_rounds).  Elsewhere: the four products one after the other — same values.
BN_DEV void trio4(Fp2& r0, Fp2& r1, Fp2& r2, Fp2& r3, const Fp2& x0, const Fp2& y0, const Fp2& x1, const Fp2& y1, const Fp2& x2,
                  const Fp2& y2, const Fp2& x3, const Fp2& y3) {
#if defined(BN_TRIO_DEVICE)
  const int g = trio_pair();
  Fp2 x, y;
#pragma unroll
  for (int i = 0; i < BN_LIMBS; ++i) {
    x.c[0].v[i] = (g & 2) ? ((g & 1) ? x3.c[0].v[i] : x2.c[0].v[i]) : ((g & 1) ? x1.c[0].v[i] : x0.c[0].v[i]);
    y.c[0].v[i] = (g & 2) ? ((g & 1) ? y3.c[0].v[i] : y2.c[0].v[i]) : ((g & 1) ? y1.c[0].v[i] : y0.c[0].v[i]);
  }
  trio_share2x4(fp2_mul(x, y), r0, r1, r2, r3);
#else
  Fp2 a = fp2_mul(x0, y0), b = fp2_mul(x1, y1), c = fp2_mul(x2, y2), d = fp2_mul(x3, y3);
  r0 = a; r1 = b; r2 = c; r3 = d;
#endif
}
// The six Fq2 products of a Karatsuba Fq6 product as a list, so that a caller can place them in rounds: operand K of
// x (the same K of y is its partner), and the result from the six products (formulas and sites of fp6_mul<S>).
template <int K> BN_DEV Fp2 fp6_kop(const Fp6& x) {
  if constexpr (K == 0) return x.c0;
  else if constexpr (K == 1) return x.c1;
  else if constexpr (K == 2) return x.c2;
  else if constexpr (K == 3) return fp2_add(x.c1, x.c2);
  else if constexpr (K == 4) return fp2_add(x.c0, x.c1);
  else return fp2_add(x.c0, x.c2);
}
// coefficient K of that result from the four products it depends on: p0, p1, p2 and pk = p[3 + K]
template <int S, int K> BN_DEV Fp2 fp6_kfin_coef(const Fp2& p0, const Fp2& p1, const Fp2& p2, const Fp2& pk) {
  if constexpr (K == 0) return NS(S + 1, fp2_add(fp2_mul_xi(NS(S, fp2_sub(fp2_sub(pk, p1), p2))), p0));
  else if constexpr (K == 1) return NS(S + 2, fp2_add(fp2_sub(fp2_sub(pk, p0), p1), fp2_mul_xi(p2)));
  else return NS(S + 3, fp2_add(fp2_sub(fp2_sub(pk, p0), p2), p1));
}
template <int S> BN_DEV void fp6_kfin(Fp6& r, const Fp2 (&p)[6]) {   // p: x0y0, x1y1, x2y2, (x1+x2)(y1+y2), (x0+x1)(y0+y1), (x0+x2)(y0+y2)
  r.c0 = fp6_kfin_coef<S, 0>(p[0], p[1], p[2], p[3]);
  r.c1 = fp6_kfin_coef<S, 1>(p[0], p[1], p[2], p[4]);
  r.c2 = fp6_kfin_coef<S, 2>(p[0], p[1], p[2], p[5]);
}
// an Fq scalar as an Fq2 value (k + 0 i), so that a scaling by it can ride in a round as an ordinary Fq2 product
BN_DEV Fp2 fp2_from_fp(const Fp& k) { return fp2_make(k, fp_zero()); }

// r = a * b in Fq12 as the FOUR Fq6 products a0 b0, a1 b1, a0 b1, a1 b0 (no operand sums):
//   r0 = a0 b0 + v a1 b1,  r1 = a0 b1 + a1 b0.                                Sites S .. S+9.
// On the device of the octet layout lane pair g computes product g; pairs 0, 1 then form r0 and pairs 2, 3 form r1 from
// each other's products (one instruction stream: out = P_even + (r0 ? v P_odd : P_odd), carried and weakly reduced), and a
// second exchange hands both halves to every pair — each pair does a quarter of the products and half of the
// recombination.  Elsewhere: the four products one after the other, same values and sites.
template <int S> BN_DEV void fp12_kmul4(Fp12& r, const Fp12& a, const Fp12& b) {
#if defined(BN_TRIO_DEVICE)
  const int g = trio_pair();
  Fp6 x, y, p, pe, po;
  {
    const Fp2* ax[2][3] = {{&a.c0.c0, &a.c0.c1, &a.c0.c2}, {&a.c1.c0, &a.c1.c1, &a.c1.c2}};
    const Fp2* bx[2][3] = {{&b.c0.c0, &b.c0.c1, &b.c0.c2}, {&b.c1.c0, &b.c1.c1, &b.c1.c2}};
    Fp2* xs[3] = {&x.c0, &x.c1, &x.c2};
    Fp2* ys[3] = {&y.c0, &y.c1, &y.c2};
    const bool xa1 = (g & 1) != 0, yb1 = g == 1 || g == 2;               // pair 0: a0 b0, 1: a1 b1, 2: a0 b1, 3: a1 b0
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int i = 0; i < BN_LIMBS; ++i) {
        xs[c]->c[0].v[i] = xa1 ? ax[1][c]->c[0].v[i] : ax[0][c]->c[0].v[i];
        ys[c]->c[0].v[i] = yb1 ? bx[1][c]->c[0].v[i] : bx[0][c]->c[0].v[i];
      }
  }
  fp6_mul<S>(p, x, y);
  const unsigned base = BN_TRIO_X6_OFF + (threadIdx.x >> 3) * BN_TRIO_X6_STRIDE, mine = base + (threadIdx.x & 7u) * (3 * BN_LIMBS);
  const unsigned role = threadIdx.x & 1u;
  auto publish = [&](const Fp6& v) {
    const Fp2* src[3] = {&v.c0, &v.c1, &v.c2};
    BN_TRIO_FENCE();
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int i = 0; i < BN_LIMBS; ++i) bn_trio_lds[mine + c * BN_LIMBS + i] = src[c]->c[0].v[i];
    BN_TRIO_FENCE();
  };
  auto fetch = [&](Fp6& v, unsigned pair) {
    const unsigned from = base + (2 * pair + role) * (3 * BN_LIMBS);
    Fp2* d[3] = {&v.c0, &v.c1, &v.c2};
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int i = 0; i < BN_LIMBS; ++i) d[c]->c[0].v[i] = bn_trio_lds[from + c * BN_LIMBS + i];
  };
  publish(p);
  fetch(pe, (unsigned)(g & 2));
  fetch(po, (unsigned)(g | 1));
  {
    // v * po = (xi po2, po0, po1) for the r0 pairs, po itself for the r1 pairs
    const Fp2 t = fp2_mul_xi(po.c2);
    const bool lo = g < 2;
    Fp6 q;
#pragma unroll
    for (int i = 0; i < BN_LIMBS; ++i) {
      q.c0.c[0].v[i] = lo ? t.c[0].v[i] : po.c0.c[0].v[i];
      q.c1.c[0].v[i] = lo ? po.c0.c[0].v[i] : po.c1.c[0].v[i];
      q.c2.c[0].v[i] = lo ? po.c1.c[0].v[i] : po.c2.c[0].v[i];
    }
    fp6_add(q, pe, q);
    constexpr int m0 = BN_SITE_MODE(S + 4, 2) > BN_SITE_MODE(S + 7, 2) ? BN_SITE_MODE(S + 4, 2) : BN_SITE_MODE(S + 7, 2);
    constexpr int m1 = BN_SITE_MODE(S + 5, 2) > BN_SITE_MODE(S + 8, 2) ? BN_SITE_MODE(S + 5, 2) : BN_SITE_MODE(S + 8, 2);
    constexpr int m2 = BN_SITE_MODE(S + 6, 2) > BN_SITE_MODE(S + 9, 2) ? BN_SITE_MODE(S + 6, 2) : BN_SITE_MODE(S + 9, 2);
    p.c0 = fp2_site(q.c0, m0); p.c1 = fp2_site(q.c1, m1); p.c2 = fp2_site(q.c2, m2);
  }
  publish(p);
  fetch(pe, 0u);
  fetch(po, 2u);
  BN_TRIO_FENCE();
  r.c0 = pe; r.c1 = po;
#else
  Fp6 p0, p1, p2, p3, w;
  fp6_mul<S>(p0, a.c0, b.c0);
  fp6_mul<S>(p1, a.c1, b.c1);
  fp6_mul<S>(p2, a.c0, b.c1);
  fp6_mul<S>(p3, a.c1, b.c0);
  fp6_mul_v(w, p1);
  fp6_add(w, p0, w);
  fp6_add(p2, p2, p3);
  fp6_site_r<S + 4>(r.c0, w);
  fp6_site_r<S + 7>(r.c1, p2);
#endif
}
#if defined(BN_TRIO_FORMULAS)
BN_DEV void fp12_mul_body(Fp12& r, const Fp12& a, const Fp12& b) { fp12_kmul4<260>(r, a, b); }    // sites 260 .. 269
BN_DEVN void fp12_mul(Fp12& r, const Fp12& a, const Fp12& b) { fp12_mul_body(r, a, b); }
BN_DEVF void fp12_mul_hot(Fp12& r, const Fp12& a, const Fp12& b) { fp12_mul_body(r, a, b); }
#endif
#if defined(BN_SPLIT_FP2) && defined(BN_FP6_LAZY) && !defined(BN_FP6_LAZY_ONLY_LINE2)
BN_DEVH void fp12_sqr(Fp12& r, const Fp12& a) {                       // sites 300 .. 305
  Fp6 ab, s, t, u;
  fp6_mul_lazy(ab, a.c0, a.c1);
  fp6_add(s, a.c0, a.c1);                                             // lazy: the other operand is carried
  fp6_mul_v(t, a.c1);
  fp6_add(t, t, a.c0); fp6_norm(t, t);
  fp6_mul_lazy(u, s, t);
  fp6_sub(u, u, ab);
  fp6_mul_v(s, ab);
  fp6_sub(u, u, s);
  fp6_site_r<300>(r.c0, u);
  fp6_add(s, ab, ab);
  fp6_site_r<303>(r.c1, s);
}
#else
BN_DEVH void fp12_sqr(Fp12& r, const Fp12& a) {                       // sites 50 .. 79
  Fp6 ab, s, t, u;
  fp6_mul<50>(ab, a.c0, a.c1);
  fp6_add(s, a.c0, a.c1); fp6_site_n<54>(s, s);
  fp6_mul_v(t, a.c1);
  fp6_add(t, t, a.c0); fp6_site_n<57>(t, t);
  fp6_mul<60>(u, s, t);
  fp6_sub(u, u, ab);
  fp6_mul_v(s, ab);
  fp6_sub(u, u, s);
  fp6_site_r<64>(r.c0, u);
  fp6_add(s, ab, ab);
  fp6_site_r<67>(r.c1, s);
}
#endif
// the negated half keeps balanced digits balanced: no carry needed
BN_DEV void fp12_conj(Fp12& r, const Fp12& a) { r.c0 = a.c0; fp6_neg(r.c1, a.c1); }
BN_DEVN void fp12_inv(Fp12& r, const Fp12& a) {                       // sites 80 .. 109
  Fp6 t0, t1, d;
  fp6_mul<80>(t0, a.c0, a.c0);
  fp6_mul<84>(t1, a.c1, a.c1);
  fp6_mul_v(t1, t1);
  fp6_sub(d, t0, t1); fp6_site_r<88>(d, d);
  fp6_inv(d, d);
  fp6_mul<91>(t0, a.c1, d);
  fp6_mul<95>(r.c0, a.c0, d);
  fp6_neg(r.c1, t0);
}
// f * (l0 + (l1 + l2 v) w): the sparse shape of a D-twist line (l0 at w^0, l1 at w^1, l2 at w^3); l0, l1, l2 tight
#if defined(BN_INLINE_MUL_LINE)       // bn254_pair.hip: one copy per call site of the single-pair loops instead of a call with its operands in memory
BN_DEVH void fp12_mul_line(Fp12& r, const Fp12& f, const Fp2& l0, const Fp2& l1, const Fp2& l2) {   // sites 110 .. 139
#else
BN_DEVN void fp12_mul_line(Fp12& r, const Fp12& f, const Fp2& l0, const Fp2& l1, const Fp2& l2) {
#endif
  Fp6 t0, t1, s, u;
  fp6_mul_fp2(t0, f.c0, l0);
  fp6_mul_01<110>(t1, f.c1, l1, l2);
  fp6_add(s, f.c0, f.c1); fp6_site_n<113>(s, s);
  fp6_mul_01<116>(u, s, NS(119, fp2_add(l0, l1)), l2);
  fp6_sub(u, u, t0);
  fp6_sub(u, u, t1);
  fp6_mul_v(s, t1);
  fp6_add(s, t0, s);
  fp6_site_r<120>(r.c0, s);
  fp6_site_r<123>(r.c1, u);
}
// f * (b0 + b1 w): b0 a full Fq6, b1 = b10 + b11 v — the shape of a product of two lines; the b's tight
#if defined(BN_SPLIT_FP2) && defined(BN_FP6_LAZY) && !defined(BN_FP6_LAZY_ONLY_SQR)
BN_DEVH void fp12_mul_line2(Fp12& r, const Fp12& f, const Fp6& b0, const Fp2& b10, const Fp2& b11) {   // sites 306 .. 313
  Fp6 t0, t1, s, u, bs;
  fp6_mul_lazy(t0, f.c0, b0);
  fp6_mul_01_lazy(t1, f.c1, b10, b11);
  fp6_add(s, f.c0, f.c1);                                             // lazy: bs is carried
  bs.c0 = NS(306, fp2_add(b0.c0, b10)); bs.c1 = NS(307, fp2_add(b0.c1, b11)); bs.c2 = b0.c2;
  fp6_mul_lazy(u, s, bs);
  fp6_sub(u, u, t0);
  fp6_sub(u, u, t1);
  fp6_mul_v(s, t1);
  fp6_add(s, t0, s);
  fp6_site_r<308>(r.c0, s);
  fp6_site_r<311>(r.c1, u);
}
#else
BN_DEVH void fp12_mul_line2(Fp12& r, const Fp12& f, const Fp6& b0, const Fp2& b10, const Fp2& b11) {   // sites 140 .. 169
  Fp6 t0, t1, s, u, bs;
  fp6_mul<140>(t0, f.c0, b0);
  fp6_mul_01<144>(t1, f.c1, b10, b11);
  fp6_add(s, f.c0, f.c1); fp6_site_n<147>(s, s);
  bs.c0 = NS(150, fp2_add(b0.c0, b10)); bs.c1 = NS(151, fp2_add(b0.c1, b11)); bs.c2 = b0.c2;
  fp6_mul<152>(u, s, bs);
  fp6_sub(u, u, t0);
  fp6_sub(u, u, t1);
  fp6_mul_v(s, t1);
  fp6_add(s, t0, s);
  fp6_site_r<156>(r.c0, s);
  fp6_site_r<159>(r.c1, u);
}
#endif
// coefficient k of w^k in the polynomial basis: c[2i] = c0.c_i, c[2i+1] = c1.c_i
BN_DEV Fp2& fp12_coef(Fp12& a, int k) {
  Fp6& h = (k & 1) ? a.c1 : a.c0;
  return (k >> 1) == 0 ? h.c0 : (k >> 1) == 1 ? h.c1 : h.c2;
}
// q^power Frobenius, power in {1,2,3}
BN_DEV void fp12_frob_body(Fp12& r, const Fp12& a, int power) {
  Fp12 t = a;
  for (int k = 0; k < 6; ++k) {
    Fp2& c = fp12_coef(t, k);
    Fp2 x = (power & 1) ? fp2_conj(c) : c;
    const int32_t (*g)[BN_LIMBS] = power == 1 ? C_FROB1[k] : power == 2 ? C_FROB2[k] : C_FROB3[k];
    c = fp2_mul(x, fp2_load_const(g));
  }
  r = t;
}
BN_DEVN void fp12_frob(Fp12& r, const Fp12& a, int power) { fp12_frob_body(r, a, power); }
// (a + b s)^2 in Fq4 = Fq2[s]/(s^2 - xi): r0 = a^2 + xi b^2, r1 = 2ab; a, b tight.  Sites S .. S+2.
template <int S> BN_DEV void fp4_sqr(Fp2& r0, Fp2& r1, const Fp2& a, const Fp2& b) {
  Fp2 a2 = fp2_sqr(a), b2 = fp2_sqr(b);               // (the three leaves inlined here: no gain, profiles/r04_h_ab_inline_csqr_leaves.log)
  r1 = NS(S + 1, fp2_sub(fp2_sub(fp2_sqr(NS(S, fp2_add(a, b))), a2), b2));
  r0 = NS(S + 2, fp2_add(a2, fp2_mul_xi(b2)));
}
// Granger-Scott squaring for the cyclotomic subgroup (after the easy part of the final exp.).
// The outputs 3t -+ 2a are linear in a and xi-fold in the squares, so every output is weakly reduced
// (fp2_lin2_reduce): across a run of squarings the values stay below ~0.6 q.
template <int S> BN_DEV void fp12_cyclotomic_sqr_body(Fp12& r, const Fp12& a) {       // sites S .. S+15 (170.., 240..)
#if defined(BN_TRIO_DEVICE)
  // octet layout: lane pair g = 0, 1, 2 squares one Fq4 element (sites S+20 .. S+22: the safe defaults, this instance
  // stands for all three of the serial form) AND forms the two outputs that come from it — 3 t -+ 2 a with the weak
  // reduction — before the exchange, so that this linear work is done once per output instead of in every pair:
  //   pair 0: (t0, t1) = (a00 + a11 s)^2  ->  r00 = 3 t0 - 2 a00,  r11 = 3 t1 + 2 a11
  //   pair 1: (t2, t3) = (a10 + a02 s)^2  ->  r01 = 3 t2 - 2 a01,  r12 = 3 t3 + 2 a12
  //   pair 2: (t4, t5) = (a01 + a12 s)^2  ->  r02 = 3 t4 - 2 a02,  r10 = 3 xi t5 + 2 a10
  const int g = trio_g3();
  Fp2 x, y, ae, ao, e, o;
#pragma unroll
  for (int i = 0; i < BN_LIMBS; ++i) {
    x.c[0].v[i] = g == 0 ? a.c0.c0.c[0].v[i] : g == 1 ? a.c1.c0.c[0].v[i] : a.c0.c1.c[0].v[i];
    y.c[0].v[i] = g == 0 ? a.c1.c1.c[0].v[i] : g == 1 ? a.c0.c2.c[0].v[i] : a.c1.c2.c[0].v[i];
    ae.c[0].v[i] = g == 0 ? a.c0.c0.c[0].v[i] : g == 1 ? a.c0.c1.c[0].v[i] : a.c0.c2.c[0].v[i];
    ao.c[0].v[i] = g == 0 ? a.c1.c1.c[0].v[i] : g == 1 ? a.c1.c2.c[0].v[i] : a.c1.c0.c[0].v[i];
  }
  fp4_sqr<S + 20>(e, o, x, y);
  const Fp2 oxi = NS(S + 9, fp2_mul_xi(o));
#pragma unroll
  for (int i = 0; i < BN_LIMBS; ++i) o.c[0].v[i] = g == 2 ? oxi.c[0].v[i] : o.c[0].v[i];
  e = fp2_lin2_reduce(e, 3, ae, -2);
  o = fp2_lin2_reduce(o, 3, ao, 2);
  const unsigned base = BN_TRIO_X6_OFF + (threadIdx.x >> 3) * BN_TRIO_X6_STRIDE, mine = base + (threadIdx.x & 7u) * (3 * BN_LIMBS);
  BN_TRIO_FENCE();
#pragma unroll
  for (int i = 0; i < BN_LIMBS; ++i) { bn_trio_lds[mine + i] = e.c[0].v[i]; bn_trio_lds[mine + BN_LIMBS + i] = o.c[0].v[i]; }
  BN_TRIO_FENCE();
  const unsigned role = threadIdx.x & 1u;
  Fp12 out;
  Fp2* dst[6] = {&out.c0.c0, &out.c1.c1, &out.c0.c1, &out.c1.c2, &out.c0.c2, &out.c1.c0};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const unsigned from = base + (2 * k + role) * (3 * BN_LIMBS);
#pragma unroll
    for (int i = 0; i < BN_LIMBS; ++i) { dst[2 * k]->c[0].v[i] = bn_trio_lds[from + i]; dst[2 * k + 1]->c[0].v[i] = bn_trio_lds[from + BN_LIMBS + i]; }
  }
  BN_TRIO_FENCE();
  r = out;
#else
  Fp2 t0, t1, t2, t3, t4, t5;
  fp4_sqr<S>(t0, t1, a.c0.c0, a.c1.c1);
  fp4_sqr<S + 3>(t2, t3, a.c1.c0, a.c0.c2);
  fp4_sqr<S + 6>(t4, t5, a.c0.c1, a.c1.c2);
  Fp12 o;
  // outputs 3 t -+ 2 a: one fused pass each (combination, carry and weak reduction), so the t's may stay lazy
  o.c0.c0 = fp2_lin2_reduce(t0, 3, a.c0.c0, -2);
  o.c1.c1 = fp2_lin2_reduce(t1, 3, a.c1.c1, 2);
  t5 = NS(S + 9, fp2_mul_xi(t5));
  o.c1.c0 = fp2_lin2_reduce(t5, 3, a.c1.c0, 2);
  o.c0.c2 = fp2_lin2_reduce(t4, 3, a.c0.c2, -2);
  o.c0.c1 = fp2_lin2_reduce(t2, 3, a.c0.c1, -2);
  o.c1.c2 = fp2_lin2_reduce(t3, 3, a.c1.c2, 2);
  r = o;
#endif
}
BN_DEVN void fp12_cyclotomic_sqr(Fp12& r, const Fp12& a) { fp12_cyclotomic_sqr_body<170>(r, a); }
BN_DEVF void fp12_cyclotomic_sqr_hot(Fp12& r, const Fp12& a) { fp12_cyclotomic_sqr_body<170>(r, a); }      // the loop of fp12_pow_u

}  // namespace bn254
