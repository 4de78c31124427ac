// BN254 base field Fq and the tower Fq2 / Fq6 / Fq12 for gfx950 lanes.
//
// One field element per lane: 8 x 32-bit limbs in VGPRs, Montgomery form (R = 2^256), always
// fully reduced to [0, q).  The hot primitive is the 8x8 CIOS Montgomery product built from
// v_mad_u64_u32 (measured on MI355X: ~5 cycles per wave64 instruction per SIMD — the same
// issue cost as v_mul_lo/hi_u32 or v_fma_f64, so one instruction per 32x32->64 MAC is the
// best this ISA offers; no MFMA — this is carry-chain integer work).
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
#else
#define BN_DEV static inline __attribute__((always_inline))
#define BN_DEVN static __attribute__((noinline))
#define BN_CONST static const
#endif

#include "bn254_constants.h"

// host-only instrumentation (tests/hostsim): exact count of Montgomery products per kernel stage,
// the algorithmic-work figure behind bench.py's roofline (1 product = 136 MAC32)
#if defined(BN_COUNT_FP_MUL) && !defined(__HIPCC__)
extern "C" unsigned long long bn_fp_mul_counter;
#define BN_COUNT_MUL() (++bn_fp_mul_counter)
#else
#define BN_COUNT_MUL()
#endif

namespace bn254 {

#define BN_Q_ARRAY {BN_Q0, BN_Q1, BN_Q2, BN_Q3, BN_Q4, BN_Q5, BN_Q6, BN_Q7}

struct Fp { uint32_t v[8]; };
struct Fp2 { Fp c0, c1; };
struct Fp6 { Fp2 c0, c1, c2; };
struct Fp12 { Fp6 c0, c1; };

// ------------------------------------------------------------------------------------------
// Fq
// ------------------------------------------------------------------------------------------
BN_DEV Fp fp_load_const(const uint32_t* c) {
  Fp r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = c[i];
  return r;
}
BN_DEV Fp fp_zero() {
  Fp r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = 0;
  return r;
}
BN_DEV Fp fp_one() { return fp_load_const(C_ONE); }
BN_DEV bool fp_is_zero(const Fp& a) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) o |= a.v[i];
  return o == 0;
}
BN_DEV bool fp_eq(const Fp& a, const Fp& b) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) o |= a.v[i] ^ b.v[i];
  return o == 0;
}
// r = c ? a : b
BN_DEV Fp fp_select(bool c, const Fp& a, const Fp& b) {
  Fp r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = c ? a.v[i] : b.v[i];
  return r;
}
// a >= b as 256-bit integers (plain limbs)
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
  const uint32_t q[8] = BN_Q_ARRAY;
  uint32_t s[8], d[8];
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    uint64_t x = (uint64_t)a.v[i] + b.v[i] + c;
    s[i] = (uint32_t)x; c = (uint32_t)(x >> 32);
  }
  uint32_t bw = 0;   // a + b < 2q < 2^255: no carry out of the top limb
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    uint64_t x = (uint64_t)s[i] - q[i] - bw;
    d[i] = (uint32_t)x; bw = (uint32_t)(x >> 63);
  }
  Fp r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = bw ? s[i] : d[i];
  return r;
}
BN_DEV Fp fp_sub(const Fp& a, const Fp& b) {
  const uint32_t q[8] = BN_Q_ARRAY;
  uint32_t d[8];
  uint32_t bw = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    uint64_t x = (uint64_t)a.v[i] - b.v[i] - bw;
    d[i] = (uint32_t)x; bw = (uint32_t)(x >> 63);
  }
  uint32_t mask = 0u - bw;   // borrow -> add q back
  uint32_t c = 0;
  Fp r;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    uint64_t x = (uint64_t)d[i] + (q[i] & mask) + c;
    r.v[i] = (uint32_t)x; c = (uint32_t)(x >> 32);
  }
  return r;
}
BN_DEV Fp fp_neg(const Fp& a) { return fp_sub(fp_zero(), a); }
BN_DEV Fp fp_dbl(const Fp& a) { return fp_add(a, a); }

// Montgomery product a*b*R^-1 mod q, CIOS over 8 x 32-bit limbs.
BN_DEVN Fp fp_mul(Fp a, Fp b) {
  BN_COUNT_MUL();
  const uint32_t q[8] = BN_Q_ARRAY;
  uint32_t t[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) t[i] = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    uint32_t c = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      uint64_t uv = (uint64_t)a.v[j] * b.v[i] + t[j] + c;
      t[j] = (uint32_t)uv; c = (uint32_t)(uv >> 32);
    }
    uint64_t s = (uint64_t)t[8] + c;
    t[8] = (uint32_t)s; t[9] = (uint32_t)(s >> 32);
    uint32_t m = t[0] * BN_N0;
    uint64_t uv = (uint64_t)m * q[0] + t[0];
    c = (uint32_t)(uv >> 32);
#pragma unroll
    for (int j = 1; j < 8; ++j) {
      uv = (uint64_t)m * q[j] + t[j] + c;
      t[j - 1] = (uint32_t)uv; c = (uint32_t)(uv >> 32);
    }
    s = (uint64_t)t[8] + c;
    t[7] = (uint32_t)s; t[8] = t[9] + (uint32_t)(s >> 32);
  }
  // result < 2q < 2^255 (t[8] == 0): one conditional subtraction
  uint32_t d[8];
  uint32_t bw = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    uint64_t x = (uint64_t)t[i] - q[i] - bw;
    d[i] = (uint32_t)x; bw = (uint32_t)(x >> 63);
  }
  Fp r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = bw ? t[i] : d[i];
  return r;
}
BN_DEV Fp fp_sqr(const Fp& a) { return fp_mul(a, a); }

BN_DEV Fp fp_to_mont(const Fp& plain) { return fp_mul(plain, fp_load_const(C_R2)); }
BN_DEV Fp fp_from_mont(const Fp& a) {
  Fp one = fp_zero();
  one.v[0] = 1;
  return fp_mul(a, one);
}
// a^e for a fixed public exponent (plain limbs in constant memory); wave-uniform control flow
BN_DEVN Fp fp_pow_const(Fp a, const uint32_t* e) {
  Fp acc = fp_one();
  for (int i = 255; i >= 0; --i) {
    acc = fp_sqr(acc);
    if ((e[i >> 5] >> (i & 31)) & 1) acc = fp_mul(acc, a);
  }
  return acc;
}
BN_DEV Fp fp_inv(const Fp& a) { return fp_pow_const(a, C_EXP_QM2); }   // Fermat; inv(0) = 0
// y = a^((q+1)/4) (q = 3 mod 4); returns true iff y^2 == a
BN_DEV bool fp_sqrt(Fp& y, const Fp& a) {
  y = fp_pow_const(a, C_EXP_QP1D4);
  return fp_eq(fp_sqr(y), a);
}

// ------------------------------------------------------------------------------------------
// Fq2
// ------------------------------------------------------------------------------------------
BN_DEV Fp2 fp2_zero() { Fp2 r; r.c0 = fp_zero(); r.c1 = fp_zero(); return r; }
BN_DEV Fp2 fp2_one() { Fp2 r; r.c0 = fp_one(); r.c1 = fp_zero(); return r; }
BN_DEV Fp2 fp2_load_const(const uint32_t (*c)[8]) { Fp2 r; r.c0 = fp_load_const(c[0]); r.c1 = fp_load_const(c[1]); return r; }
BN_DEV Fp2 fp2_add(const Fp2& a, const Fp2& b) { Fp2 r; r.c0 = fp_add(a.c0, b.c0); r.c1 = fp_add(a.c1, b.c1); return r; }
BN_DEV Fp2 fp2_sub(const Fp2& a, const Fp2& b) { Fp2 r; r.c0 = fp_sub(a.c0, b.c0); r.c1 = fp_sub(a.c1, b.c1); return r; }
BN_DEV Fp2 fp2_neg(const Fp2& a) { Fp2 r; r.c0 = fp_neg(a.c0); r.c1 = fp_neg(a.c1); return r; }
BN_DEV Fp2 fp2_dbl(const Fp2& a) { return fp2_add(a, a); }
BN_DEV Fp2 fp2_conj(const Fp2& a) { Fp2 r; r.c0 = a.c0; r.c1 = fp_neg(a.c1); return r; }
BN_DEV bool fp2_is_zero(const Fp2& a) { return fp_is_zero(a.c0) && fp_is_zero(a.c1); }
BN_DEV bool fp2_eq(const Fp2& a, const Fp2& b) { return fp_eq(a.c0, b.c0) && fp_eq(a.c1, b.c1); }
BN_DEV Fp2 fp2_select(bool c, const Fp2& a, const Fp2& b) { Fp2 r; r.c0 = fp_select(c, a.c0, b.c0); r.c1 = fp_select(c, a.c1, b.c1); return r; }
BN_DEV Fp2 fp2_mul(const Fp2& a, const Fp2& b) {   // Karatsuba, 3 Fq products
  Fp t0 = fp_mul(a.c0, b.c0), t1 = fp_mul(a.c1, b.c1);
  Fp t2 = fp_mul(fp_add(a.c0, a.c1), fp_add(b.c0, b.c1));
  Fp2 r;
  r.c0 = fp_sub(t0, t1);
  r.c1 = fp_sub(fp_sub(t2, t0), t1);
  return r;
}
BN_DEV Fp2 fp2_sqr(const Fp2& a) {                 // 2 Fq products
  Fp m = fp_mul(a.c0, a.c1);
  Fp2 r;
  r.c0 = fp_mul(fp_add(a.c0, a.c1), fp_sub(a.c0, a.c1));
  r.c1 = fp_dbl(m);
  return r;
}
BN_DEV Fp2 fp2_mul_fp(const Fp2& a, const Fp& k) { Fp2 r; r.c0 = fp_mul(a.c0, k); r.c1 = fp_mul(a.c1, k); return r; }
BN_DEV Fp2 fp2_mul_xi(const Fp2& a) {              // (9 + i) * a
  Fp a2 = fp_dbl(a.c0), a4 = fp_dbl(a2), a8 = fp_dbl(a4);
  Fp b2 = fp_dbl(a.c1), b4 = fp_dbl(b2), b8 = fp_dbl(b4);
  Fp2 r;
  r.c0 = fp_sub(fp_add(a8, a.c0), a.c1);
  r.c1 = fp_add(fp_add(b8, a.c1), a.c0);
  return r;
}
BN_DEV Fp2 fp2_inv(const Fp2& a) {
  Fp n = fp_inv(fp_add(fp_sqr(a.c0), fp_sqr(a.c1)));
  Fp2 r;
  r.c0 = fp_mul(a.c0, n);
  r.c1 = fp_neg(fp_mul(a.c1, n));
  return r;
}

// ------------------------------------------------------------------------------------------
// Fq6, Fq12 — operate on memory (the per-lane private segment): real functions
// ------------------------------------------------------------------------------------------
BN_DEV void fp6_add(Fp6& r, const Fp6& a, const Fp6& b) { r.c0 = fp2_add(a.c0, b.c0); r.c1 = fp2_add(a.c1, b.c1); r.c2 = fp2_add(a.c2, b.c2); }
BN_DEV void fp6_sub(Fp6& r, const Fp6& a, const Fp6& b) { r.c0 = fp2_sub(a.c0, b.c0); r.c1 = fp2_sub(a.c1, b.c1); r.c2 = fp2_sub(a.c2, b.c2); }
BN_DEV void fp6_neg(Fp6& r, const Fp6& a) { r.c0 = fp2_neg(a.c0); r.c1 = fp2_neg(a.c1); r.c2 = fp2_neg(a.c2); }
BN_DEV void fp6_mul_v(Fp6& r, const Fp6& a) { Fp2 t = fp2_mul_xi(a.c2); r.c2 = a.c1; r.c1 = a.c0; r.c0 = t; }

BN_DEVN void fp6_mul(Fp6& r, const Fp6& a, const Fp6& b) {
  Fp2 v0 = fp2_mul(a.c0, b.c0), v1 = fp2_mul(a.c1, b.c1), v2 = fp2_mul(a.c2, b.c2);
  Fp2 c0 = fp2_add(fp2_mul_xi(fp2_sub(fp2_sub(fp2_mul(fp2_add(a.c1, a.c2), fp2_add(b.c1, b.c2)), v1), v2)), v0);
  Fp2 c1 = fp2_add(fp2_sub(fp2_sub(fp2_mul(fp2_add(a.c0, a.c1), fp2_add(b.c0, b.c1)), v0), v1), fp2_mul_xi(v2));
  Fp2 c2 = fp2_add(fp2_sub(fp2_sub(fp2_mul(fp2_add(a.c0, a.c2), fp2_add(b.c0, b.c2)), v0), v2), v1);
  r.c0 = c0; r.c1 = c1; r.c2 = c2;
}
BN_DEVN void fp6_mul_fp2(Fp6& r, const Fp6& a, const Fp2& k) {
  Fp2 c0 = fp2_mul(a.c0, k), c1 = fp2_mul(a.c1, k), c2 = fp2_mul(a.c2, k);
  r.c0 = c0; r.c1 = c1; r.c2 = c2;
}
// a * (b0 + b1 v)
BN_DEVN void fp6_mul_01(Fp6& r, const Fp6& a, const Fp2& b0, const Fp2& b1) {
  Fp2 v0 = fp2_mul(a.c0, b0), v1 = fp2_mul(a.c1, b1);
  Fp2 c0 = fp2_add(fp2_mul_xi(fp2_mul(a.c2, b1)), v0);
  Fp2 c1 = fp2_sub(fp2_sub(fp2_mul(fp2_add(a.c0, a.c1), fp2_add(b0, b1)), v0), v1);
  Fp2 c2 = fp2_add(fp2_mul(a.c2, b0), v1);
  r.c0 = c0; r.c1 = c1; r.c2 = c2;
}
BN_DEVN void fp6_inv(Fp6& r, const Fp6& a) {
  Fp2 t0 = fp2_sub(fp2_sqr(a.c0), fp2_mul_xi(fp2_mul(a.c1, a.c2)));
  Fp2 t1 = fp2_sub(fp2_mul_xi(fp2_sqr(a.c2)), fp2_mul(a.c0, a.c1));
  Fp2 t2 = fp2_sub(fp2_sqr(a.c1), fp2_mul(a.c0, a.c2));
  Fp2 d = fp2_add(fp2_mul_xi(fp2_add(fp2_mul(a.c2, t1), fp2_mul(a.c1, t2))), fp2_mul(a.c0, t0));
  d = fp2_inv(d);
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
  fp6_add(s, a.c0, a.c1);
  fp6_add(t, b.c0, b.c1);
  fp6_mul(u, s, t);
  fp6_sub(u, u, t0);
  fp6_sub(u, u, t1);
  fp6_mul_v(s, t1);
  fp6_add(r.c0, t0, s);
  r.c1 = u;
}
BN_DEVN void fp12_sqr(Fp12& r, const Fp12& a) {
  Fp6 ab, s, t, u;
  fp6_mul(ab, a.c0, a.c1);
  fp6_add(s, a.c0, a.c1);
  fp6_mul_v(t, a.c1);
  fp6_add(t, t, a.c0);
  fp6_mul(u, s, t);
  fp6_sub(u, u, ab);
  fp6_mul_v(s, ab);
  fp6_sub(r.c0, u, s);
  fp6_add(r.c1, ab, ab);
}
BN_DEV void fp12_conj(Fp12& r, const Fp12& a) { r.c0 = a.c0; fp6_neg(r.c1, a.c1); }
BN_DEVN void fp12_inv(Fp12& r, const Fp12& a) {
  Fp6 t0, t1, d;
  fp6_mul(t0, a.c0, a.c0);
  fp6_mul(t1, a.c1, a.c1);
  fp6_mul_v(t1, t1);
  fp6_sub(d, t0, t1);
  fp6_inv(d, d);
  fp6_mul(t0, a.c1, d);
  fp6_mul(r.c0, a.c0, d);
  fp6_neg(r.c1, t0);
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
  fp6_add(r.c0, t0, s);
  r.c1 = u;
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
    const uint32_t (*g)[8] = power == 1 ? C_FROB1[k] : power == 2 ? C_FROB2[k] : C_FROB3[k];
    c = fp2_mul(x, fp2_load_const(g));
  }
  r = t;
}
// (a + b s)^2 in Fq4 = Fq2[s]/(s^2 - xi): r0 = a^2 + xi b^2, r1 = 2ab
BN_DEV void fp4_sqr(Fp2& r0, Fp2& r1, const Fp2& a, const Fp2& b) {
  Fp2 a2 = fp2_sqr(a), b2 = fp2_sqr(b);
  r1 = fp2_sub(fp2_sub(fp2_sqr(fp2_add(a, b)), a2), b2);
  r0 = fp2_add(a2, fp2_mul_xi(b2));
}
// Granger-Scott squaring for the cyclotomic subgroup (after the easy part of the final exp.)
BN_DEVN void fp12_cyclotomic_sqr(Fp12& r, const Fp12& a) {
  Fp2 t0, t1, t2, t3, t4, t5;
  fp4_sqr(t0, t1, a.c0.c0, a.c1.c1);
  fp4_sqr(t2, t3, a.c1.c0, a.c0.c2);
  fp4_sqr(t4, t5, a.c0.c1, a.c1.c2);
  Fp12 o;
  o.c0.c0 = fp2_add(fp2_dbl(fp2_sub(t0, a.c0.c0)), t0);
  o.c1.c1 = fp2_add(fp2_dbl(fp2_add(t1, a.c1.c1)), t1);
  t5 = fp2_mul_xi(t5);
  o.c1.c0 = fp2_add(fp2_dbl(fp2_add(t5, a.c1.c0)), t5);
  o.c0.c2 = fp2_add(fp2_dbl(fp2_sub(t4, a.c0.c2)), t4);
  o.c0.c1 = fp2_add(fp2_dbl(fp2_sub(t2, a.c0.c1)), t2);
  o.c1.c2 = fp2_add(fp2_dbl(fp2_add(t3, a.c1.c2)), t3);
  r = o;
}

}  // namespace bn254
