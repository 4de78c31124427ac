/*
 * bn254_oracle.c — CPU restatement of the sedaprotocol/bn254 verify path, in plain C.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this library; the product (libbn254hip.so, bn254_amd/) never links, loads or
 * calls it.
 *
 * Parity status: PINNED.  The reference's arithmetic lives in the third-party crate
 * `zeropool-bn 0.5.11` (/root/reference/Cargo.toml:24), which is not vendored and cannot be
 * built here (no rustc/cargo).  This file therefore restates the published BN254 /
 * alt_bn128 algorithm (EIP-196/197 curve, optimal-ate pairing, tower Fq2/Fq6/Fq12) and is
 * anchored on (i) every known-answer vector in the reference's tests
 * (tests/golden/reference_kats.json, checked by tests/test_oracle_c.py) and (ii) the
 * independent big-integer model oracle/bn254_model.py (different representation, affine
 * formulas, naive final exponentiation), with which it must agree bit for bit.
 *
 * Structure mirrors what the reference executes:
 *   ECDSA::verify            /root/reference/src/ecdsa.rs:49-64   -> bn254o_verify
 *   hash_to_try_and_increment /root/reference/src/hash.rs:29-63   -> bn254o_hash_to_g1
 *   mod_u256                  /root/reference/src/utils.rs:27-37  -> inside hash (strict '>')
 *   arbitrary_string_to_g1    /root/reference/src/utils.rs:56-63  -> g1_decompress_even
 *   from_uncompressed_to_g1/2 /root/reference/src/utils.rs:107-127 -> decode_g1 / decode_g2
 *   bn::pairing_batch         call sites ecdsa.rs:57,86           -> miller_loop_multi + final_exp
 *   Add for PublicKey/Signature /root/reference/src/types.rs:126-132,264-270 -> bn254o_g1_add/g2_add
 *   G1*Fr, G2*Fr              ecdsa.rs:31, types.rs:86,156        -> bn254o_g1_mul/g2_mul
 *
 * Representation: Fq in Montgomery form, 4 x 64-bit limbs, R = 2^256, portable C with
 * unsigned __int128 (the same class of code as substrate-bn's u128 limbs; no asm, no SIMD).
 * All derived constants (R^2, -q^-1, Frobenius coefficients, 3/xi) are computed at start-up
 * from q and xi alone — nothing is shared with the product's generated tables.
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef uint64_t u64;
typedef uint32_t u32;
typedef uint8_t u8;

/* status codes: 0 = Ok, else 1 + index of the variant in /root/reference/src/error.rs:6-29 */
enum {
  ST_OK = 0, ST_HASH_TO_POINT = 1, ST_INDEX_OOB = 2, ST_INVALID_ENCODING = 3, ST_INVALID_GROUP_POINT = 4,
  ST_INVALID_LENGTH = 5, ST_NOT_MEMBER = 6, ST_TO_AFFINE = 7, ST_POINT_IN_JACOBIAN = 8,
  ST_VERIFICATION_FAILED = 9, ST_SERIALIZATION = 10, ST_HEX_DECODE = 11
};
#define FLAG_G2_SUBGROUP_CHECK 1u
#define FLAG_REJECT_IDENTITY 2u

/* ------------------------------------------------------------------------------------------ */
/* Fq                                                                                         */
/* ------------------------------------------------------------------------------------------ */
typedef struct { u64 l[4]; } fp;

static const fp FP_Q = {{0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL}};
static u64 N0INV;        /* -q^-1 mod 2^64 */
static fp FP_R2;         /* R^2 mod q */
static fp FP_ONE;        /* R mod q */
static fp FP_ZERO;

static __thread u64 g_fp_mul_count; /* instrumented: Montgomery multiplications (incl. squarings) */

static int fp_is_zero(const fp *a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }
static int fp_eq(const fp *a, const fp *b) {
  return ((a->l[0] ^ b->l[0]) | (a->l[1] ^ b->l[1]) | (a->l[2] ^ b->l[2]) | (a->l[3] ^ b->l[3])) == 0;
}
/* a >= b as 256-bit integers */
static int u256_geq(const u64 *a, const u64 *b) {
  for (int i = 3; i >= 0; --i) { if (a[i] != b[i]) return a[i] > b[i]; }
  return 1;
}
static u64 u256_sub(u64 *r, const u64 *a, const u64 *b) {
  u64 borrow = 0;
  for (int i = 0; i < 4; ++i) {
    u128 d = (u128)a[i] - b[i] - borrow;
    r[i] = (u64)d; borrow = (u64)(d >> 64) & 1;
  }
  return borrow;
}
static u64 u256_add(u64 *r, const u64 *a, const u64 *b) {
  u64 carry = 0;
  for (int i = 0; i < 4; ++i) {
    u128 s = (u128)a[i] + b[i] + carry;
    r[i] = (u64)s; carry = (u64)(s >> 64);
  }
  return carry;
}
static void fp_add(fp *r, const fp *a, const fp *b) {
  u64 t[4]; u64 c = u256_add(t, a->l, b->l);
  if (c || u256_geq(t, FP_Q.l)) u256_sub(r->l, t, FP_Q.l); else memcpy(r->l, t, 32);
}
static void fp_sub(fp *r, const fp *a, const fp *b) {
  u64 t[4];
  if (u256_sub(t, a->l, b->l)) u256_add(r->l, t, FP_Q.l); else memcpy(r->l, t, 32);
}
static void fp_neg(fp *r, const fp *a) {
  if (fp_is_zero(a)) *r = *a; else u256_sub(r->l, FP_Q.l, a->l);
}
static void fp_dbl(fp *r, const fp *a) { fp_add(r, a, a); }

/* Montgomery product a*b*R^-1 mod q (CIOS) */
static void fp_mul(fp *r, const fp *a, const fp *b) {
  u64 t[6] = {0, 0, 0, 0, 0, 0};
  ++g_fp_mul_count;
  for (int i = 0; i < 4; ++i) {
    u128 c = 0;
    for (int j = 0; j < 4; ++j) {
      c += (u128)a->l[j] * b->l[i] + t[j];
      t[j] = (u64)c; c >>= 64;
    }
    c += t[4]; t[4] = (u64)c; t[5] = (u64)(c >> 64);
    u64 m = t[0] * N0INV;
    c = (u128)m * FP_Q.l[0] + t[0]; c >>= 64;
    for (int j = 1; j < 4; ++j) {
      c += (u128)m * FP_Q.l[j] + t[j];
      t[j - 1] = (u64)c; c >>= 64;
    }
    c += t[4]; t[3] = (u64)c; t[4] = t[5] + (u64)(c >> 64);
  }
  if (t[4] || u256_geq(t, FP_Q.l)) u256_sub(r->l, t, FP_Q.l); else memcpy(r->l, t, 32);
}
static void fp_sqr(fp *r, const fp *a) { fp_mul(r, a, a); }

static void fp_from_u256(fp *r, const u64 *v) { fp t; memcpy(t.l, v, 32); fp_mul(r, &t, &FP_R2); }
static void fp_to_u256(u64 *v, const fp *a) { fp one = {{1, 0, 0, 0}}, t; fp_mul(&t, a, &one); memcpy(v, t.l, 32); }

/* a^e, e a 256-bit little-endian limb exponent */
static void fp_pow(fp *r, const fp *a, const u64 *e) {
  fp acc = FP_ONE, base = *a;
  int started = 0;
  for (int i = 255; i >= 0; --i) {
    if (started) fp_sqr(&acc, &acc);
    if ((e[i >> 6] >> (i & 63)) & 1) { if (started) fp_mul(&acc, &acc, &base); else { acc = base; started = 1; } }
  }
  *r = acc;
}
static u64 EXP_QM2[4], EXP_QP1D4[4];
static void fp_inv(fp *r, const fp *a) { fp_pow(r, a, EXP_QM2); }          /* Fermat; 0 -> 0 */
/* square root for q = 3 mod 4: a^((q+1)/4); returns 1 if a is a square */
static int fp_sqrt(fp *r, const fp *a) {
  fp y, y2; fp_pow(&y, a, EXP_QP1D4); fp_sqr(&y2, &y);
  *r = y; return fp_eq(&y2, a);
}
static void be32_to_u256(u64 *v, const u8 *b) {
  for (int i = 0; i < 4; ++i) {
    u64 w = 0;
    for (int j = 0; j < 8; ++j) w = (w << 8) | b[(3 - i) * 8 + j];
    v[i] = w;
  }
}
static void u256_to_be32(u8 *b, const u64 *v) {
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) b[(3 - i) * 8 + j] = (u8)(v[i] >> (56 - 8 * j));
}
/* Fq::from_slice: big-endian, must be < q */
static int fp_from_be(fp *r, const u8 *b) {
  u64 v[4]; be32_to_u256(v, b);
  if (u256_geq(v, FP_Q.l)) return 0;
  fp_from_u256(r, v); return 1;
}
static void fp_to_be(u8 *b, const fp *a) { u64 v[4]; fp_to_u256(v, a); u256_to_be32(b, v); }
static void fp_set_u64(fp *r, u64 x) { u64 v[4] = {x, 0, 0, 0}; fp_from_u256(r, v); }

/* ------------------------------------------------------------------------------------------ */
/* Fq2 = Fq[i]/(i^2+1)                                                                        */
/* ------------------------------------------------------------------------------------------ */
typedef struct { fp c0, c1; } fp2;
static fp2 FP2_ZERO, FP2_ONE, TWIST_B, TWIST_3B;
static fp2 FROB1[6], FROB2[6], FROB3[6];   /* xi^(k(q^j-1)/6), k = 0..5 */
static fp2 TW_FROB_X1, TW_FROB_Y1, TW_FROB_X2; /* xi^((q-1)/3), xi^((q-1)/2), xi^((q^2-1)/3) */

static void fp2_add(fp2 *r, const fp2 *a, const fp2 *b) { fp_add(&r->c0, &a->c0, &b->c0); fp_add(&r->c1, &a->c1, &b->c1); }
static void fp2_sub(fp2 *r, const fp2 *a, const fp2 *b) { fp_sub(&r->c0, &a->c0, &b->c0); fp_sub(&r->c1, &a->c1, &b->c1); }
static void fp2_neg(fp2 *r, const fp2 *a) { fp_neg(&r->c0, &a->c0); fp_neg(&r->c1, &a->c1); }
static void fp2_dbl(fp2 *r, const fp2 *a) { fp2_add(r, a, a); }
static void fp2_conj(fp2 *r, const fp2 *a) { r->c0 = a->c0; fp_neg(&r->c1, &a->c1); }
static int fp2_is_zero(const fp2 *a) { return fp_is_zero(&a->c0) && fp_is_zero(&a->c1); }
static int fp2_eq(const fp2 *a, const fp2 *b) { return fp_eq(&a->c0, &b->c0) && fp_eq(&a->c1, &b->c1); }
static void fp2_mul(fp2 *r, const fp2 *a, const fp2 *b) {
  fp t0, t1, s0, s1, t2;
  fp_mul(&t0, &a->c0, &b->c0); fp_mul(&t1, &a->c1, &b->c1);
  fp_add(&s0, &a->c0, &a->c1); fp_add(&s1, &b->c0, &b->c1);
  fp_mul(&t2, &s0, &s1);
  fp_sub(&r->c0, &t0, &t1);
  fp_sub(&t2, &t2, &t0); fp_sub(&r->c1, &t2, &t1);
}
static void fp2_sqr(fp2 *r, const fp2 *a) {
  fp s, d, m;
  fp_add(&s, &a->c0, &a->c1); fp_sub(&d, &a->c0, &a->c1); fp_mul(&m, &a->c0, &a->c1);
  fp_mul(&r->c0, &s, &d); fp_dbl(&r->c1, &m);
}
static void fp2_mul_fp(fp2 *r, const fp2 *a, const fp *k) { fp_mul(&r->c0, &a->c0, k); fp_mul(&r->c1, &a->c1, k); }
/* multiply by xi = 9 + i */
static void fp2_mul_xi(fp2 *r, const fp2 *a) {
  fp t0, t1, a8, b8;
  fp_dbl(&a8, &a->c0); fp_dbl(&a8, &a8); fp_dbl(&a8, &a8); fp_add(&t0, &a8, &a->c0); /* 9 a0 */
  fp_dbl(&b8, &a->c1); fp_dbl(&b8, &b8); fp_dbl(&b8, &b8); fp_add(&t1, &b8, &a->c1); /* 9 a1 */
  fp r0, r1;
  fp_sub(&r0, &t0, &a->c1); fp_add(&r1, &t1, &a->c0);
  r->c0 = r0; r->c1 = r1;
}
static void fp2_inv(fp2 *r, const fp2 *a) {
  fp n, t, ninv;
  fp_sqr(&n, &a->c0); fp_sqr(&t, &a->c1); fp_add(&n, &n, &t); fp_inv(&ninv, &n);
  fp_mul(&r->c0, &a->c0, &ninv); fp_mul(&t, &a->c1, &ninv); fp_neg(&r->c1, &t);
}
/* a^e for a multi-limb little-endian exponent of nlimbs 64-bit words */
static void fp2_pow(fp2 *r, const fp2 *a, const u64 *e, int nlimbs) {
  fp2 acc = FP2_ONE;
  for (int i = nlimbs * 64 - 1; i >= 0; --i) {
    fp2_sqr(&acc, &acc);
    if ((e[i >> 6] >> (i & 63)) & 1) fp2_mul(&acc, &acc, a);
  }
  *r = acc;
}

/* ------------------------------------------------------------------------------------------ */
/* Fq6 = Fq2[v]/(v^3 - xi), Fq12 = Fq6[w]/(w^2 - v)                                           */
/* ------------------------------------------------------------------------------------------ */
typedef struct { fp2 c0, c1, c2; } fp6;
typedef struct { fp6 c0, c1; } fp12;
static fp12 FP12_ONE;

static void fp6_add(fp6 *r, const fp6 *a, const fp6 *b) { fp2_add(&r->c0, &a->c0, &b->c0); fp2_add(&r->c1, &a->c1, &b->c1); fp2_add(&r->c2, &a->c2, &b->c2); }
static void fp6_sub(fp6 *r, const fp6 *a, const fp6 *b) { fp2_sub(&r->c0, &a->c0, &b->c0); fp2_sub(&r->c1, &a->c1, &b->c1); fp2_sub(&r->c2, &a->c2, &b->c2); }
static void fp6_neg(fp6 *r, const fp6 *a) { fp2_neg(&r->c0, &a->c0); fp2_neg(&r->c1, &a->c1); fp2_neg(&r->c2, &a->c2); }
static void fp6_mul_v(fp6 *r, const fp6 *a) { fp2 t; fp2_mul_xi(&t, &a->c2); r->c2 = a->c1; r->c1 = a->c0; r->c0 = t; }
static void fp6_mul(fp6 *r, const fp6 *a, const fp6 *b) {
  fp2 v0, v1, v2, s, t, u, c0, c1, c2;
  fp2_mul(&v0, &a->c0, &b->c0); fp2_mul(&v1, &a->c1, &b->c1); fp2_mul(&v2, &a->c2, &b->c2);
  fp2_add(&s, &a->c1, &a->c2); fp2_add(&t, &b->c1, &b->c2); fp2_mul(&u, &s, &t);
  fp2_sub(&u, &u, &v1); fp2_sub(&u, &u, &v2); fp2_mul_xi(&u, &u); fp2_add(&c0, &u, &v0);
  fp2_add(&s, &a->c0, &a->c1); fp2_add(&t, &b->c0, &b->c1); fp2_mul(&u, &s, &t);
  fp2_sub(&u, &u, &v0); fp2_sub(&u, &u, &v1); fp2_mul_xi(&s, &v2); fp2_add(&c1, &u, &s);
  fp2_add(&s, &a->c0, &a->c2); fp2_add(&t, &b->c0, &b->c2); fp2_mul(&u, &s, &t);
  fp2_sub(&u, &u, &v0); fp2_sub(&u, &u, &v2); fp2_add(&c2, &u, &v1);
  r->c0 = c0; r->c1 = c1; r->c2 = c2;
}
static void fp6_mul_fp2(fp6 *r, const fp6 *a, const fp2 *k) { fp2_mul(&r->c0, &a->c0, k); fp2_mul(&r->c1, &a->c1, k); fp2_mul(&r->c2, &a->c2, k); }
/* a * (b0 + b1 v) */
static void fp6_mul_01(fp6 *r, const fp6 *a, const fp2 *b0, const fp2 *b1) {
  fp2 v0, v1, s, t, u, c0, c1, c2;
  fp2_mul(&v0, &a->c0, b0); fp2_mul(&v1, &a->c1, b1);
  fp2_mul(&u, &a->c2, b1); fp2_mul_xi(&u, &u); fp2_add(&c0, &u, &v0);
  fp2_add(&s, &a->c0, &a->c1); fp2_add(&t, b0, b1); fp2_mul(&u, &s, &t); fp2_sub(&u, &u, &v0); fp2_sub(&c1, &u, &v1);
  fp2_mul(&u, &a->c2, b0); fp2_add(&c2, &u, &v1);
  r->c0 = c0; r->c1 = c1; r->c2 = c2;
}
static void fp6_inv(fp6 *r, const fp6 *a) {
  fp2 t0, t1, t2, s, d;
  fp2_sqr(&t0, &a->c0); fp2_mul(&s, &a->c1, &a->c2); fp2_mul_xi(&s, &s); fp2_sub(&t0, &t0, &s);
  fp2_sqr(&t1, &a->c2); fp2_mul_xi(&t1, &t1); fp2_mul(&s, &a->c0, &a->c1); fp2_sub(&t1, &t1, &s);
  fp2_sqr(&t2, &a->c1); fp2_mul(&s, &a->c0, &a->c2); fp2_sub(&t2, &t2, &s);
  fp2_mul(&d, &a->c2, &t1); fp2_mul(&s, &a->c1, &t2); fp2_add(&d, &d, &s); fp2_mul_xi(&d, &d);
  fp2_mul(&s, &a->c0, &t0); fp2_add(&d, &d, &s);
  fp2_inv(&d, &d);
  fp2_mul(&r->c0, &t0, &d); fp2_mul(&r->c1, &t1, &d); fp2_mul(&r->c2, &t2, &d);
}

static void fp12_mul(fp12 *r, const fp12 *a, const fp12 *b) {
  fp6 t0, t1, s, t, u;
  fp6_mul(&t0, &a->c0, &b->c0); fp6_mul(&t1, &a->c1, &b->c1);
  fp6_add(&s, &a->c0, &a->c1); fp6_add(&t, &b->c0, &b->c1); fp6_mul(&u, &s, &t);
  fp6_sub(&u, &u, &t0); fp6_sub(&u, &u, &t1);
  fp6_mul_v(&s, &t1); fp6_add(&r->c0, &t0, &s);
  r->c1 = u;
}
static void fp12_sqr(fp12 *r, const fp12 *a) {
  fp6 ab, s, t, u;
  fp6_mul(&ab, &a->c0, &a->c1);
  fp6_add(&s, &a->c0, &a->c1); fp6_mul_v(&t, &a->c1); fp6_add(&t, &t, &a->c0); fp6_mul(&u, &s, &t);
  fp6_sub(&u, &u, &ab); fp6_mul_v(&s, &ab); fp6_sub(&r->c0, &u, &s);
  fp6_add(&r->c1, &ab, &ab);
}
static void fp12_conj(fp12 *r, const fp12 *a) { r->c0 = a->c0; fp6_neg(&r->c1, &a->c1); }
static void fp12_inv(fp12 *r, const fp12 *a) {
  fp6 t0, t1, d;
  fp6_mul(&t0, &a->c0, &a->c0); fp6_mul(&t1, &a->c1, &a->c1); fp6_mul_v(&t1, &t1); fp6_sub(&d, &t0, &t1);
  fp6_inv(&d, &d);
  fp6_mul(&r->c0, &a->c0, &d); fp6_mul(&t0, &a->c1, &d); fp6_neg(&r->c1, &t0);
}
static int fp12_eq(const fp12 *a, const fp12 *b) {
  return fp2_eq(&a->c0.c0, &b->c0.c0) && fp2_eq(&a->c0.c1, &b->c0.c1) && fp2_eq(&a->c0.c2, &b->c0.c2) &&
         fp2_eq(&a->c1.c0, &b->c1.c0) && fp2_eq(&a->c1.c1, &b->c1.c1) && fp2_eq(&a->c1.c2, &b->c1.c2);
}
/* f * (l0 + (l1 + l2 v) w): the sparse line shape produced by the D-twist untwist */
static void fp12_mul_line(fp12 *r, const fp12 *f, const fp2 *l0, const fp2 *l1, const fp2 *l2) {
  fp6 t0, t1, s, u; fp2 l01;
  fp6_mul_fp2(&t0, &f->c0, l0);
  fp6_mul_01(&t1, &f->c1, l1, l2);
  fp6_add(&s, &f->c0, &f->c1); fp2_add(&l01, l0, l1); fp6_mul_01(&u, &s, &l01, l2);
  fp6_sub(&u, &u, &t0); fp6_sub(&u, &u, &t1);
  fp6_mul_v(&s, &t1); fp6_add(&r->c0, &t0, &s);
  r->c1 = u;
}
/* polynomial-basis view: coefficient k of w^k.  c[2i] = c0.c_i, c[2i+1] = c1.c_i */
static fp2 *fp12_coef(fp12 *a, int k) {
  fp6 *h = (k & 1) ? &a->c1 : &a->c0;
  return (k >> 1) == 0 ? &h->c0 : (k >> 1) == 1 ? &h->c1 : &h->c2;
}
static void fp12_frob(fp12 *r, const fp12 *a, int power) {
  fp12 t = *a;
  const fp2 *tab = power == 1 ? FROB1 : power == 2 ? FROB2 : FROB3;
  for (int k = 0; k < 6; ++k) {
    fp2 *c = fp12_coef(&t, k);
    if (power & 1) fp2_conj(c, c);
    fp2_mul(c, c, &tab[k]);
  }
  *r = t;
}
/* Granger-Scott squaring, valid for elements of the cyclotomic subgroup (after the easy part) */
static void fp4_sqr(fp2 *r0, fp2 *r1, const fp2 *a, const fp2 *b) {
  fp2 a2, b2, s;
  fp2_sqr(&a2, a); fp2_sqr(&b2, b); fp2_add(&s, a, b); fp2_sqr(&s, &s);
  fp2_sub(&s, &s, &a2); fp2_sub(r1, &s, &b2);               /* 2ab */
  fp2_mul_xi(&b2, &b2); fp2_add(r0, &a2, &b2);              /* a^2 + xi b^2 */
}
static void fp12_cyclotomic_sqr(fp12 *r, const fp12 *a) {
  /* g = (g0 + g1 s) with three Fq4 pairs: (a0,b1), (b0,a2), (a1,b2) where a_i = c0.c_i, b_i = c1.c_i */
  fp2 t0, t1, t2, t3, t4, t5, x;
  fp4_sqr(&t0, &t1, &a->c0.c0, &a->c1.c1);
  fp4_sqr(&t2, &t3, &a->c1.c0, &a->c0.c2);
  fp4_sqr(&t4, &t5, &a->c0.c1, &a->c1.c2);
  fp12 o;
  /* a0' = 3 t0 - 2 a0 ; b1' = 3 t1 + 2 b1 */
  fp2_sub(&x, &t0, &a->c0.c0); fp2_dbl(&x, &x); fp2_add(&o.c0.c0, &x, &t0);
  fp2_add(&x, &t1, &a->c1.c1); fp2_dbl(&x, &x); fp2_add(&o.c1.c1, &x, &t1);
  /* b0' = 3 xi t5 + 2 b0 ; a2' = 3 t4 - 2 a2 */
  fp2_mul_xi(&t5, &t5);
  fp2_add(&x, &t5, &a->c1.c0); fp2_dbl(&x, &x); fp2_add(&o.c1.c0, &x, &t5);
  fp2_sub(&x, &t4, &a->c0.c2); fp2_dbl(&x, &x); fp2_add(&o.c0.c2, &x, &t4);
  /* a1' = 3 t2 - 2 a1 ; b2' = 3 t3 + 2 b2 */
  fp2_sub(&x, &t2, &a->c0.c1); fp2_dbl(&x, &x); fp2_add(&o.c0.c1, &x, &t2);
  fp2_add(&x, &t3, &a->c1.c2); fp2_dbl(&x, &x); fp2_add(&o.c1.c2, &x, &t3);
  *r = o;
}

/* ------------------------------------------------------------------------------------------ */
/* curve constants                                                                            */
/* ------------------------------------------------------------------------------------------ */
#define BN_U 4965661367192848881ULL
static fp FP_B;             /* 3 */
typedef struct { fp x, y; int inf; } g1a;           /* affine G1 */
typedef struct { fp2 x, y; int inf; } g2a;          /* affine G2 (twist) */
static g1a G1_GEN; static g2a G2_GEN, G2_GEN_NEG;
static u64 ORDER_R[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};

/* signed-digit expansion of 6u+2 after the leading 1, most-significant first; built at init (NAF) */
static int8_t ATE_NAF[80]; static int ATE_NAF_LEN;

static pthread_once_t g_once = PTHREAD_ONCE_INIT;

static void mul_small_512(u64 *acc, int n, u64 m) { /* acc (n limbs) *= m */
  u128 c = 0;
  for (int i = 0; i < n; ++i) { c += (u128)acc[i] * m; acc[i] = (u64)c; c >>= 64; }
}
static void sub_small(u64 *acc, int n, u64 s) {
  for (int i = 0; i < n && s; ++i) { u64 o = acc[i]; acc[i] = o - s; s = o < s; }
}
static u64 div_small(u64 *acc, int n, u64 d) { /* acc /= d, returns remainder */
  u128 rem = 0;
  for (int i = n - 1; i >= 0; --i) { u128 cur = (rem << 64) | acc[i]; acc[i] = (u64)(cur / d); rem = cur % d; }
  return (u64)rem;
}

static void oracle_init(void) {
  /* -q^-1 mod 2^64 by Newton iteration */
  u64 inv = 1;
  for (int i = 0; i < 6; ++i) inv *= 2 - FP_Q.l[0] * inv;
  N0INV = (u64)0 - inv;
  /* R mod q and R^2 mod q by repeated doubling of 1 (non-Montgomery add) */
  fp x = {{1, 0, 0, 0}};
  for (int i = 0; i < 256; ++i) fp_add(&x, &x, &x);
  FP_ONE = x;
  for (int i = 0; i < 256; ++i) fp_add(&x, &x, &x);
  FP_R2 = x;
  memset(&FP_ZERO, 0, sizeof FP_ZERO);
  /* exponents q-2, (q+1)/4 */
  memcpy(EXP_QM2, FP_Q.l, 32); EXP_QM2[0] -= 2;
  { u64 t[4]; memcpy(t, FP_Q.l, 32); t[0] += 1; /* q+1: no carry since low limb ...47 */
    for (int i = 0; i < 4; ++i) EXP_QP1D4[i] = (t[i] >> 2) | (i < 3 ? t[i + 1] << 62 : 0); }
  fp_set_u64(&FP_B, 3);
  FP2_ZERO.c0 = FP_ZERO; FP2_ZERO.c1 = FP_ZERO; FP2_ONE.c0 = FP_ONE; FP2_ONE.c1 = FP_ZERO;
  memset(&FP12_ONE, 0, sizeof FP12_ONE); FP12_ONE.c0.c0 = FP2_ONE;
  /* xi = 9 + i, twist b' = 3/xi */
  fp2 xi; fp_set_u64(&xi.c0, 9); fp_set_u64(&xi.c1, 1);
  fp2 xinv; fp2_inv(&xinv, &xi);
  fp2_mul_fp(&TWIST_B, &xinv, &FP_B);
  fp2_add(&TWIST_3B, &TWIST_B, &TWIST_B); fp2_add(&TWIST_3B, &TWIST_3B, &TWIST_B);
  /* Frobenius coefficients gamma_{j,k} = xi^(k (q^j - 1)/6): compute g_j = xi^((q^j-1)/6) then powers */
  for (int j = 1; j <= 3; ++j) {
    u64 e[12]; memset(e, 0, sizeof e); e[0] = 1;
    for (int t = 0; t < j; ++t) { /* e *= q (schoolbook on up to 12 limbs) */
      u64 prod[12]; memset(prod, 0, sizeof prod);
      for (int a = 0; a < 12; ++a) { u128 c = 0; for (int b = 0; b < 4 && a + b < 12; ++b) { c += (u128)e[a] * FP_Q.l[b] + prod[a + b]; prod[a + b] = (u64)c; c >>= 64; }
        for (int k = a + 4; c && k < 12; ++k) { c += prod[k]; prod[k] = (u64)c; c >>= 64; } }
      memcpy(e, prod, sizeof e);
    }
    sub_small(e, 12, 1); div_small(e, 12, 6);
    fp2 g; fp2_pow(&g, &xi, e, 12);
    fp2 *tab = j == 1 ? FROB1 : j == 2 ? FROB2 : FROB3;
    tab[0] = FP2_ONE;
    for (int k = 1; k < 6; ++k) fp2_mul(&tab[k], &tab[k - 1], &g);
  }
  TW_FROB_X1 = FROB1[2]; TW_FROB_Y1 = FROB1[3]; TW_FROB_X2 = FROB2[2];
  /* generators */
  fp_set_u64(&G1_GEN.x, 1); fp_set_u64(&G1_GEN.y, 2); G1_GEN.inf = 0;
  static const u8 g2x0[32] = {0x18,0x00,0xde,0xef,0x12,0x1f,0x1e,0x76,0x42,0x6a,0x00,0x66,0x5e,0x5c,0x44,0x79,0x67,0x43,0x22,0xd4,0xf7,0x5e,0xda,0xdd,0x46,0xde,0xbd,0x5c,0xd9,0x92,0xf6,0xed};
  static const u8 g2x1[32] = {0x19,0x8e,0x93,0x93,0x92,0x0d,0x48,0x3a,0x72,0x60,0xbf,0xb7,0x31,0xfb,0x5d,0x25,0xf1,0xaa,0x49,0x33,0x35,0xa9,0xe7,0x12,0x97,0xe4,0x85,0xb7,0xae,0xf3,0x12,0xc2};
  static const u8 g2y0[32] = {0x12,0xc8,0x5e,0xa5,0xdb,0x8c,0x6d,0xeb,0x4a,0xab,0x71,0x80,0x8d,0xcb,0x40,0x8f,0xe3,0xd1,0xe7,0x69,0x0c,0x43,0xd3,0x7b,0x4c,0xe6,0xcc,0x01,0x66,0xfa,0x7d,0xaa};
  static const u8 g2y1[32] = {0x09,0x06,0x89,0xd0,0x58,0x5f,0xf0,0x75,0xec,0x9e,0x99,0xad,0x69,0x0c,0x33,0x95,0xbc,0x4b,0x31,0x33,0x70,0xb3,0x8e,0xf3,0x55,0xac,0xda,0xdc,0xd1,0x22,0x97,0x5b};
  fp_from_be(&G2_GEN.x.c0, g2x0); fp_from_be(&G2_GEN.x.c1, g2x1);
  fp_from_be(&G2_GEN.y.c0, g2y0); fp_from_be(&G2_GEN.y.c1, g2y1); G2_GEN.inf = 0;
  G2_GEN_NEG = G2_GEN; fp2_neg(&G2_GEN_NEG.y, &G2_GEN.y);
  /* NAF of 6u+2 (65 bits -> use u128) */
  u128 s = (u128)6 * BN_U + 2;
  int8_t digs[80]; int nd = 0;
  while (s) {
    if (s & 1) { int d = 2 - (int)(s & 3); digs[nd++] = (int8_t)d; s -= (u128)(d < 0 ? 0 : d); if (d < 0) s += 1; }
    else digs[nd++] = 0;
    s >>= 1;
  }
  /* canonical NAF of 6u+2 starts 1,0,-1: rewrite the top as 1,1 (same value, one doubling fewer;
   * the 64-digit form of SURVEY.md Appendix A.1) */
  if (nd >= 3 && digs[nd - 1] == 1 && digs[nd - 2] == 0 && digs[nd - 3] == -1) { digs[nd - 3] = 1; digs[nd - 2] = 1; --nd; }
  /* digs[nd-1] is the leading 1; store the rest most-significant first */
  ATE_NAF_LEN = 0;
  for (int i = nd - 2; i >= 0; --i) ATE_NAF[ATE_NAF_LEN++] = digs[i];
}
static void ensure_init(void) { pthread_once(&g_once, oracle_init); }

/* ------------------------------------------------------------------------------------------ */
/* G1 / G2 Jacobian arithmetic (generic over the field via macros would obscure; written twice) */
/* ------------------------------------------------------------------------------------------ */
typedef struct { fp x, y, z; } g1j;   /* z == 0 <=> identity */
typedef struct { fp2 x, y, z; } g2j;

static void g1j_from_affine(g1j *r, const g1a *p) {
  if (p->inf) { r->x = FP_ONE; r->y = FP_ONE; r->z = FP_ZERO; } else { r->x = p->x; r->y = p->y; r->z = FP_ONE; }
}
static void g1j_to_affine(g1a *r, const g1j *p) {
  if (fp_is_zero(&p->z)) { r->inf = 1; r->x = FP_ZERO; r->y = FP_ZERO; return; }
  fp zi, zi2, zi3; fp_inv(&zi, &p->z); fp_sqr(&zi2, &zi); fp_mul(&zi3, &zi2, &zi);
  fp_mul(&r->x, &p->x, &zi2); fp_mul(&r->y, &p->y, &zi3); r->inf = 0;
}
static void g1j_dbl(g1j *r, const g1j *p) {
  if (fp_is_zero(&p->z)) { *r = *p; return; }
  fp a, b, c, d, e, f, t; g1j o;
  fp_sqr(&a, &p->x); fp_sqr(&b, &p->y); fp_sqr(&c, &b);
  fp_add(&d, &p->x, &b); fp_sqr(&d, &d); fp_sub(&d, &d, &a); fp_sub(&d, &d, &c); fp_dbl(&d, &d);
  fp_dbl(&e, &a); fp_add(&e, &e, &a); fp_sqr(&f, &e);
  fp_dbl(&t, &d); fp_sub(&o.x, &f, &t);
  fp_mul(&o.z, &p->y, &p->z); fp_dbl(&o.z, &o.z);
  fp_sub(&t, &d, &o.x); fp_mul(&t, &e, &t);
  fp_dbl(&c, &c); fp_dbl(&c, &c); fp_dbl(&c, &c); fp_sub(&o.y, &t, &c);
  *r = o;
}
static void g1j_add(g1j *r, const g1j *p, const g1j *q) {
  if (fp_is_zero(&p->z)) { *r = *q; return; }
  if (fp_is_zero(&q->z)) { *r = *p; return; }
  fp z1z1, z2z2, u1, u2, s1, s2, h, i, j, rr, v, t; g1j o;
  fp_sqr(&z1z1, &p->z); fp_sqr(&z2z2, &q->z);
  fp_mul(&u1, &p->x, &z2z2); fp_mul(&u2, &q->x, &z1z1);
  fp_mul(&s1, &p->y, &q->z); fp_mul(&s1, &s1, &z2z2);
  fp_mul(&s2, &q->y, &p->z); fp_mul(&s2, &s2, &z1z1);
  if (fp_eq(&u1, &u2)) {
    if (fp_eq(&s1, &s2)) { g1j_dbl(r, p); return; }
    r->x = FP_ONE; r->y = FP_ONE; r->z = FP_ZERO; return;
  }
  fp_sub(&h, &u2, &u1); fp_dbl(&i, &h); fp_sqr(&i, &i); fp_mul(&j, &h, &i);
  fp_sub(&rr, &s2, &s1); fp_dbl(&rr, &rr); fp_mul(&v, &u1, &i);
  fp_sqr(&o.x, &rr); fp_sub(&o.x, &o.x, &j); fp_dbl(&t, &v); fp_sub(&o.x, &o.x, &t);
  fp_sub(&t, &v, &o.x); fp_mul(&t, &rr, &t); fp_mul(&s1, &s1, &j); fp_dbl(&s1, &s1); fp_sub(&o.y, &t, &s1);
  fp_add(&t, &p->z, &q->z); fp_sqr(&t, &t); fp_sub(&t, &t, &z1z1); fp_sub(&t, &t, &z2z2); fp_mul(&o.z, &t, &h);
  *r = o;
}
/* scalar: 256-bit little-endian limbs, used as-is (not reduced), cf. bn256.json scalars up to 2^256-1 */
static void g1j_mul(g1j *r, const g1j *p, const u64 *k) {
  g1j acc; acc.x = FP_ONE; acc.y = FP_ONE; acc.z = FP_ZERO;
  for (int i = 255; i >= 0; --i) {
    g1j_dbl(&acc, &acc);
    if ((k[i >> 6] >> (i & 63)) & 1) g1j_add(&acc, &acc, p);
  }
  *r = acc;
}

static void g2j_from_affine(g2j *r, const g2a *p) {
  if (p->inf) { r->x = FP2_ONE; r->y = FP2_ONE; r->z = FP2_ZERO; } else { r->x = p->x; r->y = p->y; r->z = FP2_ONE; }
}
static void g2j_to_affine(g2a *r, const g2j *p) {
  if (fp2_is_zero(&p->z)) { r->inf = 1; r->x = FP2_ZERO; r->y = FP2_ZERO; return; }
  fp2 zi, zi2, zi3; fp2_inv(&zi, &p->z); fp2_sqr(&zi2, &zi); fp2_mul(&zi3, &zi2, &zi);
  fp2_mul(&r->x, &p->x, &zi2); fp2_mul(&r->y, &p->y, &zi3); r->inf = 0;
}
static void g2j_dbl(g2j *r, const g2j *p) {
  if (fp2_is_zero(&p->z)) { *r = *p; return; }
  fp2 a, b, c, d, e, f, t; g2j o;
  fp2_sqr(&a, &p->x); fp2_sqr(&b, &p->y); fp2_sqr(&c, &b);
  fp2_add(&d, &p->x, &b); fp2_sqr(&d, &d); fp2_sub(&d, &d, &a); fp2_sub(&d, &d, &c); fp2_dbl(&d, &d);
  fp2_dbl(&e, &a); fp2_add(&e, &e, &a); fp2_sqr(&f, &e);
  fp2_dbl(&t, &d); fp2_sub(&o.x, &f, &t);
  fp2_mul(&o.z, &p->y, &p->z); fp2_dbl(&o.z, &o.z);
  fp2_sub(&t, &d, &o.x); fp2_mul(&t, &e, &t);
  fp2_dbl(&c, &c); fp2_dbl(&c, &c); fp2_dbl(&c, &c); fp2_sub(&o.y, &t, &c);
  *r = o;
}
static void g2j_add(g2j *r, const g2j *p, const g2j *q) {
  if (fp2_is_zero(&p->z)) { *r = *q; return; }
  if (fp2_is_zero(&q->z)) { *r = *p; return; }
  fp2 z1z1, z2z2, u1, u2, s1, s2, h, i, j, rr, v, t; g2j o;
  fp2_sqr(&z1z1, &p->z); fp2_sqr(&z2z2, &q->z);
  fp2_mul(&u1, &p->x, &z2z2); fp2_mul(&u2, &q->x, &z1z1);
  fp2_mul(&s1, &p->y, &q->z); fp2_mul(&s1, &s1, &z2z2);
  fp2_mul(&s2, &q->y, &p->z); fp2_mul(&s2, &s2, &z1z1);
  if (fp2_eq(&u1, &u2)) {
    if (fp2_eq(&s1, &s2)) { g2j_dbl(r, p); return; }
    r->x = FP2_ONE; r->y = FP2_ONE; r->z = FP2_ZERO; return;
  }
  fp2_sub(&h, &u2, &u1); fp2_dbl(&i, &h); fp2_sqr(&i, &i); fp2_mul(&j, &h, &i);
  fp2_sub(&rr, &s2, &s1); fp2_dbl(&rr, &rr); fp2_mul(&v, &u1, &i);
  fp2_sqr(&o.x, &rr); fp2_sub(&o.x, &o.x, &j); fp2_dbl(&t, &v); fp2_sub(&o.x, &o.x, &t);
  fp2_sub(&t, &v, &o.x); fp2_mul(&t, &rr, &t); fp2_mul(&s1, &s1, &j); fp2_dbl(&s1, &s1); fp2_sub(&o.y, &t, &s1);
  fp2_add(&t, &p->z, &q->z); fp2_sqr(&t, &t); fp2_sub(&t, &t, &z1z1); fp2_sub(&t, &t, &z2z2); fp2_mul(&o.z, &t, &h);
  *r = o;
}
static void g2j_mul(g2j *r, const g2j *p, const u64 *k) {
  g2j acc; acc.x = FP2_ONE; acc.y = FP2_ONE; acc.z = FP2_ZERO;
  for (int i = 255; i >= 0; --i) {
    g2j_dbl(&acc, &acc);
    if ((k[i >> 6] >> (i & 63)) & 1) g2j_add(&acc, &acc, p);
  }
  *r = acc;
}
static int g1a_on_curve(const g1a *p) {
  if (p->inf) return 1;
  fp l, r; fp_sqr(&l, &p->y); fp_sqr(&r, &p->x); fp_mul(&r, &r, &p->x); fp_add(&r, &r, &FP_B);
  return fp_eq(&l, &r);
}
static int g2a_on_curve(const g2a *p) {
  if (p->inf) return 1;
  fp2 l, r; fp2_sqr(&l, &p->y); fp2_sqr(&r, &p->x); fp2_mul(&r, &r, &p->x); fp2_add(&r, &r, &TWIST_B);
  return fp2_eq(&l, &r);
}
static int g2a_in_subgroup(const g2a *p) {
  if (p->inf) return 1;
  g2j j, o; g2j_from_affine(&j, p); g2j_mul(&o, &j, ORDER_R);
  return fp2_is_zero(&o.z);
}

/* ------------------------------------------------------------------------------------------ */
/* optimal-ate Miller loop, multi-pair with shared squaring, + final exponentiation           */
/* ------------------------------------------------------------------------------------------ */
typedef struct { fp2 x, y, z; } g2h;  /* homogeneous projective twist point (x = X/Z, y = Y/Z) */

/* T <- 2T; line (scaled by an Fq2 factor) l0 = 2YZ*yP, l1 = -3X^2*xP, l2 = Y^2 - 3b'Z^2 */
static void dbl_step(g2h *t, fp2 *l0, fp2 *l1, fp2 *l2, const g1a *p) {
  fp2 xy, b, c, e, f, h, x2, s, u; g2h o;
  fp2_mul(&xy, &t->x, &t->y); fp2_sqr(&b, &t->y); fp2_sqr(&c, &t->z);
  fp2_mul(&e, &c, &TWIST_3B);                          /* E = 3b' Z^2 */
  fp2_dbl(&f, &e); fp2_add(&f, &f, &e);                /* F = 3E */
  fp2_add(&h, &t->y, &t->z); fp2_sqr(&h, &h); fp2_sub(&h, &h, &b); fp2_sub(&h, &h, &c); /* H = 2YZ */
  fp2_sqr(&x2, &t->x);
  fp2_sub(&s, &b, &f); fp2_mul(&o.x, &xy, &s); fp2_dbl(&o.x, &o.x);        /* X3 = 2XY(B - F) */
  fp2_add(&s, &b, &f); fp2_sqr(&s, &s); fp2_sqr(&u, &e);
  fp2_dbl(&u, &u); fp2_dbl(&u, &u); { fp2 u3; fp2_dbl(&u3, &u); fp2_add(&u, &u3, &u); }  /* 12 E^2 */
  fp2_sub(&o.y, &s, &u);                                                   /* Y3 = (B+F)^2 - 12E^2 */
  fp2_mul(&o.z, &b, &h); fp2_dbl(&o.z, &o.z); fp2_dbl(&o.z, &o.z);         /* Z3 = 4BH */
  fp2_mul_fp(l0, &h, &p->y);
  fp2_dbl(&s, &x2); fp2_add(&s, &s, &x2); fp2_neg(&s, &s); fp2_mul_fp(l1, &s, &p->x);
  fp2_sub(l2, &b, &e);
  *t = o;
}
/* T <- T + Q (Q affine); line l0 = mu*yP, l1 = -theta*xP, l2 = theta*x2 - mu*y2 */
static void add_step(g2h *t, fp2 *l0, fp2 *l1, fp2 *l2, const g2a *q, const g1a *p) {
  fp2 theta, mu, c, d, e, f, g, h, s, u; g2h o;
  fp2_mul(&s, &q->y, &t->z); fp2_sub(&theta, &t->y, &s);
  fp2_mul(&s, &q->x, &t->z); fp2_sub(&mu, &t->x, &s);
  fp2_sqr(&c, &theta); fp2_sqr(&d, &mu); fp2_mul(&e, &mu, &d);
  fp2_mul(&f, &t->z, &c); fp2_mul(&g, &t->x, &d);
  fp2_add(&h, &e, &f); fp2_sub(&h, &h, &g); fp2_sub(&h, &h, &g);
  fp2_mul(&o.x, &mu, &h);
  fp2_sub(&s, &g, &h); fp2_mul(&s, &theta, &s); fp2_mul(&u, &e, &t->y); fp2_sub(&o.y, &s, &u);
  fp2_mul(&o.z, &t->z, &e);
  fp2_mul_fp(l0, &mu, &p->y);
  fp2_neg(&s, &theta); fp2_mul_fp(l1, &s, &p->x);
  fp2_mul(&s, &theta, &q->x); fp2_mul(&u, &mu, &q->y); fp2_sub(l2, &s, &u);
  *t = o;
}

#define MAX_PAIRS 16
/* product over pairs of the Miller function; pairs with an identity member are skipped */
static void miller_loop_multi(fp12 *out, const g1a *ps, const g2a *qs, int k) {
  g2h t[MAX_PAIRS]; g2a qn[MAX_PAIRS]; const g1a *pp[MAX_PAIRS]; const g2a *qq[MAX_PAIRS];
  int n = 0;
  for (int i = 0; i < k; ++i) {
    if (ps[i].inf || qs[i].inf) continue;
    pp[n] = &ps[i]; qq[n] = &qs[i];
    t[n].x = qs[i].x; t[n].y = qs[i].y; t[n].z = FP2_ONE;
    qn[n] = qs[i]; fp2_neg(&qn[n].y, &qs[i].y);
    ++n;
  }
  fp12 f = FP12_ONE; fp2 l0, l1, l2;
  if (n == 0) { *out = f; return; }
  for (int d = 0; d < ATE_NAF_LEN; ++d) {
    fp12_sqr(&f, &f);
    for (int i = 0; i < n; ++i) { dbl_step(&t[i], &l0, &l1, &l2, pp[i]); fp12_mul_line(&f, &f, &l0, &l1, &l2); }
    if (ATE_NAF[d]) for (int i = 0; i < n; ++i) {
      add_step(&t[i], &l0, &l1, &l2, ATE_NAF[d] > 0 ? qq[i] : &qn[i], pp[i]); fp12_mul_line(&f, &f, &l0, &l1, &l2);
    }
  }
  for (int i = 0; i < n; ++i) {
    g2a q1, q2;
    fp2_conj(&q1.x, &qq[i]->x); fp2_mul(&q1.x, &q1.x, &TW_FROB_X1);
    fp2_conj(&q1.y, &qq[i]->y); fp2_mul(&q1.y, &q1.y, &TW_FROB_Y1); q1.inf = 0;
    fp2_mul(&q2.x, &qq[i]->x, &TW_FROB_X2); q2.y = qq[i]->y; q2.inf = 0;   /* -pi^2(Q) = (x*g, y) */
    add_step(&t[i], &l0, &l1, &l2, &q1, pp[i]); fp12_mul_line(&f, &f, &l0, &l1, &l2);
    add_step(&t[i], &l0, &l1, &l2, &q2, pp[i]); fp12_mul_line(&f, &f, &l0, &l1, &l2);
  }
  *out = f;
}
static void fp12_pow_u(fp12 *r, const fp12 *a) {   /* a^u for a in the cyclotomic subgroup */
  fp12 acc = *a;
  for (int i = 61; i >= 0; --i) {
    fp12_cyclotomic_sqr(&acc, &acc);
    if ((BN_U >> i) & 1) fp12_mul(&acc, &acc, a);
  }
  *r = acc;
}
/* f^((q^12-1)/r) exactly: easy part (q^6-1)(q^2+1), hard part by the Devegili/Scott et al chain
 * with lambda_3 = 1, lambda_2 = 6u^2+1, lambda_1 = -36u^3-18u^2-12u+1, lambda_0 = -36u^3-30u^2-18u-2 */
static void final_exp(fp12 *r, const fp12 *fin) {
  fp12 f, t, inv;
  fp12_inv(&inv, fin); fp12_conj(&t, fin); fp12_mul(&f, &t, &inv);       /* f^(q^6-1) */
  fp12_frob(&t, &f, 2); fp12_mul(&f, &t, &f);                             /* ^(q^2+1) */
  fp12 fu, fu2, fu3, y0, y1, y2, y3, y4, y5, y6, a, b;
  fp12_pow_u(&fu, &f); fp12_pow_u(&fu2, &fu); fp12_pow_u(&fu3, &fu2);
  fp12_frob(&a, &f, 1); fp12_frob(&b, &f, 2); fp12_mul(&y0, &a, &b); fp12_frob(&a, &f, 3); fp12_mul(&y0, &y0, &a);
  fp12_conj(&y1, &f);
  fp12_frob(&y2, &fu2, 2);
  fp12_frob(&a, &fu, 1); fp12_conj(&y3, &a);
  fp12_frob(&a, &fu2, 1); fp12_mul(&a, &a, &fu); fp12_conj(&y4, &a);
  fp12_conj(&y5, &fu2);
  fp12_frob(&a, &fu3, 1); fp12_mul(&a, &a, &fu3); fp12_conj(&y6, &a);
  fp12 t0, t1;
  fp12_cyclotomic_sqr(&t0, &y6); fp12_mul(&t0, &t0, &y4); fp12_mul(&t0, &t0, &y5);
  fp12_mul(&t1, &y3, &y5); fp12_mul(&t1, &t1, &t0);
  fp12_mul(&t0, &t0, &y2);
  fp12_cyclotomic_sqr(&t1, &t1); fp12_mul(&t1, &t1, &t0); fp12_cyclotomic_sqr(&t1, &t1);
  fp12_mul(&t0, &t1, &y1); fp12_mul(&t1, &t1, &y0);
  fp12_cyclotomic_sqr(&t0, &t0); fp12_mul(r, &t0, &t1);
}
static void pairing_product(fp12 *out, const g1a *ps, const g2a *qs, int k) {
  fp12 f; miller_loop_multi(&f, ps, qs, k); final_exp(out, &f);
}

/* ------------------------------------------------------------------------------------------ */
/* SHA-256 (FIPS 180-4)                                                                       */
/* ------------------------------------------------------------------------------------------ */
static const u32 SHA_K[64] = {
  0x428a2f98,0x71374491,0xb5c0fbcf,0xe9b5dba5,0x3956c25b,0x59f111f1,0x923f82a4,0xab1c5ed5,0xd807aa98,0x12835b01,0x243185be,0x550c7dc3,0x72be5d74,0x80deb1fe,0x9bdc06a7,0xc19bf174,
  0xe49b69c1,0xefbe4786,0x0fc19dc6,0x240ca1cc,0x2de92c6f,0x4a7484aa,0x5cb0a9dc,0x76f988da,0x983e5152,0xa831c66d,0xb00327c8,0xbf597fc7,0xc6e00bf3,0xd5a79147,0x06ca6351,0x14292967,
  0x27b70a85,0x2e1b2138,0x4d2c6dfc,0x53380d13,0x650a7354,0x766a0abb,0x81c2c92e,0x92722c85,0xa2bfe8a1,0xa81a664b,0xc24b8b70,0xc76c51a3,0xd192e819,0xd6990624,0xf40e3585,0x106aa070,
  0x19a4c116,0x1e376c08,0x2748774c,0x34b0bcb5,0x391c0cb3,0x4ed8aa4a,0x5b9cca4f,0x682e6ff3,0x748f82ee,0x78a5636f,0x84c87814,0x8cc70208,0x90befffa,0xa4506ceb,0xbef9a3f7,0xc67178f2};
#define ROR(x, n) (((x) >> (n)) | ((x) << (32 - (n))))
static void sha256_block(u32 *h, const u8 *blk) {
  u32 w[64];
  for (int i = 0; i < 16; ++i) w[i] = ((u32)blk[4 * i] << 24) | ((u32)blk[4 * i + 1] << 16) | ((u32)blk[4 * i + 2] << 8) | blk[4 * i + 3];
  for (int i = 16; i < 64; ++i) {
    u32 s0 = ROR(w[i - 15], 7) ^ ROR(w[i - 15], 18) ^ (w[i - 15] >> 3);
    u32 s1 = ROR(w[i - 2], 17) ^ ROR(w[i - 2], 19) ^ (w[i - 2] >> 10);
    w[i] = w[i - 16] + s0 + w[i - 7] + s1;
  }
  u32 a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
  for (int i = 0; i < 64; ++i) {
    u32 t1 = hh + (ROR(e, 6) ^ ROR(e, 11) ^ ROR(e, 25)) + ((e & f) ^ (~e & g)) + SHA_K[i] + w[i];
    u32 t2 = (ROR(a, 2) ^ ROR(a, 13) ^ ROR(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
    hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
  }
  h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}
/* SHA-256 of (msg || last) */
static void sha256_msg_plus_byte(u8 *digest, const u8 *msg, size_t len, u8 last) {
  u32 h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
  size_t total = len + 1, off = 0;
  u8 blk[64];
  while (total - off >= 64) {
    for (int i = 0; i < 64; ++i) { size_t p = off + i; blk[i] = p < len ? msg[p] : last; }
    sha256_block(h, blk); off += 64;
  }
  size_t rem = total - off;
  memset(blk, 0, 64);
  for (size_t i = 0; i < rem; ++i) { size_t p = off + i; blk[i] = p < len ? msg[p] : last; }
  blk[rem] = 0x80;
  if (rem >= 56) { sha256_block(h, blk); memset(blk, 0, 64); }
  u64 bits = (u64)total * 8;
  for (int i = 0; i < 8; ++i) blk[56 + i] = (u8)(bits >> (56 - 8 * i));
  sha256_block(h, blk);
  for (int i = 0; i < 8; ++i) { digest[4 * i] = (u8)(h[i] >> 24); digest[4 * i + 1] = (u8)(h[i] >> 16); digest[4 * i + 2] = (u8)(h[i] >> 8); digest[4 * i + 3] = (u8)h[i]; }
}

/* ------------------------------------------------------------------------------------------ */
/* hash-to-G1, /root/reference/src/hash.rs:29-63                                              */
/* ------------------------------------------------------------------------------------------ */
static const u64 FIVE_Q[4] = {0x2ca2bc723a70f263ULL, 0xf58714d70a38f4c2ULL, 0x99915c908786b9d3ULL, 0xf1f5883e65f820d0ULL}; /* hash.rs:11-14 */

/* One candidate of the try loop from its 256-bit digest value h (hash.rs:44-59): range rule, mod_u256, decompression
 * with the 0x02 prefix.  1 = yields a point, 0 = the loop moves on to the next counter. */
static int hash_candidate_point(g1a *out, u64 *h) {
  if (u256_geq(h, FIVE_Q)) return 0;                                /* :49-51 */
  /* mod_u256, utils.rs:27-37: while reduced > modulus { reduced -= modulus } (strict) */
  while (u256_geq(h, FP_Q.l) && memcmp(h, FP_Q.l, 32) != 0) u256_sub(h, h, FP_Q.l);
  /* arbitrary_string_to_g1 -> G1::from_compressed(0x02 || x): x < q else NotMember */
  if (u256_geq(h, FP_Q.l)) return 0;                                /* h == q: rejected */
  fp x, y, rhs; fp_from_u256(&x, h);
  fp_sqr(&rhs, &x); fp_mul(&rhs, &rhs, &x); fp_add(&rhs, &rhs, &FP_B);
  if (!fp_sqrt(&y, &rhs)) return 0;
  u64 yi[4]; fp_to_u256(yi, &y);
  if (yi[0] & 1) fp_neg(&y, &y);                                    /* 0x02 prefix: even y */
  out->x = x; out->y = y; out->inf = 0;
  return 1;
}
static int hash_to_g1(g1a *out, const u8 *msg, size_t len, int *tries) {
  u8 dg[32];
  for (int ctr = 0; ctr < 255; ++ctr) {                             /* hash.rs:40 */
    sha256_msg_plus_byte(dg, msg, len, (u8)ctr);                    /* :41-42 */
    u64 h[4]; be32_to_u256(h, dg);                                  /* :44 */
    if (!hash_candidate_point(out, h)) continue;
    if (tries) *tries = ctr + 1;
    return ST_OK;
  }
  if (tries) *tries = 255;
  return ST_HASH_TO_POINT;                                          /* :62 */
}

/* ------------------------------------------------------------------------------------------ */
/* decoders: /root/reference/src/utils.rs:107-127; all-zero bytes = identity (batch ABI)       */
/* ------------------------------------------------------------------------------------------ */
static int all_zero(const u8 *b, size_t n) { u8 o = 0; for (size_t i = 0; i < n; ++i) o |= b[i]; return o == 0; }
static int decode_g1(g1a *p, const u8 *b, u32 flags) {
  if (all_zero(b, 64)) {
    if (flags & FLAG_REJECT_IDENTITY) return ST_INVALID_GROUP_POINT;
    p->inf = 1; p->x = FP_ZERO; p->y = FP_ZERO; return ST_OK;
  }
  if (!fp_from_be(&p->x, b) || !fp_from_be(&p->y, b + 32)) return ST_NOT_MEMBER;
  p->inf = 0;
  return g1a_on_curve(p) ? ST_OK : ST_INVALID_GROUP_POINT;
}
static int decode_g2(g2a *p, const u8 *b, u32 flags) {
  if (all_zero(b, 128)) {
    if (flags & FLAG_REJECT_IDENTITY) return ST_INVALID_GROUP_POINT;
    p->inf = 1; p->x = FP2_ZERO; p->y = FP2_ZERO; return ST_OK;
  }
  if (!fp_from_be(&p->x.c0, b) || !fp_from_be(&p->x.c1, b + 32) || !fp_from_be(&p->y.c0, b + 64) || !fp_from_be(&p->y.c1, b + 96))
    return ST_NOT_MEMBER;
  p->inf = 0;
  if (!g2a_on_curve(p)) return ST_INVALID_GROUP_POINT;
  if ((flags & FLAG_G2_SUBGROUP_CHECK) && !g2a_in_subgroup(p)) return ST_INVALID_GROUP_POINT;
  return ST_OK;
}
static void encode_g1(u8 *b, const g1a *p) { if (p->inf) memset(b, 0, 64); else { fp_to_be(b, &p->x); fp_to_be(b + 32, &p->y); } }
static void encode_g2(u8 *b, const g2a *p) {
  if (p->inf) { memset(b, 0, 128); return; }
  fp_to_be(b, &p->x.c0); fp_to_be(b + 32, &p->x.c1); fp_to_be(b + 64, &p->y.c0); fp_to_be(b + 96, &p->y.c1);
}
static void encode_fp12(u8 *b, const fp12 *f) {
  const fp2 *c[6] = {&f->c0.c0, &f->c0.c1, &f->c0.c2, &f->c1.c0, &f->c1.c1, &f->c1.c2};
  for (int i = 0; i < 6; ++i) { fp_to_be(b + 64 * i, &c[i]->c0); fp_to_be(b + 64 * i + 32, &c[i]->c1); }
}

/* ------------------------------------------------------------------------------------------ */
/* exported API (ctypes)                                                                      */
/* ------------------------------------------------------------------------------------------ */
#define API __attribute__((visibility("default")))

API u64 bn254o_fp_mul_count(void) { return g_fp_mul_count; }
API void bn254o_fp_mul_count_reset(void) { g_fp_mul_count = 0; }

/* hash_to_try_and_increment -> 64-byte uncompressed point */
API int bn254o_hash_to_g1(const u8 *msg, size_t len, u8 *out64, int *tries) {
  ensure_init();
  g1a p; int st = hash_to_g1(&p, msg, len, tries);
  if (st == ST_OK) encode_g1(out64, &p); else memset(out64, 0, 64);
  return st;
}

/* ECDSA::verify on byte-encoded inputs -> status */
API int bn254o_verify(const u8 *msg, size_t len, const u8 *sig64, const u8 *pk128, u32 flags) {
  ensure_init();
  g1a ps[2]; g2a qs[2]; int st;
  if ((st = decode_g1(&ps[1], sig64, flags)) != ST_OK) return st;
  if ((st = decode_g2(&qs[0], pk128, flags)) != ST_OK) return st;
  if ((st = hash_to_g1(&ps[0], msg, len, NULL)) != ST_OK) return st;       /* ecdsa.rs:53 */
  qs[1] = G2_GEN_NEG;                                                        /* ecdsa.rs:56 */
  fp12 gt; pairing_product(&gt, ps, qs, 2);                                  /* ecdsa.rs:57 */
  return fp12_eq(&gt, &FP12_ONE) ? ST_OK : ST_VERIFICATION_FAILED;           /* ecdsa.rs:59-63 */
}

/* check_public_keys(pk_g2, pk_g1), ecdsa.rs:78-93 */
API int bn254o_check_public_keys(const u8 *pk_g2_128, const u8 *pk_g1_64, u32 flags) {
  ensure_init();
  g1a ps[2]; g2a qs[2]; int st;
  if ((st = decode_g2(&qs[0], pk_g2_128, flags)) != ST_OK) return st;
  if ((st = decode_g1(&ps[1], pk_g1_64, flags)) != ST_OK) return st;
  ps[0] = G1_GEN; qs[1] = G2_GEN_NEG;
  fp12 gt; pairing_product(&gt, ps, qs, 2);
  return fp12_eq(&gt, &FP12_ONE) ? ST_OK : ST_VERIFICATION_FAILED;
}

typedef struct {
  const u8 *msgs; const u64 *off; const u8 *sigs; const u8 *pks; size_t lo, hi; u32 flags; u8 *status; u64 fp_muls;
} batch_job;
static void *batch_worker(void *arg) {
  batch_job *j = (batch_job *)arg;
  g_fp_mul_count = 0;
  for (size_t i = j->lo; i < j->hi; ++i)
    j->status[i] = (u8)bn254o_verify(j->msgs + j->off[i], (size_t)(j->off[i + 1] - j->off[i]), j->sigs + 64 * i, j->pks + 128 * i, j->flags);
  j->fp_muls = g_fp_mul_count;
  return NULL;
}
/* n independent verifies, sharded contiguously over nthreads; returns total Fq multiplications */
API u64 bn254o_batch_verify(const u8 *msgs, const u64 *off, const u8 *sigs, const u8 *pks, size_t n, u32 flags,
                            u8 *status, int nthreads) {
  ensure_init();
  if (nthreads < 1) nthreads = 1;
  if (nthreads > 256) nthreads = 256;
  pthread_t th[256]; batch_job jobs[256];
  size_t per = (n + nthreads - 1) / nthreads;
  int used = 0;
  for (int t = 0; t < nthreads; ++t) {
    size_t lo = per * t, hi = lo + per > n ? n : lo + per;
    if (lo >= hi) break;
    jobs[t] = (batch_job){msgs, off, sigs, pks, lo, hi, flags, status, 0};
    if (nthreads == 1) batch_worker(&jobs[t]); else pthread_create(&th[t], NULL, batch_worker, &jobs[t]);
    ++used;
  }
  u64 total = 0;
  for (int t = 0; t < used; ++t) { if (nthreads != 1) pthread_join(th[t], NULL); total += jobs[t].fp_muls; }
  return total;
}

/* Randomised batch verification (no counterpart in the reference, which verifies
 * one tuple at a time, src/ecdsa.rs:49-64).  Restates include/bn254_hip.h:bn254_batch_verify_randomized:
 *   items are grouped 64 at a time; r_i = the first 16 (flag RAND64: 8) bytes of SHA-256(seed || le64(i)) read
 *   little-endian (0 -> 1); a group passes iff
 *       prod_{valid i} e(r_i H(m_i), pk_i) * e(sum_{valid i} r_i sig_i, -G2) == 1
 *   where "valid" = decoded and hashed without error.  Items of a passing group get their decode / hash
 *   status (0 if none); items of a failing group are verified one by one exactly as bn254o_verify. */
#define FLAG_RAND64 0x100u
#define FLAG_RAND_GLV 0x200u
/* eigenvalue of (x, y) -> (beta x, y) on G1 (include/bn254_hip.h: BN254_FLAG_RAND_GLV); lambda^2 + lambda + 1 = 0 mod r */
static const u64 GLV_LAMBDA[4] = {0x8b17ea66b99c90ddULL, 0x5bfc41088d8daaa7ULL, 0xb3c4d79d41a91758ULL, 0};
/* r_i * P for the scalar of item i; with FLAG_RAND_GLV r_i = k1 + k2*lambda, computed WITHOUT the endomorphism:
 * k1*P + k2*(lambda*P) by plain scalar multiplications */
static void rand_mul(g1j *out, const g1a *p, const u64 *k, u32 flags) {
  g1j pj, t; g1j_from_affine(&pj, p);
  if (!(flags & FLAG_RAND_GLV)) { g1j_mul(out, &pj, k); return; }
  u64 k1[4] = {k[0], 0, 0, 0}, k2[4] = {k[1], 0, 0, 0};
  g1j lp; g1j_mul(&lp, &pj, GLV_LAMBDA);
  g1j_mul(out, &pj, k1); g1j_mul(&t, &lp, k2); g1j_add(out, out, &t);
}
static void rand_scalar(u64 *k, const u8 *seed32, u64 i, u32 flags) {
  u8 buf[40], dg[32];
  memcpy(buf, seed32, 32);
  for (int b = 0; b < 8; ++b) buf[32 + b] = (u8)(i >> (8 * b));
  sha256_msg_plus_byte(dg, buf, 39, buf[39]);
  k[0] = k[1] = k[2] = k[3] = 0;
  for (int b = 0; b < 8; ++b) k[0] |= (u64)dg[b] << (8 * b);
  if (!(flags & FLAG_RAND64)) for (int b = 0; b < 8; ++b) k[1] |= (u64)dg[8 + b] << (8 * b);
  if ((k[0] | k[1]) == 0) k[0] = 1;
}
API int bn254o_batch_verify_randomized(const u8 *msgs, const u64 *off, const u8 *sigs, const u8 *pks, size_t n, u32 flags,
                                       const u8 *seed32, u8 *status, u8 *group_ok) {
  ensure_init();
  for (size_t g0 = 0; g0 < n; g0 += 64) {
    size_t g1 = g0 + 64 > n ? n : g0 + 64;
    fp12 f = FP12_ONE, part;
    g1a ps[MAX_PAIRS]; g2a qs[MAX_PAIRS]; int np = 0;
    g1j sum; g1j_from_affine(&sum, &(g1a){.inf = 1});
    for (size_t i = g0; i < g1; ++i) {
      g1a sig, h; g2a pk; int st;
      if ((st = decode_g1(&sig, sigs + 64 * i, flags)) == ST_OK && (st = decode_g2(&pk, pks + 128 * i, flags)) == ST_OK)
        st = hash_to_g1(&h, msgs + off[i], (size_t)(off[i + 1] - off[i]), NULL);
      status[i] = (u8)st;
      if (st != ST_OK) continue;
      u64 k[4]; rand_scalar(k, seed32, (u64)i, flags);
      g1j t;
      rand_mul(&t, &h, k, flags); g1j_to_affine(&ps[np], &t); qs[np] = pk; ++np;
      rand_mul(&t, &sig, k, flags); g1j_add(&sum, &sum, &t);
      if (np == MAX_PAIRS) { miller_loop_multi(&part, ps, qs, np); fp12_mul(&f, &f, &part); np = 0; }
    }
    g1j_to_affine(&ps[np], &sum); qs[np] = G2_GEN_NEG; ++np;
    miller_loop_multi(&part, ps, qs, np); fp12_mul(&f, &f, &part);
    fp12 gt; final_exp(&gt, &f);
    int ok = fp12_eq(&gt, &FP12_ONE);
    if (group_ok) group_ok[g0 / 64] = (u8)ok;
    if (!ok)
      for (size_t i = g0; i < g1; ++i)
        if (status[i] == ST_OK)
          status[i] = (u8)bn254o_verify(msgs + off[i], (size_t)(off[i + 1] - off[i]), sigs + 64 * i, pks + 128 * i, flags);
  }
  return 0;
}

/* product of k pairings compared with one: bn::pairing_batch(..) == Gt::one() */
API int bn254o_pairing_check(const u8 *g1s, const u8 *g2s, size_t k, u32 flags) {
  ensure_init();
  if (k > MAX_PAIRS) return ST_INVALID_LENGTH;
  g1a ps[MAX_PAIRS]; g2a qs[MAX_PAIRS]; int st;
  for (size_t i = 0; i < k; ++i) {
    if ((st = decode_g1(&ps[i], g1s + 64 * i, flags)) != ST_OK) return st;
    if ((st = decode_g2(&qs[i], g2s + 128 * i, flags)) != ST_OK) return st;
  }
  fp12 gt; pairing_product(&gt, ps, qs, (int)k);
  return fp12_eq(&gt, &FP12_ONE) ? ST_OK : ST_VERIFICATION_FAILED;
}
/* canonical Gt = prod e(P_i,Q_i) as 384 bytes (tower order, big-endian words) */
API int bn254o_pairing(const u8 *g1s, const u8 *g2s, size_t k, u32 flags, u8 *gt384) {
  ensure_init();
  if (k > MAX_PAIRS) return ST_INVALID_LENGTH;
  g1a ps[MAX_PAIRS]; g2a qs[MAX_PAIRS]; int st;
  for (size_t i = 0; i < k; ++i) {
    if ((st = decode_g1(&ps[i], g1s + 64 * i, flags)) != ST_OK) return st;
    if ((st = decode_g2(&qs[i], g2s + 128 * i, flags)) != ST_OK) return st;
  }
  fp12 gt; pairing_product(&gt, ps, qs, (int)k); encode_fp12(gt384, &gt);
  return ST_OK;
}
/* n independent pairing products of k pairs each (BASELINE config 4 checker), sharded over nthreads:
 * gt[i] = canonical bytes, status[i] = decode error, else 0 if the product is one, else 9 */
typedef struct { const u8 *g1s, *g2s; size_t k, lo, hi; u32 flags; u8 *gt, *status; } pairing_job;
static void *pairing_worker(void *arg) {
  pairing_job *j = (pairing_job *)arg;
  for (size_t i = j->lo; i < j->hi; ++i) {
    g1a ps[MAX_PAIRS]; g2a qs[MAX_PAIRS]; int st = ST_OK;
    for (size_t e = 0; e < j->k && st == ST_OK; ++e) {
      if ((st = decode_g1(&ps[e], j->g1s + 64 * (i * j->k + e), j->flags)) != ST_OK) break;
      st = decode_g2(&qs[e], j->g2s + 128 * (i * j->k + e), j->flags);
    }
    if (st != ST_OK) { if (j->gt) memset(j->gt + 384 * i, 0, 384); j->status[i] = (u8)st; continue; }
    fp12 gt; pairing_product(&gt, ps, qs, (int)j->k);
    if (j->gt) encode_fp12(j->gt + 384 * i, &gt);
    j->status[i] = fp12_eq(&gt, &FP12_ONE) ? ST_OK : ST_VERIFICATION_FAILED;
  }
  return NULL;
}
API int bn254o_batch_pairing(const u8 *g1s, const u8 *g2s, size_t n, size_t k, u32 flags, u8 *gt, u8 *status, int nthreads) {
  ensure_init();
  if (k > MAX_PAIRS) return ST_INVALID_LENGTH;
  if (nthreads < 1) nthreads = 1;
  if (nthreads > 256) nthreads = 256;
  pthread_t th[256]; pairing_job jobs[256];
  size_t per = (n + nthreads - 1) / nthreads;
  int used = 0;
  for (int t = 0; t < nthreads; ++t) {
    size_t lo = per * t, hi = lo + per > n ? n : lo + per;
    if (lo >= hi) break;
    jobs[t] = (pairing_job){g1s, g2s, k, lo, hi, flags, gt, status};
    if (nthreads == 1) pairing_worker(&jobs[t]); else pthread_create(&th[t], NULL, pairing_worker, &jobs[t]);
    ++used;
  }
  if (nthreads != 1) for (int t = 0; t < used; ++t) pthread_join(th[t], NULL);
  return ST_OK;
}
/* Aggregate verification over shared pools (BASELINE config 2; restates include/bn254_hip.h:bn254_batch_aggregate_verify):
 * tuple i = message tuple_msg[i] + signer list signer_idx[tuple_off[i] .. tuple_off[i+1]); the aggregate signature /
 * public key are the sums of the listed pool entries (Add for Signature / PublicKey, types.rs:264-270, :126-132), then
 * ECDSA::verify (ecdsa.rs:49-64).  An out-of-range index gives 2 (IndexOutOfBounds); an undecodable pool entry gives its
 * decode status; the first problem in list order wins, signature before public key. */
typedef struct {
  const u8 *msgs; const u64 *msg_off; size_t n_msgs; const u8 *pk_pool; size_t n_signers; const u8 *sig_pool;
  const u32 *tuple_msg; const u64 *tuple_off; const u32 *signer_idx; size_t lo, hi; u32 flags; u8 *status;
} agg_job;
static void *agg_worker(void *arg) {
  agg_job *j = (agg_job *)arg;
  for (size_t i = j->lo; i < j->hi; ++i) {
    int st = ST_OK;
    u32 m = j->tuple_msg[i];
    if (m >= j->n_msgs) { j->status[i] = ST_INDEX_OOB; continue; }
    g1j s1; g2j s2;
    g1j_from_affine(&s1, &(g1a){.inf = 1});
    g2j_from_affine(&s2, &(g2a){.inf = 1});
    for (u64 t = j->tuple_off[i]; t < j->tuple_off[i + 1] && st == ST_OK; ++t) {
      u32 sg = j->signer_idx[t];
      if (sg >= j->n_signers) { st = ST_INDEX_OOB; break; }
      g1a sa; g2a pa;
      if ((st = decode_g1(&sa, j->sig_pool + 64 * ((size_t)m * j->n_signers + sg), j->flags)) != ST_OK) break;
      if ((st = decode_g2(&pa, j->pk_pool + 128 * (size_t)sg, j->flags)) != ST_OK) break;
      g1j tj; g2j uj;
      g1j_from_affine(&tj, &sa); g1j_add(&s1, &s1, &tj);
      g2j_from_affine(&uj, &pa); g2j_add(&s2, &s2, &uj);
    }
    if (st != ST_OK) { j->status[i] = (u8)st; continue; }
    g1a ps[2]; g2a qs[2];
    g1j_to_affine(&ps[1], &s1); g2j_to_affine(&qs[0], &s2);
    if ((st = hash_to_g1(&ps[0], j->msgs + j->msg_off[m], (size_t)(j->msg_off[m + 1] - j->msg_off[m]), NULL)) != ST_OK) { j->status[i] = (u8)st; continue; }
    qs[1] = G2_GEN_NEG;
    fp12 gt; pairing_product(&gt, ps, qs, 2);
    j->status[i] = fp12_eq(&gt, &FP12_ONE) ? ST_OK : ST_VERIFICATION_FAILED;
  }
  return NULL;
}
API int bn254o_batch_aggregate_verify(const u8 *msgs, const u64 *msg_off, size_t n_msgs, const u8 *pk_pool, size_t n_signers, const u8 *sig_pool,
                                      const u32 *tuple_msg, const u64 *tuple_off, const u32 *signer_idx, size_t n, u32 flags, u8 *status, int nthreads) {
  ensure_init();
  if (nthreads < 1) nthreads = 1;
  if (nthreads > 256) nthreads = 256;
  pthread_t th[256]; agg_job jobs[256];
  size_t per = (n + nthreads - 1) / nthreads;
  int used = 0;
  for (int t = 0; t < nthreads; ++t) {
    size_t lo = per * t, hi = lo + per > n ? n : lo + per;
    if (lo >= hi) break;
    jobs[t] = (agg_job){msgs, msg_off, n_msgs, pk_pool, n_signers, sig_pool, tuple_msg, tuple_off, signer_idx, lo, hi, flags, status};
    if (nthreads == 1) agg_worker(&jobs[t]); else pthread_create(&th[t], NULL, agg_worker, &jobs[t]);
    ++used;
  }
  if (nthreads != 1) for (int t = 0; t < used; ++t) pthread_join(th[t], NULL);
  return ST_OK;
}
/* the try loop's treatment of ONE digest value h (32 B big-endian): 1 + point if it yields one, else 0
 * (hash.rs:44-59 incl. mod_u256 of utils.rs:27-37; no SHA-256 preimage of h = k*q exists, hence this hook) */
API int bn254o_hash_candidate(const u8 *h32, u8 *out64) {
  ensure_init();
  u64 h[4]; be32_to_u256(h, h32);
  g1a p;
  if (!hash_candidate_point(&p, h)) { memset(out64, 0, 64); return 0; }
  encode_g1(out64, &p);
  return 1;
}
/* un-exponentiated Miller-loop value (for kernel-by-kernel debugging of the HIP path) */
API int bn254o_miller_loop(const u8 *g1s, const u8 *g2s, size_t k, u8 *f384) {
  ensure_init();
  if (k > MAX_PAIRS) return ST_INVALID_LENGTH;
  g1a ps[MAX_PAIRS]; g2a qs[MAX_PAIRS]; int st;
  for (size_t i = 0; i < k; ++i) {
    if ((st = decode_g1(&ps[i], g1s + 64 * i, 0)) != ST_OK) return st;
    if ((st = decode_g2(&qs[i], g2s + 128 * i, 0)) != ST_OK) return st;
  }
  fp12 f; miller_loop_multi(&f, ps, qs, (int)k); encode_fp12(f384, &f);
  return ST_OK;
}

static void scalar_from_be(u64 *k, const u8 *b) { be32_to_u256(k, b); }

API int bn254o_g1_add(const u8 *a64, const u8 *b64, u8 *out64) {
  ensure_init();
  g1a a, b, o; int st;
  if ((st = decode_g1(&a, a64, 0)) != ST_OK) return st;
  if ((st = decode_g1(&b, b64, 0)) != ST_OK) return st;
  g1j ja, jb, jo; g1j_from_affine(&ja, &a); g1j_from_affine(&jb, &b); g1j_add(&jo, &ja, &jb); g1j_to_affine(&o, &jo);
  encode_g1(out64, &o); return ST_OK;
}
API int bn254o_g1_mul(const u8 *p64, const u8 *scalar32, u8 *out64) {
  ensure_init();
  g1a a, o; int st; u64 k[4];
  if ((st = decode_g1(&a, p64, 0)) != ST_OK) return st;
  scalar_from_be(k, scalar32);
  g1j ja, jo; g1j_from_affine(&ja, &a); g1j_mul(&jo, &ja, k); g1j_to_affine(&o, &jo);
  encode_g1(out64, &o); return ST_OK;
}
API int bn254o_g2_add(const u8 *a128, const u8 *b128, u8 *out128) {
  ensure_init();
  g2a a, b, o; int st;
  if ((st = decode_g2(&a, a128, 0)) != ST_OK) return st;
  if ((st = decode_g2(&b, b128, 0)) != ST_OK) return st;
  g2j ja, jb, jo; g2j_from_affine(&ja, &a); g2j_from_affine(&jb, &b); g2j_add(&jo, &ja, &jb); g2j_to_affine(&o, &jo);
  encode_g2(out128, &o); return ST_OK;
}
API int bn254o_g2_mul(const u8 *p128, const u8 *scalar32, u8 *out128) {
  ensure_init();
  g2a a, o; int st; u64 k[4];
  if ((st = decode_g2(&a, p128, 0)) != ST_OK) return st;
  scalar_from_be(k, scalar32);
  g2j ja, jo; g2j_from_affine(&ja, &a); g2j_mul(&jo, &ja, k); g2j_to_affine(&o, &jo);
  encode_g2(out128, &o); return ST_OK;
}
API void bn254o_g1_generator(u8 *out64) { ensure_init(); encode_g1(out64, &G1_GEN); }
API void bn254o_g2_generator(u8 *out128) { ensure_init(); encode_g2(out128, &G2_GEN); }
/* decode with the reference's validation (flags as in verify) -> status */
API int bn254o_g1_validate(const u8 *p64, u32 flags) { ensure_init(); g1a a; return decode_g1(&a, p64, flags); }
API int bn254o_g2_validate(const u8 *p128, u32 flags) { ensure_init(); g2a a; return decode_g2(&a, p128, flags); }

/* ECDSA::sign, ecdsa.rs:26-35; sk: 32 bytes big-endian, reduced mod r like Fr::from_slice */
API int bn254o_sign(const u8 *msg, size_t len, const u8 *sk32, u8 *sig64) {
  ensure_init();
  g1a h, o; int st;
  if ((st = hash_to_g1(&h, msg, len, NULL)) != ST_OK) return st;
  u64 k[4]; scalar_from_be(k, sk32);
  while (u256_geq(k, ORDER_R)) u256_sub(k, k, ORDER_R);
  g1j jh, jo; g1j_from_affine(&jh, &h); g1j_mul(&jo, &jh, k); g1j_to_affine(&o, &jo);
  encode_g1(sig64, &o); return ST_OK;
}
/* PublicKey::from_private_key (G2) and PublicKeyG1::from_private_key, types.rs:85-87,155-157 */
API void bn254o_public_key_g2(const u8 *sk32, u8 *pk128) {
  ensure_init();
  u64 k[4]; scalar_from_be(k, sk32);
  while (u256_geq(k, ORDER_R)) u256_sub(k, k, ORDER_R);
  g2j jg, jo; g2a o; g2j_from_affine(&jg, &G2_GEN); g2j_mul(&jo, &jg, k); g2j_to_affine(&o, &jo); encode_g2(pk128, &o);
}
API void bn254o_public_key_g1(const u8 *sk32, u8 *pk64) {
  ensure_init();
  u64 k[4]; scalar_from_be(k, sk32);
  while (u256_geq(k, ORDER_R)) u256_sub(k, k, ORDER_R);
  g1j jg, jo; g1a o; g1j_from_affine(&jg, &G1_GEN); g1j_mul(&jo, &jg, k); g1j_to_affine(&o, &jo); encode_g1(pk64, &o);
}
/* G1 compressed codec, utils.rs:84-104 and bn::G1::from_compressed */
API int bn254o_g1_compress(const u8 *p64, u8 *out33) {
  ensure_init();
  if (all_zero(p64, 64)) return ST_POINT_IN_JACOBIAN;
  out33[0] = (p64[63] & 1) ? 3 : 2; memcpy(out33 + 1, p64, 32); return ST_OK;
}
API int bn254o_g1_decompress(const u8 *in33, u8 *out64) {
  ensure_init();
  fp x, y, rhs;
  /* order of bn::G1::from_compressed (oracle/bn254_model.py: g1_from_compressed): range, square root, then the prefix */
  if (!fp_from_be(&x, in33 + 1)) return ST_NOT_MEMBER;
  fp_sqr(&rhs, &x); fp_mul(&rhs, &rhs, &x); fp_add(&rhs, &rhs, &FP_B);
  if (!fp_sqrt(&y, &rhs)) return ST_NOT_MEMBER;
  if (in33[0] != 2 && in33[0] != 3) return ST_INVALID_ENCODING;
  u64 yi[4]; fp_to_u256(yi, &y);
  if ((int)(yi[0] & 1) != (in33[0] == 3)) fp_neg(&y, &y);
  fp_to_be(out64, &x); fp_to_be(out64 + 32, &y); return ST_OK;
}
