// SHA-256 and the try-and-increment hash-to-G1 of the reference, one message per lane.
//
// Follows /root/reference/src/hash.rs:29-63 step by step (see hash_to_g1 below); the SHA-256
// is the `sha2::Sha256::digest` call at /root/reference/src/hash.rs:42, restated from FIPS 180-4.
// Byte/status formats: include/bn254_hip.h.
#pragma once
#include "bn254_curve.h"

namespace bn254 {

BN_CONST uint32_t C_SHA_K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,
    0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,
    0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,
    0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,
    0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,
    0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

BN_DEV uint32_t rotr32(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

BN_DEVN void sha256_compress(uint32_t* h, const uint32_t* blk) {
  uint32_t w[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) w[i] = blk[i];
  uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
#pragma unroll
  for (int i = 0; i < 64; ++i) {
    uint32_t wi;
    if (i < 16) {
      wi = w[i];
    } else {
      uint32_t w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15];
      uint32_t s0 = rotr32(w15, 7) ^ rotr32(w15, 18) ^ (w15 >> 3);
      uint32_t s1 = rotr32(w2, 17) ^ rotr32(w2, 19) ^ (w2 >> 10);
      wi = w[i & 15] + s0 + w[(i - 7) & 15] + s1;
      w[i & 15] = wi;
    }
    uint32_t t1 = hh + (rotr32(e, 6) ^ rotr32(e, 11) ^ rotr32(e, 25)) + ((e & f) ^ (~e & g)) + C_SHA_K[i] + wi;
    uint32_t t2 = (rotr32(a, 2) ^ rotr32(a, 13) ^ rotr32(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
    hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
  }
  h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}

// byte `pos` of the padded stream  msg || ctr || 0x80 || 0.. || be64(8*(len+1))
BN_DEV uint32_t padded_byte(const uint8_t* msg, uint64_t len, uint32_t ctr, uint64_t pos, uint64_t padded_len) {
  if (pos < len) return msg[pos];
  if (pos == len) return ctr;
  if (pos == len + 1) return 0x80;
  if (pos >= padded_len - 8) {
    uint64_t bits = (len + 1) * 8;
    return (uint32_t)(bits >> (8 * (padded_len - 1 - pos))) & 0xFF;
  }
  return 0;
}
BN_DEV void load_block(uint32_t* blk, const uint8_t* msg, uint64_t len, uint32_t ctr, uint64_t b, uint64_t padded_len) {
  for (int i = 0; i < 16; ++i) {
    uint64_t p = b * 64 + 4 * (uint64_t)i;
    blk[i] = (padded_byte(msg, len, ctr, p, padded_len) << 24) | (padded_byte(msg, len, ctr, p + 1, padded_len) << 16) |
             (padded_byte(msg, len, ctr, p + 2, padded_len) << 8) | padded_byte(msg, len, ctr, p + 3, padded_len);
  }
}

struct HashState {
  uint32_t mid[8];       // SHA-256 state after the blocks that do not contain the counter byte
  uint64_t first_block;  // index of the block holding the counter byte
  uint64_t n_blocks;     // total blocks of the padded stream
  uint64_t padded_len;
};
BN_DEV void hash_state_init(HashState& s, const uint8_t* msg, uint64_t len) {
  const uint32_t iv[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
  for (int i = 0; i < 8; ++i) s.mid[i] = iv[i];
  s.n_blocks = (len + 1 + 1 + 8 + 63) / 64;
  s.padded_len = s.n_blocks * 64;
  s.first_block = len / 64;
  uint32_t blk[16];
  for (uint64_t b = 0; b < s.first_block; ++b) {
    load_block(blk, msg, len, 0, b, s.padded_len);
    sha256_compress(s.mid, blk);
  }
}

// Is a (0 <= a < q, plain integer) a square mod q?  Jacobi symbol (a/q) by the binary algorithm — shifts,
// subtractions and the quadratic-reciprocity sign rules, no multiplications: ~190-250 passes (every pass strips ALL trailing
// zeros of a, applies (2/n) once for the strip, subtracts) against the ~370 Montgomery products of the square-root
// exponentiation it guards.  Zero counts as a square (its root is 0), exactly like fp_sqrt.
// The operands shrink by ~2 bits per pass, so the passes run in PHASES of 8, 6, 4 and 2 words: a phase ends when every lane of
// the wave that is still working has both operands inside the next narrower width (one wave vote per pass) — on average the
// multiword subtractions, shifts and selects touch little more than half of the eight words.
#if defined(__HIPCC__)
#define BN_WAVE_ALL(x) (__all((int)(x)) != 0)
#else
#define BN_WAVE_ALL(x) (x)
#endif
// d = a - b over W words, returns the borrow as a mask (0 or ~0).  On the device the borrow chain is spelled out (v_sub_co_u32 /
// v_subb_co_u32 through VCC): the compiler's own lowering of the 64-bit emulation takes five instructions per word and a branch per
// pass (~150 instructions per pass where this takes ~55).
template <int W> BN_DEV uint32_t u32xw_sub(uint32_t* d, const uint32_t* a, const uint32_t* b) {
  uint32_t mask;
#if defined(__HIP_DEVICE_COMPILE__)
  static_assert(W == 2 || W == 4 || W == 6 || W == 8, "phases of the Jacobi passes");
  if constexpr (W == 8) {
    asm("v_sub_co_u32 %0, vcc, %9, %17\n\tv_subb_co_u32 %1, vcc, %10, %18, vcc\n\tv_subb_co_u32 %2, vcc, %11, %19, vcc\n\t"
        "v_subb_co_u32 %3, vcc, %12, %20, vcc\n\tv_subb_co_u32 %4, vcc, %13, %21, vcc\n\tv_subb_co_u32 %5, vcc, %14, %22, vcc\n\t"
        "v_subb_co_u32 %6, vcc, %15, %23, vcc\n\tv_subb_co_u32 %7, vcc, %16, %24, vcc\n\tv_subb_co_u32 %8, vcc, 0, 0, vcc"
        : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6]), "=&v"(d[7]), "=&v"(mask)
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]),
          "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7])
        : "vcc");
  } else if constexpr (W == 6) {
    asm("v_sub_co_u32 %0, vcc, %7, %13\n\tv_subb_co_u32 %1, vcc, %8, %14, vcc\n\tv_subb_co_u32 %2, vcc, %9, %15, vcc\n\t"
        "v_subb_co_u32 %3, vcc, %10, %16, vcc\n\tv_subb_co_u32 %4, vcc, %11, %17, vcc\n\tv_subb_co_u32 %5, vcc, %12, %18, vcc\n\t"
        "v_subb_co_u32 %6, vcc, 0, 0, vcc"
        : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(mask)
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5])
        : "vcc");
  } else if constexpr (W == 4) {
    asm("v_sub_co_u32 %0, vcc, %5, %9\n\tv_subb_co_u32 %1, vcc, %6, %10, vcc\n\tv_subb_co_u32 %2, vcc, %7, %11, vcc\n\t"
        "v_subb_co_u32 %3, vcc, %8, %12, vcc\n\tv_subb_co_u32 %4, vcc, 0, 0, vcc"
        : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(mask)
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3])
        : "vcc");
  } else {
    asm("v_sub_co_u32 %0, vcc, %3, %5\n\tv_subb_co_u32 %1, vcc, %4, %6, vcc\n\tv_subb_co_u32 %2, vcc, 0, 0, vcc"
        : "=&v"(d[0]), "=&v"(d[1]), "=&v"(mask)
        : "v"(a[0]), "v"(a[1]), "v"(b[0]), "v"(b[1])
        : "vcc");
  }
#else
  uint32_t bw = 0;
  for (int i = 0; i < W; ++i) {
    const uint64_t x = (uint64_t)a[i] - b[i] - bw;
    d[i] = (uint32_t)x; bw = (uint32_t)(x >> 63);
  }
  mask = 0u - bw;
#endif
  return mask;
}
template <int W> BN_DEV void jacobi_passes(uint32_t* a, uint32_t* n, uint32_t& t, uint32_t& any, int& budget) {
  while (budget > 0) {
    const bool act = any != 0;
    if (!BN_WAVE_ANY(act)) return;
    if constexpr (W > 2) {
      if (BN_WAVE_ALL(!act || (a[W - 1] | a[W - 2] | n[W - 1] | n[W - 2]) == 0)) return;      // everyone fits W - 2 words
    }
    --budget;                                          // each pass removes >= 1 bit of |a| + |n| (<= 508)
    if (!act) continue;
    if (a[0] == 0) {                                   // 32 trailing zeros: an even number of halvings, no sign change
#pragma unroll
      for (int i = 0; i < W - 1; ++i) a[i] = a[i + 1];
      a[W - 1] = 0;
      continue;
    }
    const uint32_t s = (uint32_t)__builtin_ctz(a[0]);
    // (2/n) = -1 iff n = 3, 5 (mod 8); applied s times
    t ^= s & ((n[0] >> 1) ^ (n[0] >> 2));
    uint32_t o[W];                                     // a with its trailing zeros stripped: odd
#pragma unroll
    for (int i = 0; i < W - 1; ++i) o[i] = (uint32_t)((((uint64_t)a[i + 1] << 32) | a[i]) >> s);        // one v_alignbit_b32
    o[W - 1] = a[W - 1] >> s;
    // d = o - n; o < n (borrow): swap (reciprocity: flip iff both = 3 mod 4) and take n - o = -d instead.  -d = (d ^ m) - m with m = ~0
    uint32_t d[W], x[W], mz[W];
    const uint32_t m = u32xw_sub<W>(d, o, n);
    t ^= m & ((o[0] & n[0]) >> 1);
#pragma unroll
    for (int i = 0; i < W; ++i) { x[i] = d[i] ^ m; mz[i] = m; }
    (void)u32xw_sub<W>(a, x, mz);
    any = 0;
#pragma unroll
    for (int i = 0; i < W; ++i) {
      n[i] = m ? o[i] : n[i];
      any |= a[i];
    }
  }
}
BN_DEVN bool u256_is_square_mod_q(const U256& a_in) {
  uint32_t a[8], n[8];
  uint32_t any = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = a_in.w[i]; n[i] = C_Q[i]; any |= a[i]; }
  const bool zero = any == 0;
  uint32_t t = 0;                                      // bit 0: parity of the sign flips
  int budget = 600;
  jacobi_passes<8>(a, n, t, any, budget);
  jacobi_passes<6>(a, n, t, any, budget);
  jacobi_passes<4>(a, n, t, any, budget);
  jacobi_passes<2>(a, n, t, any, budget);
  // gcd(a_in, q) = n = 1 for a prime q and 0 < a_in < q
  return zero || (t & 1u) == 0;
}

// The range rules applied to a digest value x (hash.rs:49-54): rejected if >= 5q, reduced by mod_u256's strict rule;
// false = this value yields no x.  (Separate from the SHA-256 so that tests can feed chosen values: no preimage of
// k*q exists — bn254_debug_hash_candidate.)
BN_DEV bool hash_reduce_candidate(U256& x) {
  if (u256_geq(x.w, C_QMULT[4])) return false;                   // hash.rs:49-51: h >= 5q -> next ctr
  // utils.rs:27-37 mod_u256: while x > q { x -= q } (strict), i.e. x mod q except exact multiples
  // k*q (k >= 1), which stop at q and are then rejected by Fq::from_slice
  bool was_reduced = false;
  for (int k = 3; k >= 0; --k) {
    if (!was_reduced && u256_geq(x.w, C_QMULT[k])) {
      uint32_t bw = 0;
      for (int i = 0; i < 8; ++i) {
        uint64_t d = (uint64_t)x.w[i] - C_QMULT[k][i] - bw;
        x.w[i] = (uint32_t)d; bw = (uint32_t)(d >> 63);
      }
      was_reduced = true;
    }
  }
  uint32_t any = 0;
  for (int i = 0; i < 8; ++i) any |= x.w[i];
  return !(was_reduced && any == 0);
}
// The candidate x of counter `ctr` (hash.rs:40-54): SHA-256(msg || ctr) read big-endian, then the range rules
BN_DEV bool hash_candidate(U256& x, const HashState& s, const uint8_t* msg, uint64_t len, uint32_t ctr) {
  uint32_t h[8], blk[16];
  for (int i = 0; i < 8; ++i) h[i] = s.mid[i];
  for (uint64_t b = s.first_block; b < s.n_blocks; ++b) {        // hash.rs:41-42  SHA256(msg || ctr)
    load_block(blk, msg, len, ctr, b, s.padded_len);
    sha256_compress(h, blk);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) x.w[i] = h[7 - i];                 // hash.rs:44: digest read big-endian
  return hash_reduce_candidate(x);
}
// x^3 + 3 for a candidate (utils.rs:56-63 arbitrary_string_to_g1 -> G1::from_compressed(0x02 || x))
BN_DEV void hash_curve_rhs(Fp& xm, Fp& rhs, const U256& x) {
  xm = fp_from_u256(x);
  rhs = fp_add(fp_mul(fp_sqr(xm), xm), fp_load_const(C_THREE));
}
// Cheap test of counter `ctr`: does it yield a point?  (candidate exists and x^3 + 3 is a square)
BN_DEV bool hash_try_filter(const HashState& s, const uint8_t* msg, uint64_t len, uint32_t ctr) {
  U256 x;
  if (!hash_candidate(x, s, msg, len, ctr)) return false;
  Fp xm, rhs;
  hash_curve_rhs(xm, rhs, x);
  return u256_is_square_mod_q(fp_to_u256(rhs));
}
// A reduced candidate x -> the point with the even root (utils.rs:56-63), false if x^3 + 3 is no square
BN_DEV bool hash_point_from_candidate(G1Affine& out, const U256& x) {
  Fp xm, rhs, y;
  hash_curve_rhs(xm, rhs, x);
  if (!fp_sqrt(y, rhs)) return false;
  U256 yp = fp_to_u256(y);
  if (yp.w[0] & 1) y = fp_neg(y);
  out.x = xm; out.y = y; out.inf = false;
  return true;
}
// One try of /root/reference/src/hash.rs:40-59 for counter `ctr`; true iff it yields a point (the even root).
BN_DEV bool hash_try(G1Affine& out, const HashState& s, const uint8_t* msg, uint64_t len, uint32_t ctr) {
  U256 x;
  if (!hash_candidate(x, s, msg, len, ctr)) return false;
  return hash_point_from_candidate(out, x);
}

// Scalar of item i for the randomised batch verification (include/bn254_hip.h:
// bn254_batch_verify_randomized): the first 16 bytes (rand64: 8) of SHA-256(seed || le64(i)) read as a
// little-endian integer, 0 replaced by 1.  seed_be = the 32 seed bytes as big-endian words.
BN_DEV void rand_scalar(uint32_t* k, const uint32_t* seed_be, uint64_t i, bool rand64) {
  uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
  uint32_t blk[16];
  for (int j = 0; j < 8; ++j) blk[j] = seed_be[j];
  blk[8] = __builtin_bswap32((uint32_t)i);
  blk[9] = __builtin_bswap32((uint32_t)(i >> 32));
  blk[10] = 0x80000000u;
  for (int j = 11; j < 15; ++j) blk[j] = 0;
  blk[15] = 40 * 8;
  sha256_compress(h, blk);
  k[0] = __builtin_bswap32(h[0]); k[1] = __builtin_bswap32(h[1]);
  k[2] = rand64 ? 0u : __builtin_bswap32(h[2]); k[3] = rand64 ? 0u : __builtin_bswap32(h[3]);
  if ((k[0] | k[1] | k[2] | k[3]) == 0) k[0] = 1;
}

}  // namespace bn254
