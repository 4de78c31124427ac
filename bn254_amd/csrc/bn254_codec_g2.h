// The compressed G2 codec of the reference (65 B: sign || BE64(x.im * q + x.re), /root/reference/src/utils.rs:130-158,
// bn::G2::from_compressed at src/types.rs:92), written against the fp2_* interface only so that it serves both the
// one-lane layout (bn254_io.h) and the pair layout (bn254_pair.hip).
#pragma once
#include "bn254_curve.h"

namespace bn254 {

// U512 (16 words, little-endian) = hi * q + lo with lo < q: restoring long division by q
BN_DEV void u512_divmod_q(U256& hi, U256& lo, bool& hi_overflow, const uint32_t* v) {
  uint32_t rem[9];
  for (int i = 0; i < 9; ++i) rem[i] = 0;
  uint32_t quo[16];
  for (int i = 0; i < 16; ++i) quo[i] = 0;
  for (int bit = 511; bit >= 0; --bit) {
    uint32_t carry = (v[bit >> 5] >> (bit & 31)) & 1;
    for (int i = 0; i < 9; ++i) { uint32_t nc = rem[i] >> 31; rem[i] = (rem[i] << 1) | carry; carry = nc; }
    uint32_t d[9], bw = 0;
    for (int i = 0; i < 9; ++i) {
      uint64_t x = (uint64_t)rem[i] - (i < 8 ? C_Q[i] : 0u) - bw;
      d[i] = (uint32_t)x; bw = (uint32_t)(x >> 63);
    }
    if (!bw) { for (int i = 0; i < 9; ++i) rem[i] = d[i]; quo[bit >> 5] |= 1u << (bit & 31); }
  }
  hi_overflow = false;
  for (int i = 8; i < 16; ++i) hi_overflow = hi_overflow || quo[i] != 0;
  for (int i = 0; i < 8; ++i) { hi.w[i] = quo[i]; lo.w[i] = rem[i]; }
}
// G2 (65 B): sign || BE64(x.im * q + x.re), sign 0x0b iff u512(y) > u512(-y), else 0x0a.
// bn::G2::from_compressed as used at /root/reference/src/types.rs:92, in the order that decoder works: x.im >= q (the
// U512 does not split into two field elements) -> NotMemberError(6); no square root -> NotMemberError(6); a sign byte
// other than 0x0a / 0x0b -> InvalidEncoding(3); not in the order-r subgroup -> NotMemberError(6) (the caller runs the
// wave-uniform subgroup ladder).  An input with two faults reports the first in this order (oracle/bn254_model.py:
// g2_from_compressed).
// The x.im >= q code is UNPINNED: the reference holds no vector for it and zeropool-bn 0.5.11 is not vendored.  Upstream's
// Fq2::from_slice is recalled as `U512::from_slice(..).map_err(InvalidU512Encoding)` (a LENGTH fault, unreachable behind
// the 65-byte check) followed by `divrem`, whose missing quotient goes through `ok_or(FieldError::NotMember)` — and
// FieldError::NotMember is Error::NotMemberError at /root/reference/src/error.rs:44-51.  Rounds 1-2 reported 3 here.
BN_DEV uint8_t decompress_g2(G2Affine& pt, const uint8_t* b) {
  uint8_t sign = b[0];
  uint32_t v[16];
  for (int w = 0; w < 16; ++w) {
    const uint8_t* p = b + 1 + 4 * (15 - w);
    v[w] = ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
  }
  U256 hi, lo;
  bool overflow;
  u512_divmod_q(hi, lo, overflow, v);
  bool split_ok = !overflow && !u256_geq(hi.w, C_Q), sign_ok = sign == 0x0a || sign == 0x0b;
  Fp2 x = fp2_make(fp_from_u256(lo), fp_from_u256(hi));
  Fp2 rhs = fp2_add(fp2_mul(fp2_sqr(x), x), fp2_load_const(C_TWIST_B));
  Fp2 y;
  bool has_root = fp2_sqrt(y, rhs);
  Fp2 yn = fp2_neg(y);
  bool y_gt = fp2_u512_greater(y, yn);
  bool want_gt = sign == 0x0b;
  pt.x = x;
  pt.y = fp2_select(y_gt == want_gt, y, yn);
  pt.inf = false;
  if (!split_ok) return ST_NOT_MEMBER;
  if (!has_root) return ST_NOT_MEMBER;
  if (!sign_ok) return ST_INVALID_ENCODING;
  return ST_OK;
}

}  // namespace bn254
