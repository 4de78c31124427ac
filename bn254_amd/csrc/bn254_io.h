// Byte formats of the batch ABI <-> in-register field elements, and the per-item decoders.
//
// Formats are exactly the reference's *uncompressed* encodings:
//   G1: x || y, 32-byte big-endian each            (/root/reference/src/utils.rs:182-194)
//   G2: x.re || x.im || y.re || y.im, BE32 each     (/root/reference/src/utils.rs:161-179)
// Decoders follow from_uncompressed_to_g1/g2 (/root/reference/src/utils.rs:107-127):
//   coordinate >= q -> NotMemberError(6); not on the curve (or, for G2 with flag bit0, not in the
//   order-r subgroup) -> InvalidGroupPoint(4).  All-zero bytes denote the identity, which the
//   reference's typed API can hold but not encode; flag bit1 makes the
//   decoder reject it like from_uncompressed would.
#pragma once
#include "bn254_curve.h"

namespace bn254 {

// 32 big-endian bytes (4-byte aligned) -> plain limbs
BN_DEV void u256_from_be(uint32_t* v, const uint8_t* p) {
  const uint32_t* w = (const uint32_t*)p;
#pragma unroll
  for (int i = 0; i < 8; ++i) v[7 - i] = __builtin_bswap32(w[i]);
}
BN_DEV void u256_to_be(uint8_t* p, const uint32_t* v) {
  uint32_t* w = (uint32_t*)p;
#pragma unroll
  for (int i = 0; i < 8; ++i) w[i] = __builtin_bswap32(v[7 - i]);
}
// Fq::from_slice: value must be < q.  Returns false otherwise.  `any` ORs in the raw bits.
BN_DEV bool fp_from_be(Fp& r, const uint8_t* p, uint32_t& any) {
  U256 t;
  u256_from_be(t.w, p);
#pragma unroll
  for (int i = 0; i < 8; ++i) any |= t.w[i];
  bool ok = !u256_geq(t.w, C_Q);
  r = fp_from_u256(t);
  return ok;
}
BN_DEV void fp_to_be(uint8_t* p, const Fp& a) {
  U256 t = fp_to_u256(a);
  u256_to_be(p, t.w);
}

BN_DEV uint8_t decode_g1(G1Affine& pt, const uint8_t* b, uint32_t flags) {
  uint32_t any = 0;
  bool ok = fp_from_be(pt.x, b, any);
  ok = fp_from_be(pt.y, b + 32, any) && ok;
  pt.inf = (any == 0);
  if (pt.inf) return (flags & FLAG_REJECT_IDENTITY) ? ST_INVALID_GROUP_POINT : ST_OK;
  if (!ok) return ST_NOT_MEMBER;
  return g1_on_curve(pt) ? ST_OK : ST_INVALID_GROUP_POINT;
}
BN_DEV uint8_t decode_g2(G2Affine& pt, const uint8_t* b, uint32_t flags) {
  uint32_t any = 0;
  bool ok = fp_from_be(pt.x.c0, b, any);
  ok = fp_from_be(pt.x.c1, b + 32, any) && ok;
  ok = fp_from_be(pt.y.c0, b + 64, any) && ok;
  ok = fp_from_be(pt.y.c1, b + 96, any) && ok;
  pt.inf = (any == 0);
  if (pt.inf) return (flags & FLAG_REJECT_IDENTITY) ? ST_INVALID_GROUP_POINT : ST_OK;
  if (!ok) return ST_NOT_MEMBER;
  if (!g2_on_curve(pt)) return ST_INVALID_GROUP_POINT;
  return ST_OK;   // the (expensive, wave-uniform) subgroup check is issued by the caller
}
BN_DEV void encode_g1(uint8_t* b, const G1Affine& p) {
  Fp zx = p.inf ? fp_zero() : p.x, zy = p.inf ? fp_zero() : p.y;
  fp_to_be(b, zx); fp_to_be(b + 32, zy);
}
BN_DEV void encode_g2(uint8_t* b, const G2Affine& p) {
  Fp2 zx = p.inf ? fp2_zero() : p.x, zy = p.inf ? fp2_zero() : p.y;
  fp_to_be(b, zx.c0); fp_to_be(b + 32, zx.c1); fp_to_be(b + 64, zy.c0); fp_to_be(b + 96, zy.c1);
}
// canonical Gt bytes of this build: 12 x BE32 in tower order a0.re a0.im a1.re ... b2.im
BN_DEV void encode_fp12(uint8_t* b, const Fp12& f) {
  fp_to_be(b + 0, f.c0.c0.c0);   fp_to_be(b + 32, f.c0.c0.c1);
  fp_to_be(b + 64, f.c0.c1.c0);  fp_to_be(b + 96, f.c0.c1.c1);
  fp_to_be(b + 128, f.c0.c2.c0); fp_to_be(b + 160, f.c0.c2.c1);
  fp_to_be(b + 192, f.c1.c0.c0); fp_to_be(b + 224, f.c1.c0.c1);
  fp_to_be(b + 256, f.c1.c1.c0); fp_to_be(b + 288, f.c1.c1.c1);
  fp_to_be(b + 320, f.c1.c2.c0); fp_to_be(b + 352, f.c1.c2.c1);
}
// ---- compressed encodings (/root/reference/src/utils.rs:84-104, :130-158) ----
// G1 (33 B): 0x02 (y even) / 0x03 (y odd) || x BE32.   bn::G1::from_compressed as used at
// /root/reference/src/types.rs:234 and src/utils.rs:60, in the order that decoder works: x >= q -> NotMemberError(6)
// (Fq::from_slice), no square root -> NotMemberError(6), and only then a prefix other than 0x02 / 0x03 ->
// InvalidEncoding(3) — an input with two faults reports the first in this order (oracle/bn254_model.py:
// g1_from_compressed).  Byte loads: the 33-byte stride is unaligned.
BN_DEV uint8_t decompress_g1(G1Affine& pt, const uint8_t* b) {
  uint8_t sign = b[0];
  U256 xw;
#pragma unroll
  for (int w = 0; w < 8; ++w) {
    const uint8_t* p = b + 1 + 4 * (7 - w);
    xw.w[w] = ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
  }
  bool ok_sign = sign == 2 || sign == 3;
  bool in_range = !u256_geq(xw.w, C_Q);
  Fp x = fp_from_u256(xw);
  Fp rhs = fp_add(fp_mul(fp_sqr(x), x), fp_load_const(C_THREE));
  Fp y;
  bool has_root = fp_sqrt(y, rhs);
  U256 yp = fp_to_u256(y);
  bool odd = yp.w[0] & 1;
  if (odd != (sign == 3)) y = fp_neg(y);
  pt.x = x; pt.y = y; pt.inf = false;
  if (!in_range || !has_root) return ST_NOT_MEMBER;
  if (!ok_sign) return ST_INVALID_ENCODING;
  return ST_OK;
}
}  // namespace bn254
#include "bn254_codec_g2.h"   // decompress_g2: written against the fp2_* interface, shared with the pair layout
namespace bn254 {

// 32-byte big-endian scalar -> plain limbs; `reduce`: bring into [0, r) like Fr::from_slice
// (/root/reference/src/types.rs:36-38; examples/bn254.rs:7-12 loads keys > r)
BN_DEV void scalar_from_be(uint32_t* k, const uint8_t* p, bool reduce) {
  u256_from_be(k, p);
  if (!reduce) return;
  for (int it = 0; it < 6; ++it) {   // 2^256 / r < 6
    if (u256_geq(k, C_ORDER_R)) {
      uint32_t bw = 0;
      for (int i = 0; i < 8; ++i) {
        uint64_t d = (uint64_t)k[i] - C_ORDER_R[i] - bw;
        k[i] = (uint32_t)d; bw = (uint32_t)(d >> 63);
      }
    }
  }
}

}  // namespace bn254
