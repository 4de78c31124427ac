// hostsim — TEST INFRASTRUCTURE ONLY.
//
// Compiles the *device* arithmetic headers of the product (bn254_amd/csrc/bn254_{field,curve,
// pairing,hash,io}.h — plain C++ by design) for the host CPU, one item at a time, composed the
// same way the HIP kernels in bn254_amd/csrc/bn254_hip.hip compose them.  The CPU test-suite
// (-m "not gpu", no GPU in the build container) uses it to check the kernels' algorithm source
// against the oracle before any GPU time is spent.  It is NOT a fallback: nothing in bn254_amd/
// loads it, and the product fails loudly without a HIP device (bn254_ctx_create -> -10003).
#include <cstdint>
#include <cstring>

#define BN_COUNT_FP_MUL 1
extern "C" { unsigned long long bn_fp_mul_counter = 0; unsigned long long bn_fp_dual_counter = 0; }
#if defined(BN_TRACK_BOUNDS)
#include "../../bn254_amd/csrc/bn254_norm_sites.h"
extern "C" { signed char bn_site_mode[1024]; unsigned int bn_site_hits[1024]; signed char bn_site_dflt[1024]; int bn_bound_soft = 0; int bn_bound_failed = 0; }
static struct BnSiteInit { BnSiteInit() { for (int i = 0; i < 1024; ++i) bn_site_mode[i] = (signed char)bn_site_override(i); } } bn_site_init_;
#endif

#include "../../bn254_amd/csrc/bn254_hash.h"
#include "../../bn254_amd/csrc/bn254_io.h"
#include "../../bn254_amd/csrc/bn254_pairing.h"

using namespace bn254;

static void set_g1_gen(G1Affine& p) { p.x = fp_load_const(C_G1_GEN[0]); p.y = fp_load_const(C_G1_GEN[1]); p.inf = false; }
static void set_g2_gen(G2Affine& q) { q.x = fp2_load_const(C_G2_GEN[0]); q.y = fp2_load_const(C_G2_GEN[1]); q.inf = false; }

static uint8_t dec_g1(G1Affine& p, const uint8_t* b, uint32_t flags) {
  alignas(4) uint8_t tmp[64];
  memcpy(tmp, b, 64);
  uint8_t st = decode_g1(p, tmp, flags);
  if (st != ST_OK) set_g1_gen(p);
  return st;
}
static uint8_t dec_g2(G2Affine& q, const uint8_t* b, uint32_t flags) {
  alignas(4) uint8_t tmp[128];
  memcpy(tmp, b, 128);
  uint8_t st = decode_g2(q, tmp, flags);
  if (st != ST_OK) set_g2_gen(q);
  if (flags & FLAG_G2_SUBGROUP_CHECK) {
    bool in = g2_in_subgroup(q);
    if (st == ST_OK && !in) { st = ST_INVALID_GROUP_POINT; set_g2_gen(q); }
  }
  return st;
}
extern "C" {
// mirrors k_register_keys (one lane per key): decode with the subgroup check always on, then the 87 lines of the key in the
// c2 = 1 form, canonical limbs.  Returns the registration status; *inf_out = identity key.  tab: 87 x 2 x 2 x 9 words.
int hs_register_key(const uint8_t* pk128, uint32_t flags, int32_t* tab, int* inf_out) {
  G2Affine q;
  alignas(4) uint8_t tmp[128];
  memcpy(tmp, pk128, 128);
  uint8_t st = decode_g2(q, tmp, flags & FLAG_REJECT_IDENTITY);
  const bool inf = q.inf;
  if (st != ST_OK || q.inf) { set_g2_gen(q); q.inf = inf; }
  const bool in = g2_in_subgroup(q);
  if (st == ST_OK && !q.inf && !in) { st = ST_INVALID_GROUP_POINT; set_g2_gen(q); q.inf = inf; }
  const bool ok = g2_line_table(q, [&](int idx, const KeyLine& kl) {
    const Fp c[4] = {fp_canon(kl.c0.c0), fp_canon(kl.c0.c1), fp_canon(kl.c1.c0), fp_canon(kl.c1.c1)};
    for (int e = 0; e < 4; ++e)
      for (int k = 0; k < BN_LIMBS; ++k) tab[(idx * 4 + e) * BN_LIMBS + k] = c[e].v[k];
  });
  if (st == ST_OK && !q.inf && !ok) st = ST_INVALID_GROUP_POINT;
  *inf_out = q.inf;
  return st;
}
}
// mirrors k_hash_round (filter: candidate + Jacobi symbol) and k_hash_finish (one square root, for the winner)
static uint8_t hash_item(G1Affine& p, const uint8_t* msg, uint64_t len, int* tries) {
  HashState hs;
  hash_state_init(hs, msg, len);
  set_g1_gen(p);
  for (uint32_t ctr = 0; ctr < 255; ++ctr) {
    if (!hash_try_filter(hs, msg, len, ctr)) continue;
    if (tries) *tries = (int)ctr + 1;
    if (hash_try(p, hs, msg, len, ctr)) return ST_OK;
    set_g1_gen(p);                       // filter and exponentiation disagree: cannot happen (reported as an error)
    return ST_HASH_TO_POINT;
  }
  if (tries) *tries = 255;
  set_g1_gen(p);
  return ST_HASH_TO_POINT;
}

extern "C" {

unsigned long long hs_fp_mul_count(void) { return bn_fp_mul_counter; }
void hs_fp_mul_count_reset(void) { bn_fp_mul_counter = 0; }

// Montgomery products per stage of one verify, in kernel order: decode(sig)+decode(pk) [no subgroup
// check], hash-to-G1 (this message), Miller loop (2 pairs), final exponentiation + compare
void hs_verify_stage_counts(const uint8_t* msg, uint64_t len, const uint8_t* sig64, const uint8_t* pk128, unsigned long long* out4) {
  G1Affine sig, h;
  G2Affine pk;
  unsigned long long c0 = bn_fp_mul_counter;
  dec_g1(sig, sig64, 0); dec_g2(pk, pk128, 0);
  unsigned long long c1 = bn_fp_mul_counter;
  hash_item(h, msg, len, nullptr);
  unsigned long long c2 = bn_fp_mul_counter;
  Fp12 f;
  miller_loop<true, true>(f, h, pk, sig);
  unsigned long long c3 = bn_fp_mul_counter;
  { Fp12 acc_; final_exponentiation_check(f, f, acc_); }      // what the verify kernels run (status only)
  (void)fp12_is_one(f);
  unsigned long long c4 = bn_fp_mul_counter;
  out4[0] = c1 - c0; out4[1] = c2 - c1; out4[2] = c3 - c2; out4[3] = c4 - c3;
}

int hs_hash_to_g1(const uint8_t* msg, uint64_t len, uint8_t* out64, int* tries) {
  G1Affine p;
  uint8_t st = hash_item(p, msg, len, tries);
  if (st != ST_OK) p.inf = true;
  alignas(4) uint8_t tmp[64];
  encode_g1(tmp, p);
  memcpy(out64, tmp, 64);
  return st;
}

// mirrors k_decode_g1 + k_decode_g2 + k_hash_to_g1 + k_miller_verify + k_final_exp
int hs_verify(const uint8_t* msg, uint64_t len, const uint8_t* sig64, const uint8_t* pk128, uint32_t flags) {
  G1Affine sig, h;
  G2Affine pk;
  uint8_t st = dec_g1(sig, sig64, flags);
  uint8_t s2 = dec_g2(pk, pk128, flags);
  if (st == ST_OK) st = s2;
  uint8_t sh = hash_item(h, msg, len, nullptr);
  Fp12 f;
  miller_loop<true, true>(f, h, pk, sig);
  if (st == ST_OK) st = sh;
  { Fp12 acc_; final_exponentiation_check(f, f, acc_); }
  return st != ST_OK ? st : (fp12_is_one(f) ? ST_OK : ST_VERIFICATION_FAILED);
}

// mirrors bn254_batch_verify_randomized_device: k_decode_* + hash + k_rand_scale + k_miller_rand + k_rand_tail +
// k_final_exp + k_rand_collect + the exact kernels for the items of failed groups (same wave tree reductions)
int hs_verify_randomized(const uint8_t* msgs, const uint64_t* off, const uint8_t* sigs, const uint8_t* pks, uint64_t n, uint32_t flags,
                         const uint8_t* seed32, uint8_t* status, uint8_t* group_ok) {
  uint32_t seed[8];
  for (int j = 0; j < 8; ++j)
    seed[j] = ((uint32_t)seed32[4 * j] << 24) | ((uint32_t)seed32[4 * j + 1] << 16) | ((uint32_t)seed32[4 * j + 2] << 8) | seed32[4 * j + 3];
  const uint32_t dflags = flags & 3u;
  for (uint64_t g0 = 0; g0 < n; g0 += 64) {
    static G1Jac s_lds[64];
    static Fp12 f_lds[64];
    static G1Affine a_ws[64];
    static G2Affine pk_ws[64];
    const bool two_per_lane = (flags & 0x80000000u) != 0;   // test knob: the k_miller_rand2 composition
    uint8_t st[64];
    for (unsigned t = 0; t < 64; ++t) {
      uint64_t i = g0 + t;
      bool live = i < n;
      uint64_t ii = live ? i : n - 1;
      G1Affine sig, h;
      G2Affine pk;
      uint8_t s1 = dec_g1(sig, sigs + 64 * ii, dflags), s2 = dec_g2(pk, pks + 128 * ii, dflags);
      if (s1 == ST_OK) s1 = s2;
      uint8_t sh = hash_item(h, msgs + off[ii], off[ii + 1] - off[ii], nullptr);
      if (s1 == ST_OK) s1 = sh;
      st[t] = s1;
      bool valid = live && s1 == ST_OK;
      uint32_t k[4];
      rand_scalar(k, seed, ii, (flags & 0x100u) != 0);
      G1Jac a, sj, id;
      const bool rand64 = (flags & 0x100u) != 0, glv = !rand64 && (flags & 0x200u) != 0;
      if (glv) g1_mul_glv(a, h, k, k + 2); else if (rand64) jac_mul_u64(a, h, k); else jac_mul_u128(a, h, k);
      G1Affine aa;
      jac_to_affine(aa, a);
      aa.inf = aa.inf || !valid;
      if (glv) g1_mul_glv(sj, sig, k, k + 2); else if (rand64) jac_mul_u64(sj, sig, k); else jac_mul_u128(sj, sig, k);
      jac_set_identity(id);
      jac_select(sj, !valid, id, sj);
      s_lds[t] = sj;
      a_ws[t] = aa; pk_ws[t] = pk;
      if (!two_per_lane) miller_loop<true, false>(f_lds[t], aa, pk, aa);
    }
    if (two_per_lane)
      for (unsigned t = 0; t < 32; ++t) miller_loop_2var(f_lds[t], a_ws[2 * t], pk_ws[2 * t], a_ws[2 * t + 1], pk_ws[2 * t + 1]);
    for (unsigned stride = 32; stride >= 1; stride >>= 1)
      for (unsigned t = 0; t < stride; ++t) {
        jac_add(s_lds[t], s_lds[t], s_lds[t + stride]);
        if (!two_per_lane) fp12_mul(f_lds[t], f_lds[t], f_lds[t + stride]);
        else if (stride <= 16) fp12_mul(f_lds[t], f_lds[t], f_lds[t + stride]);
      }
    G1Affine sa, ug1;
    G2Affine ug2;
    jac_to_affine(sa, s_lds[0]);
    set_g1_gen(ug1); set_g2_gen(ug2);
    Fp12 f;
    miller_loop<false, true>(f, ug1, ug2, sa);
    fp12_mul(f, f, f_lds[0]);
    { Fp12 acc_; final_exponentiation(f, f, acc_); }
    bool ok = fp12_is_one(f);
    if (group_ok) group_ok[g0 / 64] = ok ? 1 : 0;
    for (unsigned t = 0; t < 64 && g0 + t < n; ++t) {
      uint64_t i = g0 + t;
      status[i] = (ok || st[t] != ST_OK) ? st[t] : (uint8_t)hs_verify(msgs + off[i], off[i + 1] - off[i], sigs + 64 * i, pks + 128 * i, dflags);
    }
  }
  return 0;
}

// mirrors pairing_device: k pairs, one Miller loop each, product, final exponentiation
int hs_pairing(const uint8_t* g1s, const uint8_t* g2s, uint64_t k, uint32_t flags, uint8_t* gt384, int raw_only) {
  Fp12 f, g;
  uint8_t st = ST_OK;
  for (uint64_t j = 0; j < k; ++j) {
    G1Affine p;
    G2Affine q;
    uint8_t s1 = dec_g1(p, g1s + 64 * j, flags);
    uint8_t s2 = dec_g2(q, g2s + 128 * j, flags);
    if (s1 == ST_OK) s1 = s2;
    if (st == ST_OK) st = s1;
    miller_loop<true, false>(g, p, q, p);
    if (j == 0) f = g; else fp12_mul(f, f, g);
  }
  if (!raw_only) { Fp12 acc_; final_exponentiation(f, f, acc_); }
  alignas(4) uint8_t tmp[384];
  encode_fp12(tmp, f);
  if (gt384) memcpy(gt384, tmp, 384);
  return st != ST_OK ? st : (fp12_is_one(f) ? ST_OK : ST_VERIFICATION_FAILED);
}

int hs_check_public_keys(const uint8_t* pk_g2, const uint8_t* pk_g1, uint32_t flags) {
  G1Affine pk1, g;
  G2Affine pk2;
  uint8_t st = dec_g2(pk2, pk_g2, flags);
  uint8_t s1 = dec_g1(pk1, pk_g1, flags);
  if (st == ST_OK) st = s1;
  set_g1_gen(g);
  Fp12 f;
  miller_loop<true, true>(f, g, pk2, pk1);
  { Fp12 acc_; final_exponentiation(f, f, acc_); }
  return st != ST_OK ? st : (fp12_is_one(f) ? ST_OK : ST_VERIFICATION_FAILED);
}

int hs_g1_add(const uint8_t* a, const uint8_t* b, uint8_t* out) {
  G1Affine pa, pb, r;
  uint8_t st = dec_g1(pa, a, 0), sb = dec_g1(pb, b, 0);
  if (st == ST_OK) st = sb;
  G1Jac ja, jb, jo;
  jac_from_affine(ja, pa); jac_from_affine(jb, pb);
  jac_add(jo, ja, jb);
  jac_to_affine(r, jo);
  if (st != ST_OK) r.inf = true;
  alignas(4) uint8_t tmp[64];
  encode_g1(tmp, r);
  memcpy(out, tmp, 64);
  return st;
}
int hs_g2_add(const uint8_t* a, const uint8_t* b, uint8_t* out) {
  G2Affine pa, pb, r;
  uint8_t st = dec_g2(pa, a, 0), sb = dec_g2(pb, b, 0);
  if (st == ST_OK) st = sb;
  G2Jac ja, jb, jo;
  jac_from_affine(ja, pa); jac_from_affine(jb, pb);
  jac_add(jo, ja, jb);
  jac_to_affine(r, jo);
  if (st != ST_OK) r.inf = true;
  alignas(4) uint8_t tmp[128];
  encode_g2(tmp, r);
  memcpy(out, tmp, 128);
  return st;
}
// k * P with the 128-bit windowed ladder of the randomised batch verification (k: 16 bytes little-endian)
int hs_g1_mul_u128(const uint8_t* p, const uint8_t* k16, uint8_t* out) {
  G1Affine a, o;
  uint8_t st = dec_g1(a, p, 0);
  if (st != ST_OK) return st;
  uint32_t k[4];
  for (int i = 0; i < 4; ++i) k[i] = (uint32_t)k16[4 * i] | ((uint32_t)k16[4 * i + 1] << 8) | ((uint32_t)k16[4 * i + 2] << 16) | ((uint32_t)k16[4 * i + 3] << 24);
  G1Jac r;
  jac_mul_u128(r, a, k);
  jac_to_affine(o, r);
  alignas(4) uint8_t tmp[64];
  encode_g1(tmp, o);
  memcpy(out, tmp, 64);
  return 0;
}
int hs_g1_mul(const uint8_t* p, const uint8_t* scalar32, int reduce, uint8_t* out) {
  G1Affine pa, r;
  uint8_t st = dec_g1(pa, p, 0);
  alignas(4) uint8_t sc[32];
  memcpy(sc, scalar32, 32);
  uint32_t k[8];
  scalar_from_be(k, sc, reduce != 0);
  G1Jac jo;
  if (reduce & 2) jac_mul(jo, pa, k); else g1_mul_glv_full(jo, pa, k);    // reduce & 2: the plain 256-step ladder (what the kernels used before round 6)
  jac_to_affine(r, jo);
  if (st != ST_OK) r.inf = true;
  alignas(4) uint8_t tmp[64];
  encode_g1(tmp, r);
  memcpy(out, tmp, 64);
  return st;
}
// the GLV decomposition of a scalar in [0, r): k1 and |k2| as 16 little-endian bytes each, returns the sign of k2
int hs_glv_decompose(const uint8_t* k32_be, uint8_t* k1_le16, uint8_t* k2_le16) {
  alignas(4) uint8_t sc[32];
  memcpy(sc, k32_be, 32);
  uint32_t k[8], k1[4], k2[4];
  scalar_from_be(k, sc, true);
  bool neg;
  glv_decompose(k, k1, k2, neg);
  memcpy(k1_le16, k1, 16);
  memcpy(k2_le16, k2, 16);
  return neg ? 1 : 0;
}
int hs_g2_mul(const uint8_t* p, const uint8_t* scalar32, int reduce, uint8_t* out) {
  G2Affine pa, r;
  uint8_t st = ST_OK;
  if (p) st = dec_g2(pa, p, 0); else set_g2_gen(pa);
  alignas(4) uint8_t sc[32];
  memcpy(sc, scalar32, 32);
  uint32_t k[8];
  scalar_from_be(k, sc, reduce != 0);
  G2Jac jo;
  jac_mul(jo, pa, k);
  jac_to_affine(r, jo);
  if (st != ST_OK) r.inf = true;
  alignas(4) uint8_t tmp[128];
  encode_g2(tmp, r);
  memcpy(out, tmp, 128);
  return st;
}
int hs_sign(const uint8_t* msg, uint64_t len, const uint8_t* sk32, uint8_t* sig64) {
  G1Affine h, r;
  uint8_t st = hash_item(h, msg, len, nullptr);
  alignas(4) uint8_t sc[32];
  memcpy(sc, sk32, 32);
  uint32_t k[8];
  scalar_from_be(k, sc, true);
  G1Jac jo;
  g1_mul_glv_full(jo, h, k);
  jac_to_affine(r, jo);
  if (st != ST_OK) r.inf = true;
  alignas(4) uint8_t tmp[64];
  encode_g1(tmp, r);
  memcpy(sig64, tmp, 64);
  return st;
}
int hs_g1_decompress(const uint8_t* in33, uint8_t* out64) {
  G1Affine p;
  uint8_t st = decompress_g1(p, in33);
  if (st != ST_OK) p.inf = true;
  alignas(4) uint8_t tmp[64];
  encode_g1(tmp, p);
  memcpy(out64, tmp, 64);
  return st;
}
int hs_g2_decompress(const uint8_t* in65, uint8_t* out128) {
  G2Affine p;
  uint8_t st = decompress_g2(p, in65);
  if (st != ST_OK) set_g2_gen(p);
  bool in_sub = g2_in_subgroup(p);
  if (st == ST_OK && !in_sub) st = ST_NOT_MEMBER;
  if (st != ST_OK) p.inf = true;
  alignas(4) uint8_t tmp[128];
  encode_g2(tmp, p);
  memcpy(out128, tmp, 128);
  return st;
}
// sum of k points through the mixed-addition ladder used by k_aggregate
int hs_g1_msum(const uint8_t* pts, uint64_t k, uint8_t* out) {
  G1Jac acc;
  jac_set_identity(acc);
  uint8_t st = ST_OK;
  for (uint64_t j = 0; j < k; ++j) { G1Affine p; uint8_t s = dec_g1(p, pts + 64 * j, 0); if (st == ST_OK) st = s; jac_accumulate(acc, p); }
  G1Affine r;
  jac_to_affine(r, acc);
  alignas(4) uint8_t tmp[64];
  encode_g1(tmp, r);
  memcpy(out, tmp, 64);
  return st;
}
int hs_g2_msum(const uint8_t* pts, uint64_t k, uint8_t* out) {
  G2Jac acc;
  jac_set_identity(acc);
  uint8_t st = ST_OK;
  for (uint64_t j = 0; j < k; ++j) { G2Affine p; uint8_t s = dec_g2(p, pts + 128 * j, 0); if (st == ST_OK) st = s; jac_accumulate(acc, p); }
  G2Affine r;
  jac_to_affine(r, acc);
  alignas(4) uint8_t tmp[128];
  encode_g2(tmp, r);
  memcpy(out, tmp, 128);
  return st;
}
// deliberately unsafe sequences: the bound-tracking build must abort on them (tests/test_bounds.py)
int hs_unsafe_sequence(int which) {
  Fp a = fp_one(), b = fp_one();
  if (which == 0) {                 // 40 lazy additions then a product: 10 * (41 T)^2 overflows a 64-bit column
    for (int i = 0; i < 40; ++i) a = fp_add(a, fp_one());
    a = fp_mul(a, a);
  } else if (which == 1) {          // doubling 6 times: limbs leave int32
    for (int i = 0; i < 6; ++i) a = fp_dbl(a);
  } else {                          // xi-multiplication of an un-normalised value, twice
    Fp2 x = fp2_one();
    x = fp2_mul_xi(fp2_mul_xi(fp2_mul_xi(x)));
    a = x.c0;
  }
  return a.v[0] + b.v[0];
}
// op codes as bn254_debug_fp_op
int hs_fp_op(int op, const uint8_t* a, const uint8_t* b, uint8_t* out) {
  alignas(4) uint8_t ta[32], tb[32], to[32];
  memcpy(ta, a, 32);
  if (b) memcpy(tb, b, 32); else memset(tb, 0, 32);
  uint32_t any = 0;
  Fp x, y, r;
  bool ok = fp_from_be(x, ta, any);
  ok = fp_from_be(y, tb, any) && ok;
  uint8_t st = ok ? ST_OK : ST_NOT_MEMBER;
  switch (op) {
    case 0: r = fp_mul(x, y); break;
    case 1: r = fp_add(x, y); break;
    case 2: r = fp_sub(x, y); break;
    case 3: r = fp_inv(x); break;
    case 4: r = fp_sqr(x); break;
    case 6: {                                   // Jacobi-symbol square test of the hash pre-filter
      bool sq = u256_is_square_mod_q(fp_to_u256(x));
      memset(out, 0, 32); out[31] = sq ? 1 : 0;
      return st;
    }
    default: if (!fp_sqrt(r, x) && st == ST_OK) st = ST_NOT_MEMBER; break;
  }
  fp_to_be(to, r);
  memcpy(out, to, 32);
  return st;
}
}  // extern "C"
