// hostsim_pair — TEST INFRASTRUCTURE ONLY.
//
// Host compilation of the PAIR layout of the Fq2 tower (bn254_amd/csrc/bn254_fp2_pair.h, the code of
// bn254_pair.hip): an Fq2 element keeps both coefficients and every primitive runs the two lane roles in
// sequence through the same per-role code the device executes with a DPP exchange.  Used by the CPU suite for
// parity against the oracle and, built with -DBN_TRACK_BOUNDS, for the limb / value bound proof of this layout.
#include <cstdint>
#include <cstring>

#define BN_SPLIT_FP2 1
#define BN_COUNT_FP_MUL 1
extern "C" { unsigned long long bn_fp_mul_counter = 0; unsigned long long bn_fp_dual_counter = 0; }
#if defined(BN_TRACK_BOUNDS)
#include "../../bn254_amd/csrc/bn254_norm_sites.h"
extern "C" { signed char bn_site_mode[1024]; unsigned int bn_site_hits[1024]; signed char bn_site_dflt[1024]; int bn_bound_soft = 0; int bn_bound_failed = 0; }
static struct BnSiteInit { BnSiteInit() { for (int i = 0; i < 1024; ++i) bn_site_mode[i] = (signed char)bn_site_override(i); } } bn_site_init_;
#endif

#include "../../bn254_amd/csrc/bn254_pairing.h"
#include "../../bn254_amd/csrc/bn254_codec_g2.h"
#include "../../bn254_amd/csrc/bn254_nonet.h"
#include "../../bn254_amd/csrc/bn254_lmachine.h"

using namespace bn254;

static Fp fp_from_be32(const uint8_t* b) {   // inputs are valid field elements (decoding is tested elsewhere)
  U256 x;
  for (int i = 0; i < 8; ++i) x.w[i] = ((uint32_t)b[28 - 4 * i] << 24) | ((uint32_t)b[29 - 4 * i] << 16) | ((uint32_t)b[30 - 4 * i] << 8) | b[31 - 4 * i];
  return fp_from_u256(x);
}
static void fp_to_be32(uint8_t* b, const Fp& a) {
  U256 x = fp_to_u256(a);
  for (int i = 0; i < 8; ++i) { b[28 - 4 * i] = (uint8_t)(x.w[i] >> 24); b[29 - 4 * i] = (uint8_t)(x.w[i] >> 16); b[30 - 4 * i] = (uint8_t)(x.w[i] >> 8); b[31 - 4 * i] = (uint8_t)x.w[i]; }
}
static bool all_zero(const uint8_t* b, int n) { uint8_t o = 0; for (int i = 0; i < n; ++i) o |= b[i]; return o == 0; }
static void load_g1(G1Affine& p, const uint8_t* b) { p.inf = all_zero(b, 64); p.x = fp_from_be32(b); p.y = fp_from_be32(b + 32); if (p.inf) { p.x = fp_load_const(C_G1_GEN[0]); p.y = fp_load_const(C_G1_GEN[1]); } }
static void load_g2(G2Affine& q, const uint8_t* b) {
  q.inf = all_zero(b, 128);
  if (q.inf) { q.x = fp2_load_const(C_G2_GEN[0]); q.y = fp2_load_const(C_G2_GEN[1]); return; }
  q.x.c[0] = fp_from_be32(b); q.x.c[1] = fp_from_be32(b + 32); q.y.c[0] = fp_from_be32(b + 64); q.y.c[1] = fp_from_be32(b + 96);
}

extern "C" {

unsigned long long hp_fp_mul_count(void) { return bn_fp_mul_counter; }
void hp_fp_mul_count_reset(void) { bn_fp_mul_counter = 0; }

// mirrors k_miller_verify_pair + k_final_exp_pair on decoded inputs: 0 = the pairing product is one, 9 = it is not
int hp_verify_decoded(const uint8_t* h64, const uint8_t* sig64, const uint8_t* pk128) {
  G1Affine h, sig;
  G2Affine pk;
  load_g1(h, h64); load_g1(sig, sig64); load_g2(pk, pk128);
  Fp12 f, acc;
#if defined(BN_TRIO_FORMULAS)
  miller_verify_rounds(f, h, pk, sig);               // the round-structured loop of the octet layout (bn254_trio.hip)
  {
    Fp12 f2;                                         // ... must give the very Miller value of the generic loop
    miller_loop<true, true>(f2, h, pk, sig);
    Fp12 d1, d2;
    final_exponentiation(d1, f, acc); final_exponentiation(d2, f2, acc);
    const Fp2* a[6] = {&d1.c0.c0, &d1.c0.c1, &d1.c0.c2, &d1.c1.c0, &d1.c1.c1, &d1.c1.c2};
    const Fp2* b[6] = {&d2.c0.c0, &d2.c0.c1, &d2.c0.c2, &d2.c1.c0, &d2.c1.c1, &d2.c1.c2};
    for (int k = 0; k < 6; ++k) if (!fp2_eq(*a[k], *b[k])) return 254;
    Fp12 f3, d3;                                     // the same loop as wave roles (k_miller_verify_quad), in barrier order
    miller_verify_quad_model(f3, h, pk, sig);
    final_exponentiation(d3, f3, acc);
    const Fp2* c[6] = {&d3.c0.c0, &d3.c0.c1, &d3.c0.c2, &d3.c1.c0, &d3.c1.c1, &d3.c1.c2};
    for (int k = 0; k < 6; ++k) if (!fp2_eq(*c[k], *b[k])) return 253;
    Fp12 f4, d4;                                     // ... and as eight wave roles (k_miller_verify_w8), phase by phase
    miller_verify_w8_model(f4, h, pk, sig);
    final_exponentiation(d4, f4, acc);
    const Fp2* e[6] = {&d4.c0.c0, &d4.c0.c1, &d4.c0.c2, &d4.c1.c0, &d4.c1.c1, &d4.c1.c2};
    for (int k = 0; k < 6; ++k) if (!fp2_eq(*e[k], *b[k])) return 252;
  }
#else
  miller_loop<true, true>(f, h, pk, sig);
#endif
  Fp12 g, m1 = f, m2 = f;
  final_exponentiation_check(g, f, acc);            // the chain as straight-line code (octet / one-lane kernels)
  final_exponentiation(f, f, acc);                   // the exact value must agree on "is one"
  fe_machine_check(m1);                              // what k_final_exp_pair runs for a status: the same chains as programs of the
  fe_machine_exact(m2);                              // accumulator machine -> the very same values, coefficient by coefficient
  {
    const Fp2* a[6] = {&g.c0.c0, &g.c0.c1, &g.c0.c2, &g.c1.c0, &g.c1.c1, &g.c1.c2};
    const Fp2* b[6] = {&m1.c0.c0, &m1.c0.c1, &m1.c0.c2, &m1.c1.c0, &m1.c1.c1, &m1.c1.c2};
    const Fp2* c[6] = {&f.c0.c0, &f.c0.c1, &f.c0.c2, &f.c1.c0, &f.c1.c1, &f.c1.c2};
    const Fp2* d[6] = {&m2.c0.c0, &m2.c0.c1, &m2.c0.c2, &m2.c1.c0, &m2.c1.c1, &m2.c1.c2};
    for (int k = 0; k < 6; ++k) if (!fp2_eq(*a[k], *b[k]) || !fp2_eq(*c[k], *d[k])) return 251;
  }
  const bool one_check = fp12_is_one(g), one_exact = fp12_is_one(f);
  if (one_check != one_exact) return 255;
  return one_check ? 0 : 9;
}
// The NONET schedule of the final exponentiation (bn254_nonet.hip: one verify on nine lane pairs; bn254_nonet.h: nn_machine_model — the very
// phase functions of the kernel on a host box, the nine pairs of every exchange step one after the other): program C_FE_CHECK on the
// Miller value of the tuple.  0 / 9 = its verdict, which must be fe_machine_check's; 248 = some coefficient differs in VALUE from the
// pair layout's accumulator machine.  *words_equal (may be null): 1 if all 6 x 2 x 9 limbs are the same words as well (they are wherever
// the "strictest site mode of the three" rule changes no site of the flow).  Under -DBN_TRACK_BOUNDS one call is the bound proof of the flow.
int hp_nonet_check(const uint8_t* h64, const uint8_t* sig64, const uint8_t* pk128, int* words_equal) {
  G1Affine h, sig;
  G2Affine pk;
  load_g1(h, h64); load_g1(sig, sig64); load_g2(pk, pk128);
  Fp12 f;
  miller_loop<true, true>(f, h, pk, sig);
  Fp12 a = f, b = f;
  fe_machine_check(a);
  nn_machine_model(b, C_FE_CHECK);
  const Fp2* x[6] = {&a.c0.c0, &a.c0.c1, &a.c0.c2, &a.c1.c0, &a.c1.c1, &a.c1.c2};
  const Fp2* y[6] = {&b.c0.c0, &b.c0.c1, &b.c0.c2, &b.c1.c0, &b.c1.c1, &b.c1.c2};
  int same = 1;
  for (int k = 0; k < 6; ++k) {
    if (!fp2_eq(*x[k], *y[k])) return 248;
    for (int r = 0; r < 2; ++r) for (int i = 0; i < BN_LIMBS; ++i) if (x[k]->c[r].v[i] != y[k]->c[r].v[i]) same = 0;
  }
  if (words_equal) *words_equal = same;
  const bool one_a = fp12_is_one(a), one_b = fp12_is_one(b);
  if (one_a != one_b) return 247;
  return one_b ? 0 : 9;
}
// The LANE MACHINE schedule of the Miller loop (bn254_lmiller.hip: one verify on nine lane pairs in each of four waves; bn254_lmachine.h:
// lm_miller_model — the kernel's stage functions and level tables on a host box, tick by tick): 0 / 9 = the verdict of its Miller value
// under the final exponentiation, which must be the generic loop's; 246 = the two Gt values differ.  Under -DBN_TRACK_BOUNDS one call is
// the bound proof of the schedule (every product column, int32 limb and value bound of every level, and of the general Fq12 product on
// Miller values).
int hp_lm_verify(const uint8_t* h64, const uint8_t* sig64, const uint8_t* pk128) {
  G1Affine h, sig;
  G2Affine pk;
  load_g1(h, h64); load_g1(sig, sig64); load_g2(pk, pk128);
  Fp12 f, g, acc;
  miller_loop<true, true>(f, h, pk, sig);
  lm_miller_model(g, h, pk, sig);
  Fp12 a = g, b = g;
  fe_machine_check(a);                                // the check chain on the lane machine's value (within the chain's bound contract?) ...
  nn_machine_model(b, C_FE_CHECK);                    // ... and in the nonet schedule, which is what follows the kernel on the device
  if (fp12_is_one(a) != fp12_is_one(b)) return 244;
  final_exponentiation(f, f, acc);
  final_exponentiation(g, g, acc);
  const Fp2* x[6] = {&f.c0.c0, &f.c0.c1, &f.c0.c2, &f.c1.c0, &f.c1.c1, &f.c1.c2};
  const Fp2* y[6] = {&g.c0.c0, &g.c0.c1, &g.c0.c2, &g.c1.c0, &g.c1.c1, &g.c1.c2};
  for (int k = 0; k < 6; ++k) if (!fp2_eq(*x[k], *y[k])) return 246;
  if (fp12_is_one(a) != fp12_is_one(g)) return 245;
  return fp12_is_one(g) ? 0 : 9;
}
// ... and its KEYED form (k_miller_verify_lmk: the key's lines tabulated, no twist-point wave; lm_miller_keyed_model): 0 / 9, 243 = its Gt
// value differs from the keyed pair loop's, 249 = the table could not be built (the identity key: the loop then skips pair A anyway)
int hp_lm_verify_keyed(const uint8_t* h64, const uint8_t* sig64, const uint8_t* pk128) {
  G1Affine h, sig;
  G2Affine pk;
  load_g1(h, h64); load_g1(sig, sig64); load_g2(pk, pk128);
  static int32_t tab[BN_N_FIXED_LINES][2][2][BN_LIMBS];
  const bool ok = g2_line_table(pk, [&](int idx, const KeyLine& kl) {
    const Fp2* c[2] = {&kl.c0, &kl.c1};
    for (int e = 0; e < 2; ++e)
      for (int r = 0; r < 2; ++r) { Fp x = fp_canon(c[e]->c[r]); for (int k = 0; k < BN_LIMBS; ++k) tab[idx][e][r][k] = x.v[k]; }
  });
  if (!ok) return 249;
  Fp12 f, g;
  miller_loop_keyed(f, h, pk.inf, tab, sig);
  lm_miller_keyed_model(g, h, pk.inf, tab, sig);
  Fp12 a = g, b = g;
  fe_machine_check(a);
  nn_machine_model(b, C_FE_CHECK);
  if (fp12_is_one(a) != fp12_is_one(b)) return 244;
  fe_machine_exact(f);
  fe_machine_exact(g);
  const Fp2* x[6] = {&f.c0.c0, &f.c0.c1, &f.c0.c2, &f.c1.c0, &f.c1.c1, &f.c1.c2};
  const Fp2* y[6] = {&g.c0.c0, &g.c0.c1, &g.c0.c2, &g.c1.c0, &g.c1.c1, &g.c1.c2};
  for (int k = 0; k < 6; ++k) if (!fp2_eq(*x[k], *y[k])) return 243;
  if (fp12_is_one(a) != fp12_is_one(g)) return 245;
  return fp12_is_one(g) ? 0 : 9;
}
#if defined(BN_TRACK_BOUNDS)
// What the interval tracker knows about the Miller value a verify hands to its final exponentiation: per coefficient (12, Gt order)
// {limb lo, limb hi, |top limb| max, value/q lo, value/q hi} — the contract an adversarial-limb test may fill to the brim
// (tests/test_bounds.py::test_fe_input_contract_matches_tracker, tests/golden/adversarial_fe_vectors.json).
void hp_miller_output_bounds(const uint8_t* h64, const uint8_t* sig64, const uint8_t* pk128, double* out60) {
  G1Affine h, sig;
  G2Affine pk;
  load_g1(h, h64); load_g1(sig, sig64); load_g2(pk, pk128);
  Fp12 f;
  miller_loop<true, true>(f, h, pk, sig);
  const Fp2* c[6] = {&f.c0.c0, &f.c0.c1, &f.c0.c2, &f.c1.c0, &f.c1.c1, &f.c1.c2};
  for (int k = 0; k < 6; ++k)
    for (int r = 0; r < 2; ++r) {
      const FpBounds& b = c[k]->c[r].bd;
      double* o = out60 + (2 * k + r) * 5;
      o[0] = b.lo; o[1] = b.hi; o[2] = b.top; o[3] = b.vlo; o[4] = b.vhi;
    }
}
#endif
// The final exponentiation on caller-supplied LIMB vectors (12 coefficients x 9 limbs, Montgomery form, Gt order) through the pair layout's
// accumulator machine (exact != 0: program C_FE_EXACT, canonical Gt bytes out; else C_FE_CHECK) and, for the check program, through the
// nonet schedule as well: returns 0 / 9 (is one / is not), 248 if the two schedules differ.  The host twin of bn254_debug_final_exp_limbs.
int hp_final_exp_limbs(const int32_t* limbs108, int exact, uint8_t* gt384) {
  Fp12 f;
  Fp2* c[6] = {&f.c0.c0, &f.c0.c1, &f.c0.c2, &f.c1.c0, &f.c1.c1, &f.c1.c2};
  for (int k = 0; k < 6; ++k)
    for (int r = 0; r < 2; ++r) {
      Fp x = fp_zero();
      for (int i = 0; i < BN_LIMBS; ++i) x.v[i] = limbs108[(2 * k + r) * BN_LIMBS + i];
#if defined(BN_TRACK_BOUNDS)
      // the contract of a Miller output (hp_miller_output_bounds): tight limbs, |value| <= 0.5215 q
      bn_set_tight(x, -0.5215, 0.5215);
#endif
      c[k]->c[r] = x;
    }
  if (exact) {
    fe_machine_exact(f);
    if (gt384) for (int k = 0; k < 6; ++k) { fp_to_be32(gt384 + 64 * k, c[k]->c[0]); fp_to_be32(gt384 + 64 * k + 32, c[k]->c[1]); }
    return fp12_is_one(f) ? 0 : 9;
  }
  Fp12 g = f;
  fe_machine_check(f);
  nn_machine_model(g, C_FE_CHECK);
  const Fp2* d[6] = {&g.c0.c0, &g.c0.c1, &g.c0.c2, &g.c1.c0, &g.c1.c1, &g.c1.c2};
  for (int k = 0; k < 6; ++k) if (!fp2_eq(*c[k], *d[k])) return 248;
  if (gt384) for (int k = 0; k < 6; ++k) { fp_to_be32(gt384 + 64 * k, c[k]->c[0]); fp_to_be32(gt384 + 64 * k + 32, c[k]->c[1]); }
  return fp12_is_one(f) ? 0 : 9;
}
// The keyed verify (k_register_keys + k_miller_verify_keyed_pair): the key's 87 lines tabulated once (c2 = 1 form, canonical
// limbs as the kernel stores them), then the table-driven loop; returns 0 / 9 as hp_verify_decoded, 250 if its Gt value
// differs from the generic loop's, 249 if the table could not be built.  tab_out (may be null): 87 x 2 x 2 x 9 words.
int hp_verify_keyed_decoded(const uint8_t* h64, const uint8_t* sig64, const uint8_t* pk128, int32_t* tab_out) {
  G1Affine h, sig;
  G2Affine pk;
  load_g1(h, h64); load_g1(sig, sig64); load_g2(pk, pk128);
  static int32_t tab[BN_N_FIXED_LINES][2][2][BN_LIMBS];
  const bool ok = g2_line_table(pk, [&](int idx, const KeyLine& kl) {
    const Fp2* c[2] = {&kl.c0, &kl.c1};
    for (int e = 0; e < 2; ++e)
      for (int r = 0; r < 2; ++r) { Fp x = fp_canon(c[e]->c[r]); for (int k = 0; k < BN_LIMBS; ++k) tab[idx][e][r][k] = x.v[k]; }
  });
  if (!ok) return 249;
  if (tab_out) memcpy(tab_out, tab, sizeof tab);
  Fp12 f, g;
  miller_loop_keyed(f, h, pk.inf, tab, sig);
  miller_loop<true, true>(g, h, pk, sig);
  fe_machine_exact(f);
  fe_machine_exact(g);
  const Fp2* a[6] = {&f.c0.c0, &f.c0.c1, &f.c0.c2, &f.c1.c0, &f.c1.c1, &f.c1.c2};
  const Fp2* b[6] = {&g.c0.c0, &g.c0.c1, &g.c0.c2, &g.c1.c0, &g.c1.c1, &g.c1.c2};
  for (int k = 0; k < 6; ++k) if (!fp2_eq(*a[k], *b[k])) return 250;
  return fp12_is_one(f) ? 0 : 9;
}
// Products per LANE of the pair kernels for one verify / one pairing (both roles run here in sequence, so totals / 2):
// out = {miller_verify dual, single, final_exp dual, single, miller_var dual, single}.  "single" includes squares.
// bench.py prices the multiplier-class instructions of a kernel with these (profiles/lane_product_counts.json).
void hp_lane_counts(const uint8_t* h64, const uint8_t* sig64, const uint8_t* pk128, unsigned long long* out6) {
  G1Affine h, sig;
  G2Affine pk;
  load_g1(h, h64); load_g1(sig, sig64); load_g2(pk, pk128);
  Fp12 f, g, acc;
  unsigned long long m0 = bn_fp_mul_counter, d0 = bn_fp_dual_counter;
  miller_loop<true, true>(f, h, pk, sig);
  out6[0] = (bn_fp_dual_counter - d0) / 2; out6[1] = ((bn_fp_mul_counter - m0) - (bn_fp_dual_counter - d0)) / 2;
  m0 = bn_fp_mul_counter; d0 = bn_fp_dual_counter;
  g = f;
  fe_machine_check(g);
  bool one = fp12_is_one(g);
  (void)one;
  out6[2] = (bn_fp_dual_counter - d0) / 2; out6[3] = ((bn_fp_mul_counter - m0) - (bn_fp_dual_counter - d0)) / 2;
  m0 = bn_fp_mul_counter; d0 = bn_fp_dual_counter;
  miller_loop<true, false>(f, h, pk, h);
  out6[4] = (bn_fp_dual_counter - d0) / 2; out6[5] = ((bn_fp_mul_counter - m0) - (bn_fp_dual_counter - d0)) / 2;
}
// the same for the keyed Miller loop: out2 = {dual, single} per lane
void hp_lane_counts_keyed(const uint8_t* h64, const uint8_t* sig64, const uint8_t* pk128, unsigned long long* out2) {
  static int32_t tab[BN_N_FIXED_LINES][2][2][BN_LIMBS];
  if (hp_verify_keyed_decoded(h64, sig64, pk128, &tab[0][0][0][0]) > 9) { out2[0] = out2[1] = 0; return; }
  G1Affine h, sig;
  load_g1(h, h64); load_g1(sig, sig64);
  Fp12 f;
  unsigned long long m0 = bn_fp_mul_counter, d0 = bn_fp_dual_counter;
  miller_loop_keyed(f, h, false, tab, sig);
  out2[0] = (bn_fp_dual_counter - d0) / 2; out2[1] = ((bn_fp_mul_counter - m0) - (bn_fp_dual_counter - d0)) / 2;
}
// Fq products (both lanes of the pair together, i.e. per tuple) of the group operations of k_aggregate_pair:
// out = {G1 mixed addition, G2 mixed addition, G1 to affine, G2 to affine, G1 full addition}
void hp_group_op_counts(unsigned long long* out5) {
  G1Affine p1; p1.x = fp_load_const(C_G1_GEN[0]); p1.y = fp_load_const(C_G1_GEN[1]); p1.inf = false;
  G2Affine p2; p2.x = fp2_load_const(C_G2_GEN[0]); p2.y = fp2_load_const(C_G2_GEN[1]); p2.inf = false;
  G1Jac a1, b1; G2Jac a2;
  jac_set_identity(a1); jac_set_identity(a2);
  jac_accumulate(a1, p1); jac_accumulate(a1, p1);        // 2 P (second call: the doubling case)
  jac_accumulate(a2, p2); jac_accumulate(a2, p2);
  b1 = a1;
  unsigned long long m0 = bn_fp_mul_counter;
  jac_accumulate(a1, p1); out5[0] = bn_fp_mul_counter - m0; m0 = bn_fp_mul_counter;      // 3 P: the common case
  jac_accumulate(a2, p2); out5[1] = bn_fp_mul_counter - m0; m0 = bn_fp_mul_counter;
  G1Affine r1; G2Affine r2;
  jac_to_affine(r1, a1); out5[2] = bn_fp_mul_counter - m0; m0 = bn_fp_mul_counter;
  jac_to_affine(r2, a2); out5[3] = bn_fp_mul_counter - m0; m0 = bn_fp_mul_counter;
  jac_add(a1, a1, b1); out5[4] = bn_fp_mul_counter - m0;
  (void)r1; (void)r2;
}
// canonical Gt of one pairing through the pair layout
void hp_pairing(const uint8_t* g1, const uint8_t* g2, uint8_t* gt384) {
  G1Affine p;
  G2Affine q;
  load_g1(p, g1); load_g2(q, g2);
  Fp12 f, acc;
  miller_loop<true, false>(f, p, q, p);
  (void)acc;
  fe_machine_exact(f);
  const Fp2* c[6] = {&f.c0.c0, &f.c0.c1, &f.c0.c2, &f.c1.c0, &f.c1.c1, &f.c1.c2};
  for (int k = 0; k < 6; ++k) { fp_to_be32(gt384 + 64 * k, c[k]->c[0]); fp_to_be32(gt384 + 64 * k + 32, c[k]->c[1]); }
}
// two variable pairs sharing f (k_miller_rand2_pair), product of two such values (the LDS tree), final exponentiation
// the G2 subgroup test with its ladder in the lane machine's level tables (k_g2_subgroup_lm; bn254_lmachine.h: lm_g2_subgroup_model) beside
// the lane-pair form (g2_in_subgroup): returns both verdicts as bits 0 (machine) and 1 (lane pairs); the input must be on the twist
int hp_g2_subgroup_both(const uint8_t* pk128) {
  G2Affine q;
  load_g2(q, pk128);
  return (lm_g2_subgroup_model(q) ? 1 : 0) | (g2_in_subgroup(q) ? 2 : 0);
}
// ... and as the small-batch kernels compute it (bn254_batch_pairing* for batches that cannot fill the chip): the lane machine's schedule with
// the fixed pair skipped (lm_miller_model, pb = identity), then program C_FE_EXACT in the nonet schedule; under -DBN_TRACK_BOUNDS the bound
// proof of that flow.  The caller compares the bytes with hp_pairing's.
void hp_pairing_small_batch(const uint8_t* g1, const uint8_t* g2, uint8_t* gt384) {
  G1Affine p, none;
  G2Affine q;
  load_g1(p, g1); load_g2(q, g2);
  none = p; none.inf = true;
  Fp12 f;
  lm_miller_model(f, p, q, none);
  nn_machine_model(f, C_FE_EXACT);
  const Fp2* c[6] = {&f.c0.c0, &f.c0.c1, &f.c0.c2, &f.c1.c0, &f.c1.c1, &f.c1.c2};
  for (int k = 0; k < 6; ++k) { fp_to_be32(gt384 + 64 * k, c[k]->c[0]); fp_to_be32(gt384 + 64 * k + 32, c[k]->c[1]); }
}
void hp_pairing_product4(const uint8_t* g1x4, const uint8_t* g2x4, uint8_t* gt384) {
  G1Affine p[4];
  G2Affine q[4];
  for (int j = 0; j < 4; ++j) { load_g1(p[j], g1x4 + 64 * j); load_g2(q[j], g2x4 + 128 * j); }
  Fp12 f, g, acc;
  miller_loop_2var(f, p[0], q[0], p[1], q[1]);
  miller_loop_2var(g, p[2], q[2], p[3], q[3]);
  fp12_mul(f, f, g);
  final_exponentiation(f, f, acc);
  const Fp2* c[6] = {&f.c0.c0, &f.c0.c1, &f.c0.c2, &f.c1.c0, &f.c1.c1, &f.c1.c2};
  for (int k = 0; k < 6; ++k) { fp_to_be32(gt384 + 64 * k, c[k]->c[0]); fp_to_be32(gt384 + 64 * k + 32, c[k]->c[1]); }
}
// running G2 sum with the common-case addition (k_aggregate_pair) and the subgroup test (k_decode_g2_pair)
int hp_g2_sum_and_subgroup(const uint8_t* pts, uint64_t k, uint8_t* out128) {
  G2Jac acc;
  jac_set_identity(acc);
  int in_all = 1;
  for (uint64_t j = 0; j < k; ++j) {
    G2Affine q;
    load_g2(q, pts + 128 * j);
    if (!g2_in_subgroup(q)) in_all = 0;
    jac_accumulate(acc, q);
  }
  G2Affine r;
  jac_to_affine(r, acc);
  if (r.inf) memset(out128, 0, 128);
  else { fp_to_be32(out128, r.x.c[0]); fp_to_be32(out128 + 32, r.x.c[1]); fp_to_be32(out128 + 64, r.y.c[0]); fp_to_be32(out128 + 96, r.y.c[1]); }
  return in_all;
}
// G2::from_compressed through the pair layout (Fq2 square root, sign choice, subgroup test): status + 128 bytes
int hp_g2_decompress(const uint8_t* c65, uint8_t* out128) {
  G2Affine q;
  uint8_t st = decompress_g2(q, c65);
  if (st == ST_OK && !g2_in_subgroup(q)) st = ST_NOT_MEMBER;
  memset(out128, 0, 128);
  if (st == ST_OK) { fp_to_be32(out128, q.x.c[0]); fp_to_be32(out128 + 32, q.x.c[1]); fp_to_be32(out128 + 64, q.y.c[0]); fp_to_be32(out128 + 96, q.y.c[1]); }
  return st;
}
}  // extern "C"
