// The LANE MACHINE: the Miller loop of ECDSA::verify (/root/reference/src/ecdsa.rs:49-64) for the SMALLEST batches, one verify on nine lane
// pairs in each of FOUR waves (bn254_lmiller.hip) — the layout that takes the depth out of the loop instead of the width.
//
// In the wave-role kernels (bn254_quad.hip) a lane pair still runs three Fq2 products in a row per phase and six phases per doubling step;
// a lone wave issues a multiplier-class instruction only every ~10 cycles, so for one verify the loop's time is products IN SEQUENCE.
// Here every product of a dependency level runs in its own lane pair at once:
//   wave T   the twist point, two steps ahead of the accumulator.  Formulas rearranged for DEPTH: the point carries w = 3b' z beside
//            (x, y, z), so a doubling is TWO product levels (y^2, z w, y z, x y, x^2, y w | X3, b h, b h3, (b + 3e)^2, e^2) instead of
//            three, and a mixed addition THREE (theta, mu | their squares and six cross products | nine products that give X3, Y3, Z3, W3
//            without waiting for h = e + f - 2g: X3 = mu^4 + (mu z) theta^2 - 2 (mu x) mu^2, ...) instead of four;
//   wave L   the step's line at H(m) times the table line of the fixed pair e(sig, -G2) at sig, one step ahead: two levels (evaluation,
//            product), result a full Fq12 (five coefficients, the sixth zero) in a double-buffered slot;
//   waves F0 / F1  the accumulator f: f^2 and f * L as the nonet layout's Karatsuba product (bn254_nonet.h: nn_mul_*), its two rounds of
//            nine products in the two waves at once, the coefficient levels replicated in both.
// A level is DATA: per lane pair one product entry (operands = sums of two slots, output slot) and one linear entry (out = weak_reduce of
// up to four slots with small factors, optionally xi on one term, optionally replaced by another slot when the verify's pair A / pair B
// is skipped) — `LmEntry`; the slots are a register file per verify in LDS.  One instruction stream per wave, no lane-dependent control flow.
// The waves meet at workgroup barriers ("ticks"): two per step, three while wave T is in an addition; hand-overs are double-buffered by step
// parity (slots from LS_REL0 on are relocated by the parity of the step they belong to).
//
// Every linear output is weakly reduced (fp_lin4_reduce), so every slot is tight and within +-0.52 q, every product operand a sum of at
// most two such values: the bound proof is the interval tracker's run over the host emulation below (lm_miller_model: the same stage
// functions on a host box, pairs and waves one after the other per tick; tests/test_pair_layout.py::test_lane_machine_*).  The Miller
// VALUE differs from the other layouts' by factors in Fq2 (w = 3b' z changes no coordinate, but f * L here is the general product, and the
// skip selects are applied to the same places), which the final exponentiation removes: the tests compare after it.
// Include after bn254_pairing.h and bn254_nonet.h (pair layout: BN_SPLIT_FP2).
#pragma once

namespace bn254 {

enum LmSlot {
  LS_ZERO = 0, LS_ONE, LS_DUMMY,
  LS_PAX, LS_PAY, LS_PBX, LS_PBY, LS_B3,
  LS_PKX, LS_PKY, LS_NPKY, LS_CPKX, LS_CPKY, LS_FX1, LS_FY1, LS_FX2, LS_Q1X, LS_Q1Y, LS_Q2X,
  // wave T: the point (x, y, z, w = 3b' z), the point it adds in the step in flight, temporaries
  LS_TX, LS_TY, LS_TZ, LS_TW, LS_TQX, LS_TQY,
  LS_TB, LS_TE, LS_TXY, LS_TX2, LS_TYW, LS_TYZ, LS_TH3, LS_TBF, LS_TBMF, LS_TXY2, LS_TOZ, LS_TOW, LS_TOY2, LS_TE2,
  LS_T1, LS_T2, LS_TC, LS_TD, LS_TMX, LS_TMZ, LS_TMY, LS_TTM, LS_TTHZ, LS_TMW, LS_TDD, LS_TPA, LS_TPB, LS_TPC, LS_TPD, LS_TPE, LS_TPF,
  // wave L: the point of the addition step it evaluates, the table line's constants, temporaries
  LS_LQX, LS_LQY, LS_MC0, LS_MC1, LS_KC0, LS_KC1, LS_XI,
  LS_LL0, LS_LL1, LS_LM0, LS_LM1, LS_LCA, LS_LCB, LS_LL0S, LS_LL1S, LS_LL2S, LS_LW3, LS_LW4, LS_LV0, LS_LV1, LS_LX01,
  // waves F: accumulator (six coefficients), products (two buffers of 18, by parity of the product), Fq6 coefficients (9)
  LS_ACC, LS_XP0 = LS_ACC + 6, LS_XP1 = LS_XP0 + 18, LS_X1 = LS_XP1 + 18,
  // by parity of the step: what wave T hands wave L (doubling: h, -3 x^2, l2 = b - e; addition: theta, mu; keyed form: the two scaled
  // table lines l0, l1, m0, m1), the line product (6)
  LS_REL0 = LS_X1 + 9,
  LS_HOA = LS_REL0, LS_HOB, LS_HOC, LS_HOD, LS_LP,
  LS_REL_N = 4 + 6,
  LS_COUNT = LS_REL0 + 2 * LS_REL_N
};
static_assert(LS_COUNT <= 255, "slot ids are bytes");

// ---- a level's entry for one lane pair: five packed words
//   w[0]  product: a1 | a2 << 8 | b1 << 16 | b2 << 24        operands (S[a1] + S[a2]) * (S[b1] + S[b2])
//   w[1]  product: out
//   w[2]  linear:  x0 | x1 << 8 | x2 << 16 | x3 << 24
//   w[3]  linear:  k0 .. k3, signed bytes
//   w[4]  linear:  out | flags << 8 | alt << 16 | za << 24 | zb << 28
//                  flags: 1 = xi on term 2; 2 / 4 = S[alt] instead when pair A / pair B is skipped; za / zb: bit j = term j dropped when pair
//                  A / pair B is skipped
struct LmEntry { uint32_t w[5]; };
enum { LM_XI = 1, LM_SKIP_A = 2, LM_SKIP_B = 4 };
constexpr uint32_t lm_b4(int a, int b, int c, int d) { return (uint32_t)(a & 255) | (uint32_t)(b & 255) << 8 | (uint32_t)(c & 255) << 16 | (uint32_t)(d & 255) << 24; }
struct LmP { int a1, a2, b1, b2, out; };
struct LmL { int out, x0, k0, x1, k1, x2, k2, x3, k3, flags, alt, za, zb; };
constexpr LmP LM_NOP_P = {LS_ZERO, LS_ZERO, LS_ZERO, LS_ZERO, LS_DUMMY};
constexpr LmL LM_NOP_L = {LS_DUMMY, LS_ZERO, 0, LS_ZERO, 0, LS_ZERO, 0, LS_ZERO, 0, 0, LS_ZERO, 0, 0};
constexpr LmEntry lm_entry(const LmP& p, const LmL& l) {
  return LmEntry{{lm_b4(p.a1, p.a2, p.b1, p.b2), (uint32_t)p.out, lm_b4(l.x0, l.x1, l.x2, l.x3), lm_b4(l.k0, l.k1, l.k2, l.k3),
                  (uint32_t)l.out | (uint32_t)l.flags << 8 | (uint32_t)l.alt << 16 | (uint32_t)l.za << 24 | (uint32_t)l.zb << 28}};
}
constexpr LmP lm_mul(int out, int a, int b) { return LmP{a, LS_ZERO, b, LS_ZERO, out}; }
constexpr LmL lm_lin(int out, int x0, int k0, int x1 = LS_ZERO, int k1 = 0, int x2 = LS_ZERO, int k2 = 0, int x3 = LS_ZERO, int k3 = 0) {
  return LmL{out, x0, k0, x1, k1, x2, k2, x3, k3, 0, LS_ZERO, 0, 0};
}
constexpr LmL lm_lin_skip(int flag, int alt, const LmL& l) { return LmL{l.out, l.x0, l.k0, l.x1, l.k1, l.x2, l.k2, l.x3, l.k3, l.flags | flag, alt, l.za, l.zb}; }
constexpr LmL lm_lin_drop(int za, int zb, const LmL& l) { return LmL{l.out, l.x0, l.k0, l.x1, l.k1, l.x2, l.k2, l.x3, l.k3, l.flags, l.alt, za, zb}; }
#define LM_E(P, L) lm_entry(P, L)

#if defined(__HIPCC__)
#define LM_TABLE __device__ __constant__ const LmEntry
#else
#define LM_TABLE static const LmEntry
#endif

// ---- the programs: [level][lane pair]
// wave T, before the loop: pi(Q), pi^2(Q).x, -Q.y; T = (Q.x, Q.y, 1, 3b')
LM_TABLE LM_T_INIT[1][9] = {{
    LM_E(lm_mul(LS_Q1X, LS_CPKX, LS_FX1), lm_lin(LS_NPKY, LS_PKY, -1)),
    LM_E(lm_mul(LS_Q1Y, LS_CPKY, LS_FY1), lm_lin(LS_TX, LS_PKX, 1)),
    LM_E(lm_mul(LS_Q2X, LS_PKX, LS_FX2), lm_lin(LS_TY, LS_PKY, 1)),
    LM_E(LM_NOP_P, lm_lin(LS_TZ, LS_ONE, 1)),
    LM_E(LM_NOP_P, lm_lin(LS_TW, LS_B3, 1)),
    LM_E(LM_NOP_P, LM_NOP_L), LM_E(LM_NOP_P, LM_NOP_L), LM_E(LM_NOP_P, LM_NOP_L), LM_E(LM_NOP_P, LM_NOP_L)}};
// wave T, doubling (bn254_pairing.h: quad_dbl_line / quad_dbl_update with e = z w, h3 = 2 y w):
//   b = y^2, e = 3b' z^2 = z w, h = 2 y z, l2 = b - e, X3 = 2 x y (b - 3e), Y3 = (b + 3e)^2 - 12 e^2, Z3 = 4 b h, W3 = 3b' Z3 = 4 b h3
LM_TABLE LM_T_DBL[2][9] = {
    {LM_E(lm_mul(LS_TB, LS_TY, LS_TY), lm_lin(LS_HOA, LS_TYZ, 2)),
     LM_E(lm_mul(LS_TE, LS_TZ, LS_TW), lm_lin(LS_TH3, LS_TYW, 2)),
     LM_E(lm_mul(LS_TYZ, LS_TY, LS_TZ), lm_lin(LS_HOC, LS_TB, 1, LS_TE, -1)),
     LM_E(lm_mul(LS_TXY, LS_TX, LS_TY), lm_lin(LS_TBF, LS_TB, 1, LS_TE, 3)),
     LM_E(lm_mul(LS_TX2, LS_TX, LS_TX), lm_lin(LS_TBMF, LS_TB, 1, LS_TE, -3)),
     LM_E(lm_mul(LS_TYW, LS_TY, LS_TW), lm_lin(LS_TXY2, LS_TXY, 2)),
     LM_E(LM_NOP_P, lm_lin(LS_HOB, LS_TX2, -3)),
     LM_E(LM_NOP_P, LM_NOP_L), LM_E(LM_NOP_P, LM_NOP_L)},
    {LM_E(lm_mul(LS_TX, LS_TXY2, LS_TBMF), lm_lin(LS_TY, LS_TOY2, 1, LS_TE2, -12)),
     LM_E(lm_mul(LS_TOZ, LS_TB, LS_HOA), lm_lin(LS_TZ, LS_TOZ, 4)),
     LM_E(lm_mul(LS_TOW, LS_TB, LS_TH3), lm_lin(LS_TW, LS_TOW, 4)),
     LM_E(lm_mul(LS_TOY2, LS_TBF, LS_TBF), LM_NOP_L),
     LM_E(lm_mul(LS_TE2, LS_TE, LS_TE), LM_NOP_L),
     LM_E(LM_NOP_P, LM_NOP_L), LM_E(LM_NOP_P, LM_NOP_L), LM_E(LM_NOP_P, LM_NOP_L), LM_E(LM_NOP_P, LM_NOP_L)}};
// wave T, addition of (qx, qy) = slots TQX, TQY (quad_add_line / quad_add_update expanded so that nothing waits for h = e + f - 2g):
//   theta = y - qy z, mu = x - qx z; c = theta^2, d = mu^2;
//   X3 = mu h = d^2 + (mu z) c - 2 (mu x) d;  Y3 = theta (g - h) - e y = 3 (theta mu)(mu x) - (theta mu) d - (theta z) c - (mu y) d;
//   Z3 = z e = (mu z) d;  W3 = w e = (mu w) d
LM_TABLE LM_T_ADD[3][9] = {
    {LM_E(lm_mul(LS_T1, LS_TQY, LS_TZ), lm_lin(LS_HOA, LS_TY, 1, LS_T1, -1)),
     LM_E(lm_mul(LS_T2, LS_TQX, LS_TZ), lm_lin(LS_HOB, LS_TX, 1, LS_T2, -1)),
     LM_E(LM_NOP_P, LM_NOP_L), LM_E(LM_NOP_P, LM_NOP_L), LM_E(LM_NOP_P, LM_NOP_L), LM_E(LM_NOP_P, LM_NOP_L), LM_E(LM_NOP_P, LM_NOP_L),
     LM_E(LM_NOP_P, LM_NOP_L), LM_E(LM_NOP_P, LM_NOP_L)},
    {LM_E(lm_mul(LS_TC, LS_HOA, LS_HOA), LM_NOP_L),
     LM_E(lm_mul(LS_TD, LS_HOB, LS_HOB), LM_NOP_L),
     LM_E(lm_mul(LS_TMX, LS_HOB, LS_TX), LM_NOP_L),
     LM_E(lm_mul(LS_TMZ, LS_HOB, LS_TZ), LM_NOP_L),
     LM_E(lm_mul(LS_TMY, LS_HOB, LS_TY), LM_NOP_L),
     LM_E(lm_mul(LS_TTM, LS_HOA, LS_HOB), LM_NOP_L),
     LM_E(lm_mul(LS_TTHZ, LS_HOA, LS_TZ), LM_NOP_L),
     LM_E(lm_mul(LS_TMW, LS_HOB, LS_TW), LM_NOP_L),
     LM_E(LM_NOP_P, LM_NOP_L)},
    {LM_E(lm_mul(LS_TDD, LS_TD, LS_TD), lm_lin(LS_TX, LS_TDD, 1, LS_TPA, 1, LS_TPB, -2)),
     LM_E(lm_mul(LS_TPA, LS_TMZ, LS_TC), lm_lin(LS_TY, LS_TPC, 3, LS_TPD, -1, LS_TPE, -1, LS_TPF, -1)),
     LM_E(lm_mul(LS_TPB, LS_TMX, LS_TD), LM_NOP_L),
     LM_E(lm_mul(LS_TPC, LS_TTM, LS_TMX), LM_NOP_L),
     LM_E(lm_mul(LS_TPD, LS_TTM, LS_TD), LM_NOP_L),
     LM_E(lm_mul(LS_TPE, LS_TTHZ, LS_TC), LM_NOP_L),
     LM_E(lm_mul(LS_TPF, LS_TMY, LS_TD), LM_NOP_L),
     LM_E(lm_mul(LS_TZ, LS_TMZ, LS_TD), LM_NOP_L),
     LM_E(lm_mul(LS_TW, LS_TMW, LS_TD), LM_NOP_L)}};
// wave L: the line of a doubling step at pair A's G1 point (h y_A, -3 x^2 x_A, l2) and the table line's scalings (C0 y_B, C1 x_B) ...
#define LM_L_SKIP_A(alt, lin) lm_lin_skip(LM_SKIP_A, alt, lin)
#define LM_L_SKIP_B(alt, lin) lm_lin_skip(LM_SKIP_B, alt, lin)
LM_TABLE LM_L_DBL[1][9] = {{
    LM_E(lm_mul(LS_LL0, LS_HOA, LS_PAY), LM_L_SKIP_A(LS_ONE, lm_lin(LS_LL0S, LS_LL0, 1))),
    LM_E(lm_mul(LS_LL1, LS_HOB, LS_PAX), LM_L_SKIP_A(LS_ZERO, lm_lin(LS_LL1S, LS_LL1, 1))),
    LM_E(lm_mul(LS_LM0, LS_MC0, LS_PBY), LM_L_SKIP_A(LS_ZERO, lm_lin(LS_LL2S, LS_HOC, 1))),
    LM_E(lm_mul(LS_LM1, LS_MC1, LS_PBX), LM_NOP_L),
    LM_E(LM_NOP_P, LM_NOP_L), LM_E(LM_NOP_P, LM_NOP_L), LM_E(LM_NOP_P, LM_NOP_L), LM_E(LM_NOP_P, LM_NOP_L), LM_E(LM_NOP_P, LM_NOP_L)}};
// ... of an addition step: l0 = mu y_A, l1 = -theta x_A, l2 = theta x_Q - mu y_Q
LM_TABLE LM_L_ADD[1][9] = {{
    LM_E(lm_mul(LS_LL0, LS_HOB, LS_PAY), LM_L_SKIP_A(LS_ONE, lm_lin(LS_LL0S, LS_LL0, 1))),
    LM_E(lm_mul(LS_LL1, LS_HOA, LS_PAX), LM_L_SKIP_A(LS_ZERO, lm_lin(LS_LL1S, LS_LL1, -1))),
    LM_E(lm_mul(LS_LM0, LS_MC0, LS_PBY), LM_L_SKIP_A(LS_ZERO, lm_lin(LS_LL2S, LS_LCA, 1, LS_LCB, -1))),
    LM_E(lm_mul(LS_LM1, LS_MC1, LS_PBX), LM_NOP_L),
    LM_E(lm_mul(LS_LCA, LS_HOA, LS_LQX), LM_NOP_L),
    LM_E(lm_mul(LS_LCB, LS_HOB, LS_LQY), LM_NOP_L),
    LM_E(LM_NOP_P, LM_NOP_L), LM_E(LM_NOP_P, LM_NOP_L), LM_E(LM_NOP_P, LM_NOP_L)}};
// ... and (l0 + l1 w + l2 w^3)(m0 + m1 w + w^3) = b00 + b01 v + b02 v^2 + (b10 + b11 v) w (bn254_pairing.h: trio_line_product):
//   b00 = l0 m0 + xi l2, b01 = l1 m1, b02 = l1 + l2 m1, b10 = (l0 + l1)(m0 + m1) - l0 m0 - l1 m1, b11 = l0 + l2 m0; pair B skipped: the line itself
LM_TABLE LM_L_PROD[1][9] = {{
    LM_E(lm_mul(LS_LW3, LS_LL2S, LS_LM0), LM_L_SKIP_B(LS_LL0S, lm_lin_skip(LM_XI, LS_ZERO, lm_lin(LS_LP + 0, LS_LV0, 1, LS_ZERO, 0, LS_LL2S, 1)))),
    LM_E(lm_mul(LS_LW4, LS_LL2S, LS_LM1), LM_L_SKIP_B(LS_ZERO, lm_lin(LS_LP + 1, LS_LV1, 1))),
    LM_E(lm_mul(LS_LV0, LS_LL0S, LS_LM0), LM_L_SKIP_B(LS_ZERO, lm_lin(LS_LP + 2, LS_LL1S, 1, LS_LW4, 1))),
    LM_E(lm_mul(LS_LV1, LS_LL1S, LS_LM1), LM_L_SKIP_B(LS_LL1S, lm_lin(LS_LP + 3, LS_LX01, 1, LS_LV0, -1, LS_LV1, -1))),
    LM_E((LmP{LS_LL0S, LS_LL1S, LS_LM0, LS_LM1, LS_LX01}), LM_L_SKIP_B(LS_LL2S, lm_lin(LS_LP + 4, LS_LL0S, 1, LS_LW3, 1))),
    LM_E(LM_NOP_P, LM_NOP_L), LM_E(LM_NOP_P, LM_NOP_L), LM_E(LM_NOP_P, LM_NOP_L), LM_E(LM_NOP_P, LM_NOP_L)}};

// KEYED form (registered public keys: the lines of pair A tabulated in the c2 = 1 form like those of the fixed pair; no twist-point wave):
// wave LA, two steps ahead: both table lines scaled by their G1 points, a skipped pair's line replaced by 1 = (1, 0, c2 = 0) ...
LM_TABLE LM_K_EVAL[1][9] = {{
    LM_E(lm_mul(LS_LL0, LS_KC0, LS_PAY), LM_L_SKIP_A(LS_ONE, lm_lin(LS_HOA, LS_LL0, 1))),
    LM_E(lm_mul(LS_LL1, LS_KC1, LS_PAX), LM_L_SKIP_A(LS_ZERO, lm_lin(LS_HOB, LS_LL1, 1))),
    LM_E(lm_mul(LS_LM0, LS_MC0, LS_PBY), LM_L_SKIP_B(LS_ONE, lm_lin(LS_HOC, LS_LM0, 1))),
    LM_E(lm_mul(LS_LM1, LS_MC1, LS_PBX), LM_L_SKIP_B(LS_ZERO, lm_lin(LS_HOD, LS_LM1, 1))),
    LM_E(LM_NOP_P, LM_NOP_L), LM_E(LM_NOP_P, LM_NOP_L), LM_E(LM_NOP_P, LM_NOP_L), LM_E(LM_NOP_P, LM_NOP_L), LM_E(LM_NOP_P, LM_NOP_L)}};
// ... wave LB, one step ahead: (l0 + l1 w + ca w^3)(m0 + m1 w + cb w^3), ca / cb = 0 for a skipped pair (bn254_pairing.h: mul_by_two_table_lines):
//   b00 = l0 m0 + ca cb xi, b01 = l1 m1, b02 = cb l1 + ca m1, b10 = (l0 + l1)(m0 + m1) - l0 m0 - l1 m1, b11 = cb l0 + ca m0
LM_TABLE LM_K_PROD[1][9] = {{
    LM_E(lm_mul(LS_LV0, LS_HOA, LS_HOC), lm_lin_drop(2, 2, lm_lin(LS_LP + 0, LS_LV0, 1, LS_XI, 1))),
    LM_E(lm_mul(LS_LV1, LS_HOB, LS_HOD), lm_lin(LS_LP + 1, LS_LV1, 1)),
    LM_E((LmP{LS_HOA, LS_HOB, LS_HOC, LS_HOD, LS_LX01}), lm_lin_drop(2, 1, lm_lin(LS_LP + 2, LS_HOB, 1, LS_HOD, 1))),
    LM_E(LM_NOP_P, lm_lin(LS_LP + 3, LS_LX01, 1, LS_LV0, -1, LS_LV1, -1)),
    LM_E(LM_NOP_P, lm_lin_drop(2, 1, lm_lin(LS_LP + 4, LS_HOA, 1, LS_HOC, 1))),
    LM_E(LM_NOP_P, LM_NOP_L), LM_E(LM_NOP_P, LM_NOP_L), LM_E(LM_NOP_P, LM_NOP_L), LM_E(LM_NOP_P, LM_NOP_L)}};

// ---- the two stages of a level, for one lane pair.  `par`: parity of the step the wave is working on (relocates slots >= LS_REL0).
template <class Box> BN_DEV typename Box::Ref lm_ref(Box& bx, uint32_t id, unsigned par) { return bx.slot(id + (id >= (uint32_t)LS_REL0 ? par * (unsigned)LS_REL_N : 0u)); }
template <class Box> BN_DEV Fp2 lm_stage_product(const LmEntry& e, Box& bx, unsigned par) {
  const Fp2 a = fp2_add(bx.get(lm_ref(bx, e.w[0] & 255u, par)), bx.get(lm_ref(bx, (e.w[0] >> 8) & 255u, par)));
  const Fp2 b = fp2_add(bx.get(lm_ref(bx, (e.w[0] >> 16) & 255u, par)), bx.get(lm_ref(bx, e.w[0] >> 24, par)));
  return fp2_mul(a, b);
}
template <class Box> BN_DEV typename Box::Ref lm_product_out(const LmEntry& e, Box& bx, unsigned par) { return lm_ref(bx, e.w[1] & 255u, par); }
BN_DEV int32_t lm_k(uint32_t w, int j) { return (int32_t)(w << (24 - 8 * j)) >> 24; }
template <class Box> BN_DEV Fp2 lm_stage_linear(const LmEntry& e, Box& bx, unsigned par, bool skip_a, bool skip_b) {
  const Fp2 x0 = bx.get(lm_ref(bx, e.w[2] & 255u, par)), x1 = bx.get(lm_ref(bx, (e.w[2] >> 8) & 255u, par));
  const Fp2 x2 = bx.get(lm_ref(bx, (e.w[2] >> 16) & 255u, par)), x3 = bx.get(lm_ref(bx, e.w[2] >> 24, par));
  const uint32_t flags = (e.w[4] >> 8) & 255u;
  const Fp2 x2x = fp2_select_pos((flags & LM_XI) != 0, fp2_mul_xi(x2), x2);
  const uint32_t drop = (skip_a ? (e.w[4] >> 24) & 15u : 0u) | (skip_b ? e.w[4] >> 28 : 0u);      // terms that fall away with a skipped pair
  const Fp2 r = fp2_lin4_reduce(x0, (drop & 1u) ? 0 : lm_k(e.w[3], 0), x1, (drop & 2u) ? 0 : lm_k(e.w[3], 1), x2x, (drop & 4u) ? 0 : lm_k(e.w[3], 2),
                                x3, (drop & 8u) ? 0 : lm_k(e.w[3], 3));
  const bool replace = ((flags & LM_SKIP_A) != 0 && skip_a) || ((flags & LM_SKIP_B) != 0 && skip_b);
  return fp2_select(replace, bx.get(lm_ref(bx, (e.w[4] >> 16) & 255u, par)), r);
}
template <class Box> BN_DEV typename Box::Ref lm_linear_out(const LmEntry& e, Box& bx, unsigned par) { return lm_ref(bx, e.w[4] & 255u, par); }

// ---- what the waves do in a tick, as functions of the step they are working on (QuadSteps: bn254_pairing.h)
// ticks of the global step whose wave-T step has type `ty_t` (2 ahead of the accumulator's): three while T adds a point
BN_DEV int lm_ticks(int ty_t) { return (ty_t != 0 && ty_t != 4) ? 3 : 2; }
// the point an addition step of type ty adds: slots of its coordinates
BN_DEV int lm_q_x(int ty) { return ty == 2 ? LS_Q1X : ty == 3 ? LS_Q2X : LS_PKX; }
BN_DEV int lm_q_y(int ty) { return ty == 2 ? LS_Q1Y : ty == -1 ? LS_NPKY : LS_PKY; }

// the sequence of steps of the loop (bn254_pairing.h: miller_loop): 64 doublings, an addition of +-Q after those with a non-zero digit of
// 6u + 2, then + pi(Q), - pi^2(Q): type of step k (0 doubling, +-1 / 2 / 3 the additions), 4 behind the last (two of those)
template <class T> BN_DEV void lm_step_types(T* ty) {
  int k = 0;
  for (int d = 0; d < 64; ++d) {
    ty[k++] = 0;
    if (C_ATE_NAF[d] != 0) ty[k++] = (T)C_ATE_NAF[d];
  }
  ty[k++] = 2; ty[k++] = 3;
  ty[k] = ty[k + 1] = 4;
}

#if !defined(__HIPCC__)
// ---- host emulation: the register file as a plain array, the nine pairs of a stage one after the other, the waves of a tick one after the other
struct LmHostBox {
  typedef Fp2* Ref;
  typedef unsigned Rel;
  typedef Fp2* Base;
  Fp2 s[LS_COUNT];
  Ref slot(uint32_t id) { return &s[id]; }
  Ref coef(unsigned idx) { return idx < 6 ? &s[LS_ACC + idx] : &s[LS_ZERO]; }
  Ref xp(unsigned q) { return &s[LS_XP0 + q]; }
  Ref x1(unsigned q) { return q < 9 ? &s[LS_X1 + q] : &s[LS_ZERO]; }
  Ref zero() { return &s[LS_ZERO]; }
  static Rel rel(unsigned idx) { return idx; }
  static Ref at(Base b, Rel r) { return b + r; }
  Fp2 get(Ref r) const { return *r; }
  void put(Ref r, const Fp2& v) { *r = v; }
};
inline void lm_host_level(LmHostBox& bx, const LmEntry (&lvl)[9], unsigned par, bool skip_a, bool skip_b) {
  Fp2 t[9];
  for (int p = 0; p < 9; ++p) t[p] = lm_stage_product(lvl[p], bx, par);
  for (int p = 0; p < 9; ++p) bx.put(lm_product_out(lvl[p], bx, par), t[p]);
  for (int p = 0; p < 9; ++p) t[p] = lm_stage_linear(lvl[p], bx, par, skip_a, skip_b);
  for (int p = 0; p < 9; ++p) bx.put(lm_linear_out(lvl[p], bx, par), t[p]);
  bx.s[LS_DUMMY] = fp2_zero();
}
// acc <- acc * b (b = &slot of coefficient 0): the nonet layout's product, rounds 0 and 1 in waves F0 and F1
inline void lm_host_mul(LmHostBox& bx, NnLane<LmHostBox> (&L)[9], Fp2* b) {
  Fp2 t[9];
  for (unsigned r = 0; r < 2; ++r) {
    for (unsigned p = 0; p < 9; ++p) t[p] = nn_mul_product(L[p], bx, &bx.s[LS_ACC], b, r);
    for (unsigned p = 0; p < 9; ++p) bx.put(L[p].m_pub[r], t[p]);
  }
  for (unsigned p = 0; p < 9; ++p) t[p] = nn_mul_level1(L[p], bx);
  for (unsigned p = 0; p < 9; ++p) bx.put(L[p].l1_pub, t[p]);
  for (unsigned p = 0; p < 9; ++p) t[p] = nn_mul_level2(L[p], bx);
  for (unsigned p = 0; p < 6; ++p) bx.put(L[p].out_coef, t[p]);
}
// the Miller value of e(pa, qa) e(pb, -G2) in the lane machine's schedule
inline void lm_miller_model(Fp12& f, const G1Affine& pa, const G2Affine& qa, const G1Affine& pb) {
  static LmHostBox bx;
  static NnLane<LmHostBox> L[9];
  for (int i = 0; i < LS_COUNT; ++i) bx.s[i] = fp2_zero();
  for (unsigned p = 0; p < 9; ++p) nn_lane_roles(L[p], bx, p, true);
  const bool skip_a = pa.inf || qa.inf, skip_b = pb.inf;
  bx.s[LS_ONE] = fp2_one();
  bx.s[LS_PAX] = fp2_from_fp(pa.x); bx.s[LS_PAY] = fp2_from_fp(pa.y); bx.s[LS_PBX] = fp2_from_fp(pb.x); bx.s[LS_PBY] = fp2_from_fp(pb.y);
  bx.s[LS_B3] = fp2_load_const(C_TWIST_3B);
  bx.s[LS_PKX] = qa.x; bx.s[LS_PKY] = qa.y; bx.s[LS_CPKX] = fp2_conj(qa.x); bx.s[LS_CPKY] = fp2_conj(qa.y);
  bx.s[LS_FX1] = fp2_load_const(C_TW_FROB_X1); bx.s[LS_FY1] = fp2_load_const(C_TW_FROB_Y1); bx.s[LS_FX2] = fp2_load_const(C_TW_FROB_X2);
  bx.s[LS_ACC] = fp2_one();
  lm_host_level(bx, LM_T_INIT[0], 0, skip_a, skip_b);
  signed char ty[BN_N_FIXED_LINES + 2];                          // type of step k (0 doubling, +-1 / 2 / 3 the additions), 4 past the end
  lm_step_types(ty);
  // global step g: wave T works on step g + 2, wave L on step g + 1, waves F on step g
  for (int g = -2; g < BN_N_FIXED_LINES; ++g) {
    const int ty_t = ty[g + 2], ty_l = g + 1 >= 0 ? ty[g + 1] : 4, ty_f = g >= 0 ? ty[g] : 4;
    const unsigned par_t = (unsigned)(g + 2) & 1u, par_l = (unsigned)(g + 1) & 1u, par_f = (unsigned)g & 1u;
    const int ticks = lm_ticks(ty_t);
    for (int tick = 0; tick < ticks; ++tick) {
      // wave T
      if (ty_t == 0) { if (tick < 2) lm_host_level(bx, LM_T_DBL[tick], par_t, skip_a, skip_b); }
      else if (ty_t != 4) {
        if (tick == 0) { bx.s[LS_TQX] = bx.s[lm_q_x(ty_t)]; bx.s[LS_TQY] = bx.s[lm_q_y(ty_t)]; }
        lm_host_level(bx, LM_T_ADD[tick], par_t, skip_a, skip_b);
      }
      // wave L
      if (ty_l != 4) {
        if (tick == 0) {
          bx.s[LS_MC0] = fp2_load_const(C_NEG_G2_LINES[g + 1][0]); bx.s[LS_MC1] = fp2_load_const(C_NEG_G2_LINES[g + 1][1]);
          if (ty_l != 0) { bx.s[LS_LQX] = bx.s[lm_q_x(ty_l)]; bx.s[LS_LQY] = bx.s[lm_q_y(ty_l)]; }
          lm_host_level(bx, ty_l == 0 ? LM_L_DBL[0] : LM_L_ADD[0], par_l, skip_a, skip_b);
        } else if (tick == 1) lm_host_level(bx, LM_L_PROD[0], par_l, skip_a, skip_b);
      }
      // waves F
      if (ty_f != 4) {
        Fp2* lp = &bx.s[LS_LP + par_f * LS_REL_N];
        if (ty_f == 0) { if (tick == 0) lm_host_mul(bx, L, &bx.s[LS_ACC]); else if (tick == 1) lm_host_mul(bx, L, lp); }
        else if (tick == 0) lm_host_mul(bx, L, lp);
      }
    }
  }
  f.c0.c0 = bx.s[LS_ACC]; f.c0.c1 = bx.s[LS_ACC + 1]; f.c0.c2 = bx.s[LS_ACC + 2];
  f.c1.c0 = bx.s[LS_ACC + 3]; f.c1.c1 = bx.s[LS_ACC + 4]; f.c1.c2 = bx.s[LS_ACC + 5];
}
// the Miller value of e(pa, key) e(pb, -G2) in the KEYED schedule: `tab` = the key's 87 x (c0, c1) (bn254_pairing.h: miller_loop_keyed).
// Global step g: wave LA works on step g + 2, wave LB on step g + 1, waves F on step g; two ticks for a doubling step, one for an addition.
inline void lm_miller_keyed_model(Fp12& f, const G1Affine& pa, bool key_inf, const int32_t (*tab)[2][2][BN_LIMBS], const G1Affine& pb) {
  static LmHostBox bx;
  static NnLane<LmHostBox> L[9];
  for (int i = 0; i < LS_COUNT; ++i) bx.s[i] = fp2_zero();
  for (unsigned p = 0; p < 9; ++p) nn_lane_roles(L[p], bx, p, true);
  const bool skip_a = pa.inf || key_inf, skip_b = pb.inf;
  bx.s[LS_ONE] = fp2_one();
  bx.s[LS_XI] = fp2_load_const(C_XI_MONT);
  bx.s[LS_PAX] = fp2_from_fp(pa.x); bx.s[LS_PAY] = fp2_from_fp(pa.y); bx.s[LS_PBX] = fp2_from_fp(pb.x); bx.s[LS_PBY] = fp2_from_fp(pb.y);
  bx.s[LS_ACC] = fp2_one();
  signed char ty[BN_N_FIXED_LINES + 2];
  lm_step_types(ty);
  for (int g = -2; g < BN_N_FIXED_LINES; ++g) {
    const int ty_f = g >= 0 ? ty[g] : 4;
    const int ticks = ty_f == 0 ? 2 : 1;
    for (int tick = 0; tick < ticks; ++tick) {
      if (tick == 0 && g + 2 < BN_N_FIXED_LINES) {                     // wave LA
        const int k = g + 2;
        bx.s[LS_KC0] = fp2_load_const(tab[k][0]); bx.s[LS_KC1] = fp2_load_const(tab[k][1]);
        bx.s[LS_MC0] = fp2_load_const(C_NEG_G2_LINES[k][0]); bx.s[LS_MC1] = fp2_load_const(C_NEG_G2_LINES[k][1]);
        lm_host_level(bx, LM_K_EVAL[0], (unsigned)k & 1u, skip_a, skip_b);
      }
      if (tick == 0 && g + 1 >= 0 && g + 1 < BN_N_FIXED_LINES) lm_host_level(bx, LM_K_PROD[0], (unsigned)(g + 1) & 1u, skip_a, skip_b);   // wave LB
      if (ty_f != 4) {                                                   // waves F
        Fp2* lp = &bx.s[LS_LP + ((unsigned)g & 1u) * LS_REL_N];
        if (ty_f == 0) { if (tick == 0) lm_host_mul(bx, L, &bx.s[LS_ACC]); else lm_host_mul(bx, L, lp); }
        else lm_host_mul(bx, L, lp);
      }
    }
  }
  f.c0.c0 = bx.s[LS_ACC]; f.c0.c1 = bx.s[LS_ACC + 1]; f.c0.c2 = bx.s[LS_ACC + 2];
  f.c1.c0 = bx.s[LS_ACC + 3]; f.c1.c1 = bx.s[LS_ACC + 4]; f.c1.c2 = bx.s[LS_ACC + 5];
}
#endif

// ---- the G2 subgroup test of a decode (bn254_curve.h: g2_in_subgroup; what AffineG2::new enforces, /root/reference/src/utils.rs:113) with
// its ladder [u]P in wave T's level tables (k_g2_subgroup_lm): 62 doublings of two levels, 22 additions of three, then the tail of the
// test in ordinary pair-layout code on the Jacobian form of the result.  The addition formulas are the incomplete ones of the Miller loop: they
// degenerate exactly when the running point meets +-P, i.e. when [k]P = +-P for some 1 < k < u — impossible for a point of order r (k +- 1
// < r is no multiple of r), so P is then outside the subgroup; a degenerate step leaves Z = 0 (T = -P: (0, Y, 0); T = P: (0, 0, 0)), Z = 0
// survives every later level, and the tail answers "outside" for [u]P = identity and P != identity: the verdict is right on every input.
template <class Box> BN_DEV void lm_subgroup_init(Box& bx, const G2Affine& p) {
  bx.put(bx.slot(LS_ONE), fp2_one()); bx.put(bx.slot(LS_B3), fp2_load_const(C_TWIST_3B));
  bx.put(bx.slot(LS_PKX), p.x); bx.put(bx.slot(LS_PKY), p.y); bx.put(bx.slot(LS_NPKY), fp2_neg(p.y));
  bx.put(bx.slot(LS_TX), p.x); bx.put(bx.slot(LS_TY), p.y); bx.put(bx.slot(LS_TZ), fp2_one()); bx.put(bx.slot(LS_TW), fp2_load_const(C_TWIST_3B));
}
// the verdict from the ladder's result (projective x = X / Z, y = Y / Z -> Jacobian (X Z, Y Z^2, Z))
BN_DEV bool lm_subgroup_verdict(const G2Affine& p, const Fp2& X, const Fp2& Y, const Fp2& Z) {
  G2Jac up;
  const Fp2 z2 = fp2_sqr(Z);
  up.x = fp2_mul(X, Z); up.y = fp2_mul(Y, z2); up.z = Z;
  return g2_in_subgroup_tail(p, up);
}
#if !defined(__HIPCC__)
inline bool lm_g2_subgroup_model(const G2Affine& p) {
  static LmHostBox bx;
  for (int i = 0; i < LS_COUNT; ++i) bx.s[i] = fp2_zero();
  lm_subgroup_init(bx, p);
  for (int i = 0; i < BN_U_NAF_LEN; ++i) {
    lm_host_level(bx, LM_T_DBL[0], 0, false, false);
    lm_host_level(bx, LM_T_DBL[1], 0, false, false);
    const int d = C_U_NAF[i];
    if (d != 0) {
      bx.s[LS_TQX] = bx.s[LS_PKX]; bx.s[LS_TQY] = bx.s[d > 0 ? LS_PKY : LS_NPKY];
      for (int l = 0; l < 3; ++l) lm_host_level(bx, LM_T_ADD[l], 0, false, false);
    }
  }
  return lm_subgroup_verdict(p, bx.s[LS_TX], bx.s[LS_TY], bx.s[LS_TZ]);
}
#endif

}  // namespace bn254
