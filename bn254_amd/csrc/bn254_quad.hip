// Fourth device translation unit of libbn254hip.so: the Miller loop of ECDSA::verify (/root/reference/src/ecdsa.rs:49-64) for SMALL
// batches with the four lane pairs of a verify as the four WAVES of a workgroup ("wave roles"; pair p of every wave = verify
// 32 * blockIdx + p).  In the octet layout of bn254_trio.hip the four pairs share a wave, so every pair executes every
// linear instruction (Karatsuba sums, recombinations, carries) — half of the Miller loop's instructions; here each wave
// runs its own instruction stream and the waves exchange Fq2 values through LDS mailboxes between workgroup barriers
// (bn254_field.h: BN_QUAD_DEVICE; roles of the Miller loop: bn254_pairing.h quad_*).  One workgroup per CU, one wave per
// SIMD; same formulas and carry sites as the octet layout (BN_TRIO_FORMULAS), proven on the host by
// the emulation builds of tests/hostsim (miller_verify_quad_model).  The final exponentiation stays in the octet layout
// (bn254_trio.hip): measured as wave roles too — an Fq12 product as four Fq6 products in four waves, a cyclotomic squaring as
// three Fq4 squares in three — 1.23 ms against 1.12 ms: per operation the two barriers with an LDS round trip between
// them cost what the replicated linear work costs inside one wave.
#include <hip/hip_runtime.h>

#define BN_SPLIT_FP2 1
#define BN_PAIR_SQR_DPP_ASM 1
#define BN_INLINE_FP12_HOT 1
#define BN_INLINE_FE_HOT 1
#define BN_TRIO_FORMULAS 1
#define BN_QUAD_DEVICE 1
#define bn254 bn254_quad   // own namespace: same types as bn254_pair / bn254_trio, different routines
#include "bn254_pairing.h"

using namespace bn254;

#include "bn254_ws.h"

#define KERNEL_QUAD __global__ __launch_bounds__(BN_QUAD_WG) __attribute__((amdgpu_waves_per_eu(1, 1)))

__device__ __forceinline__ Fp2 ws_load_fp2_role(const Ws& ws, int plane_re, size_t i) {
  Fp2 r;
  r.c[0] = ws_load_fp(ws, plane_re + (int)(threadIdx.x & 1u), i);
  return r;
}

// ---- the Miller loop of a verify as WAVE ROLES (bn254_pairing.h: quad_*): the four lane pairs of a verify in the four waves
// of a workgroup (pair p of every wave = verify 32 * blockIdx + p), Fq2 values handed over through LDS mailboxes between
// workgroup barriers — four per loop step, the same in every wave.
enum { QS_LINE = 0, QS_LP = 6, QS_AB = 11, QS_U = 14, QS_G0 = 17, QS_G1 = 20, QS_T0 = 23, QS_UU = 26, QS_T1 = 29, QS_F0 = 32, QS_F1 = 35, QS_SLOTS = 38 };
#define BN_QUAD_LDS_WORDS (QS_SLOTS * BN_QUAD_SLOT_WORDS)
// One function per role, each with the whole step loop and the SAME barrier sequence (one before the loop, four per step):
// the register allocation of a role then covers that role's values only.
struct QuadIn { Fp2 PAX, PAY, PBX, PBY; G2Affine pk; bool skip_a, skip_b, any_skip; };
// The recombinations are spread one coefficient per wave (interval after barriers 1 and 3):
//   g1 = 2ab: wave 0;  g0_0: wave 1, g0_1: wave 2, g0_2: wave 3;   f0_0, f0_1: wave 0, f0_2: wave 3;  f1_0, f1_1: wave 1, f1_2: wave 2.
// wave 0: ab = f0 f1; t0 = g0 b0
__device__ __noinline__ void quad_role0(Fp6& f0) {
  f0.c0 = fp2_one(); f0.c1 = fp2_zero(); f0.c2 = fp2_zero();
  qbox_put6(QS_F0, f0);
  QUAD_BARRIER();
  for (QuadSteps s = quad_steps_begin(); quad_step_type(s) != 4; quad_step_next(s)) {
    const bool dbl = quad_step_type(s) == 0;
    Fp6 ab, g0, t0, x;
    f0.c2 = qbox_get(QS_F0 + 2);
    if (dbl) { qbox_get6(x, QS_F1); quad_sqr_ab(ab, f0, x); qbox_put6(QS_AB, ab); }
    QUAD_BARRIER();
    if (dbl) { quad_sqr_g1(x, ab); qbox_put6(QS_G1, x); }
    QUAD_BARRIER();
    if (dbl) qbox_get6(g0, QS_G0); else g0 = f0;
    qbox_get6(x, QS_LP);
    quad_mul_t0(t0, g0, x);
    qbox_put6(QS_T0, t0);
    QUAD_BARRIER();
    f0.c0 = quad_r0_coef<0>(t0.c0, qbox_get(QS_T1 + 2));
    f0.c1 = quad_r0_coef<1>(t0.c1, qbox_get(QS_T1));
    qbox_put(QS_F0, f0.c0); qbox_put(QS_F0 + 1, f0.c1);
    QUAD_BARRIER();
  }
  f0.c2 = qbox_get(QS_F0 + 2);
}
// wave 1: u = (f0 + f1)(f0 + v f1); uu = (g0 + g1)(b0 + b1)
__device__ __noinline__ void quad_role1(Fp6& f1) {
  f1.c0 = fp2_zero(); f1.c1 = fp2_zero(); f1.c2 = fp2_zero();
  qbox_put6(QS_F1, f1);
  QUAD_BARRIER();
  for (QuadSteps s = quad_steps_begin(); quad_step_type(s) != 4; quad_step_next(s)) {
    const bool dbl = quad_step_type(s) == 0;
    Fp6 u, g0, g1, uu, x;
    f1.c2 = qbox_get(QS_F1 + 2);
    if (dbl) { qbox_get6(x, QS_F0); quad_sqr_u(u, x, f1); qbox_put(QS_U + 1, u.c1); qbox_put(QS_U + 2, u.c2); }
    QUAD_BARRIER();
    if (dbl) { g0.c0 = quad_g0_coef<0>(u.c0, qbox_get(QS_AB), qbox_get(QS_AB + 2)); qbox_put(QS_G0, g0.c0); }
    QUAD_BARRIER();
    if (dbl) { g0.c1 = qbox_get(QS_G0 + 1); g0.c2 = qbox_get(QS_G0 + 2); qbox_get6(g1, QS_G1); }
    else { qbox_get6(g0, QS_F0); g1 = f1; }
    TrioLineProduct L;
    qbox_get6(L.b0, QS_LP); L.b10 = qbox_get(QS_LP + 3); L.b11 = qbox_get(QS_LP + 4);
    quad_mul_uu(uu, g0, g1, L);
    qbox_put(QS_UU + 2, uu.c2);
    QUAD_BARRIER();
    f1.c0 = quad_r1_coef<0>(uu.c0, qbox_get(QS_T0), qbox_get(QS_T1));
    f1.c1 = quad_r1_coef<1>(uu.c1, qbox_get(QS_T0 + 1), qbox_get(QS_T1 + 1));
    qbox_put(QS_F1, f1.c0); qbox_put(QS_F1 + 1, f1.c1);
    QUAD_BARRIER();
  }
  f1.c2 = qbox_get(QS_F1 + 2);
}
// wave 2: the product of the step's line with the table line; t1 = g1 (b10 + b11 v)
__device__ __noinline__ void quad_role2(const QuadIn& in) {
  Fp2 m1 = quad_table_m1(0, in.PBX);
  QUAD_BARRIER();
  for (QuadSteps s = quad_steps_begin(); quad_step_type(s) != 4; quad_step_next(s)) {
    const bool dbl = quad_step_type(s) == 0;
    TrioLineProduct L;
    {
      const int at = QS_LINE + 3 * (s.k & 1);
      const Fp2 l0 = qbox_get(at), l1 = qbox_get(at + 1), l2 = qbox_get(at + 2);
      quad_line_product(L, l0, l1, l2, s.k, m1, in.PBY, in.skip_a, in.skip_b, in.any_skip);
      qbox_put6(QS_LP, L.b0); qbox_put(QS_LP + 3, L.b10); qbox_put(QS_LP + 4, L.b11);
    }
    QUAD_BARRIER();
    if (dbl) qbox_put(QS_G0 + 1, quad_g0_coef<1>(qbox_get(QS_U + 1), qbox_get(QS_AB + 1), qbox_get(QS_AB)));
    QUAD_BARRIER();
    Fp6 g1, t1;
    qbox_get6(g1, dbl ? QS_G1 : QS_F1);
    quad_mul_t1(t1, g1, L.b10, L.b11);
    qbox_put6(QS_T1, t1);
    if (s.k + 1 < BN_N_FIXED_LINES) m1 = quad_table_m1(s.k + 1, in.PBX);   // for the next step
    QUAD_BARRIER();
    qbox_put(QS_F1 + 2, quad_r1_coef<2>(qbox_get(QS_UU + 2), qbox_get(QS_T0 + 2), t1.c2));
    QUAD_BARRIER();
  }
}
// wave 3: the twist point, one step ahead of the others: the line of step k + 1 during step k's first interval, T's
// update during the third (the line of step 0 before the loop)
__device__ __forceinline__ void quad_t_line(const QuadSteps& st, const QuadIn& in, const G2Proj& t, const Fp2& qa_yneg, QuadDblTmp& kd, QuadAddTmp& ka) {
  const int ty = quad_step_type(st);
  if (ty == 4) return;
  Fp2 l0, l1, l2;
  if (ty == 0) quad_dbl_line(l0, l1, l2, kd, t, in.PAX, in.PAY);
  else {
    Fp2 qx = in.pk.x, qy = ty > 0 ? in.pk.y : qa_yneg;
    if (ty == 2) { qx = fp2_mul(fp2_conj(in.pk.x), fp2_load_const(C_TW_FROB_X1)); qy = fp2_mul(fp2_conj(in.pk.y), fp2_load_const(C_TW_FROB_Y1)); }
    if (ty == 3) { qx = fp2_mul(in.pk.x, fp2_load_const(C_TW_FROB_X2)); qy = in.pk.y; }
    quad_add_line(l0, l1, l2, ka, t, qx, qy, in.PAX, in.PAY);
  }
  const int at = QS_LINE + 3 * (st.k & 1);
  qbox_put(at, l0); qbox_put(at + 1, l1); qbox_put(at + 2, l2);
  if (ty != 0) quad_add_squares(ka);
}
__device__ __noinline__ void quad_role3(const QuadIn& in) {
  G2Proj t;
  t.x = in.pk.x; t.y = in.pk.y; t.z = fp2_one();
  const Fp2 qa_yneg = fp2_neg(in.pk.y);
  QuadDblTmp kd;
  QuadAddTmp ka;
  QuadSteps ahead = quad_steps_begin();
  quad_t_line(ahead, in, t, qa_yneg, kd, ka);
  if (quad_step_type(ahead) == 0) quad_dbl_update(t, kd); else quad_add_update(t, ka);
  quad_step_next(ahead);
  QUAD_BARRIER();
  for (QuadSteps s = quad_steps_begin(); quad_step_type(s) != 4; quad_step_next(s), quad_step_next(ahead)) {
    quad_t_line(ahead, in, t, qa_yneg, kd, ka);
    QUAD_BARRIER();
    if (quad_step_type(s) == 0) qbox_put(QS_G0 + 2, quad_g0_coef<2>(qbox_get(QS_U + 2), qbox_get(QS_AB + 2), qbox_get(QS_AB + 1)));
    QUAD_BARRIER();
    const int ty = quad_step_type(ahead);
    if (ty == 0) quad_dbl_update(t, kd); else if (ty != 4) quad_add_update(t, ka);
    QUAD_BARRIER();
    qbox_put(QS_F0 + 2, quad_r0_coef<2>(qbox_get(QS_T0 + 2), qbox_get(QS_T1 + 1)));
    QUAD_BARRIER();
  }
}
KERNEL_QUAD void k_miller_verify_quad(size_t n, Ws ws, int mode) {
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // the wave's role
  size_t i = (size_t)blockIdx.x * (BN_QUAD_WG / 8) + ((threadIdx.x & 63u) >> 1);
  const bool live = i < n;
  if (!live) i = n - 1;                                 // no early exit: the barriers need all four waves
  Fp6 fh;
  if (w >= 2) {
    G1Affine sig, h;
    QuadIn in;
    ws_load_g1(ws, PL_P1X, BY_P1_INF, i, sig);
    if (mode == 1) { h.x = fp_load_const(C_G1_GEN[0]); h.y = fp_load_const(C_G1_GEN[1]); h.inf = false; }   // uniform
    else ws_load_g1(ws, PL_P2X, BY_P2_INF, i, h);
    in.pk.x = ws_load_fp2_role(ws, PL_QX0, i);
    in.pk.y = ws_load_fp2_role(ws, PL_QY0, i);
    in.pk.inf = ws_byte(ws, BY_Q_INF, i) != 0;
    in.skip_a = h.inf || in.pk.inf; in.skip_b = sig.inf;
    in.any_skip = __builtin_amdgcn_ballot_w64(in.skip_a || in.skip_b) != 0;
    in.PAX = fp2_from_fp(h.x); in.PAY = fp2_from_fp(h.y); in.PBX = fp2_from_fp(sig.x); in.PBY = fp2_from_fp(sig.y);
    if (w == 2) quad_role2(in); else quad_role3(in);
    return;
  }
  if (w == 0) quad_role0(fh); else quad_role1(fh);
  if (!live) return;
  const Fp2* c[3] = {&fh.c0, &fh.c1, &fh.c2};
#pragma unroll
  for (int k = 0; k < 3; ++k) ws_store_fp(ws, PL_F0 + 6 * w + 2 * k + (int)(threadIdx.x & 1u), i, c[k]->c[0]);
}
#define BN_GFX950_LDS_BYTES (160 * 1024)
static_assert(BN_QUAD_LDS_WORDS * sizeof(int32_t) <= BN_GFX950_LDS_BYTES, "mailboxes of the four-wave kernel exceed the 160 KB of LDS of a gfx950 CU");
int bn254_quad_miller_verify(size_t n, Ws ws, hipStream_t s, int mode) {
  k_miller_verify_quad<<<(unsigned)((n + BN_QUAD_WG / 8 - 1) / (BN_QUAD_WG / 8)), BN_QUAD_WG, BN_QUAD_LDS_WORDS * sizeof(int32_t), s>>>(n, ws, mode);
  HIP_TRY(hipGetLastError());
  return 0;
}

// ---- the same loop on EIGHT waves (bn254_pairing.h: w8_*): a 512-thread workgroup, two waves per SIMD, 32 verifies; every
// Fq6 product of the f-chain, the product of the lines and the twist-point step are split over two waves each, six phases
// per doubling step (five per addition step), each closed by a barrier.  One function per role.
#define BN_W8_WG 512
#define BN_W8_LDS_WORDS (W8_SLOTS * BN_QUAD_SLOT_WORDS)
#define KERNEL_W8 __global__ __launch_bounds__(BN_W8_WG) __attribute__((amdgpu_waves_per_eu(2, 2)))
struct W8LdsBox {
  __device__ __forceinline__ Fp2 get(int slot) const { return qbox_get(slot); }
  __device__ __forceinline__ void put(int slot, const Fp2& x) const { qbox_put(slot, x); }
};
template <int ROLE> __device__ __noinline__ void w8_role(const W8In& in) {
  W8Regs r;
  W8LdsBox bx;
  constexpr bool TWIST = ROLE == W8_T0 || ROLE == W8_T1;
  w8_init<ROLE>(r, bx, in);
  QUAD_BARRIER();
  QuadSteps s = quad_steps_begin();
  if constexpr (TWIST) w8_phase_i1<ROLE>(r, bx, in, 4, -1, quad_step_type(s));     // head start: line of step 0 ...
  QUAD_BARRIER();
  if constexpr (TWIST) w8_phase_i2<ROLE>(r, bx, in, 4, -1, quad_step_type(s));     // ... and its update
  QUAD_BARRIER();
  for (; quad_step_type(s) != 4; quad_step_next(s)) {
    const int ty = quad_step_type(s);
    QuadSteps nx = s;
    quad_step_next(nx);
    const int ty_next = quad_step_type(nx);
    w8_phase_i1<ROLE>(r, bx, in, ty, s.k, ty_next);
    QUAD_BARRIER();
    w8_phase_c1<ROLE>(r, bx, in, ty, s.k, ty_next);
    QUAD_BARRIER();
    if (ty == 0) { w8_phase_g<ROLE>(r, bx); QUAD_BARRIER(); }
    w8_phase_i2<ROLE>(r, bx, in, ty, s.k, ty_next);
    QUAD_BARRIER();
    w8_phase_c2<ROLE>(r, bx);
    QUAD_BARRIER();
    w8_phase_f<ROLE>(r, bx);
    QUAD_BARRIER();
  }
}
KERNEL_W8 void k_miller_verify_w8(size_t n, Ws ws, int mode) {
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // the wave's role
  size_t i = (size_t)blockIdx.x * 32 + ((threadIdx.x & 63u) >> 1);
  const bool live = i < n;
  if (!live) i = n - 1;                                 // no early exit: the barriers need all eight waves
  {
    G1Affine sig, h;
    W8In in;
    ws_load_g1(ws, PL_P1X, BY_P1_INF, i, sig);
    if (mode == 1) { h.x = fp_load_const(C_G1_GEN[0]); h.y = fp_load_const(C_G1_GEN[1]); h.inf = false; }   // uniform
    else ws_load_g1(ws, PL_P2X, BY_P2_INF, i, h);
    in.pk.x = ws_load_fp2_role(ws, PL_QX0, i);
    in.pk.y = ws_load_fp2_role(ws, PL_QY0, i);
    in.pk.inf = ws_byte(ws, BY_Q_INF, i) != 0;
    in.pk_yneg = fp2_neg(in.pk.y);
    in.skip_a = h.inf || in.pk.inf; in.skip_b = sig.inf;
    in.any_skip = __builtin_amdgcn_ballot_w64(in.skip_a || in.skip_b) != 0;
    in.PAX = fp2_from_fp(h.x); in.PAY = fp2_from_fp(h.y); in.PBX = fp2_from_fp(sig.x); in.PBY = fp2_from_fp(sig.y);
    switch (w) {
      case W8_A0: w8_role<W8_A0>(in); break;
      case W8_A1: w8_role<W8_A1>(in); break;
      case W8_B0: w8_role<W8_B0>(in); break;
      case W8_B1: w8_role<W8_B1>(in); break;
      case W8_L0: w8_role<W8_L0>(in); break;
      case W8_L1: w8_role<W8_L1>(in); break;
      case W8_T0: w8_role<W8_T0>(in); break;
      default: w8_role<W8_T1>(in); break;
    }
  }
  if (!live || w > 1) return;                           // waves A0 / A1 write f0 / f1 out (complete behind the last barrier)
#pragma unroll
  for (int k = 0; k < 3; ++k) ws_store_fp(ws, PL_F0 + 6 * w + 2 * k + (int)(threadIdx.x & 1u), i, qbox_get((w == 0 ? W8_F0 : W8_F1) + k).c[0]);
}
static_assert(BN_W8_LDS_WORDS * sizeof(int32_t) <= BN_GFX950_LDS_BYTES, "mailboxes of the eight-wave kernel exceed the 160 KB of LDS of a gfx950 CU");
bool bn254_quad_fits_device(int eight_waves) {
  int blocks = 0;
  hipError_t e = eight_waves ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, k_miller_verify_w8, BN_W8_WG, BN_W8_LDS_WORDS * sizeof(int32_t))
                             : hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, k_miller_verify_quad, BN_QUAD_WG, BN_QUAD_LDS_WORDS * sizeof(int32_t));
  if (e != hipSuccess) { (void)hipGetLastError(); return true; }
  return blocks > 0;
}
int bn254_w8_miller_verify(size_t n, Ws ws, hipStream_t s, int mode) {
  k_miller_verify_w8<<<(unsigned)((n + 31) / 32), BN_W8_WG, BN_W8_LDS_WORDS * sizeof(int32_t), s>>>(n, ws, mode);
  HIP_TRY(hipGetLastError());
  return 0;
}
