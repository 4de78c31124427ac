// Device translation unit of libbn254hip.so: the MILLER LOOP of ECDSA::verify (/root/reference/src/ecdsa.rs:49-64) for the SMALLEST batches
// as the LANE MACHINE of bn254_lmachine.h — one verify on nine lane pairs in each of four waves (T: twist point, two steps ahead; L: the
// step's line times the table line, one step ahead; F0 / F1: the accumulator), three verifies per 256-lane workgroup.
//
// Against the eight wave roles of bn254_quad.hip (one lane pair per role: three products in a row per phase, six phases per doubling step,
// ~11 us per step for a lone verify) a step here is two "ticks" (three while wave T adds a point) of ONE product level each.  Levels are
// data (LmEntry: bn254_lmachine.h); the register file of a verify lives in LDS ([role][slot][9 limbs]); inside a wave the stages of a level
// are ordered by wavefront-scope fences (a wave's LDS instructions execute in order), between waves by the tick's workgroup barrier, and
// everything one wave hands another is double-buffered by the parity of the step it belongs to.
//
// k_miller_verify_lmk is the KEYED form (bn254_batch_verify_keyed: the key's lines come from its table, waves LA / LB instead of T / L, an
// addition step in one tick).  Measured (profiles/r05_k_lane_machine_wave_shares.jsonl): the loop is bound by the accumulator waves — 2.25 us
// per Fq12 product, 1.0 of it the product leaf — not by the twist point (197 levels, 0.33 ms alone) or the barriers (0.3 us per tick).
//
// Proof and parity: the same stage functions and tables run on a host box in tests/hostsim (hp_lm_verify, hp_lm_verify_keyed), under the
// interval tracker too (tests/test_pair_layout.py::test_lane_machine_*); on the device: tests/test_gpu_parity.py (every small-batch test with
// this layout on and off, test_keyed_verify_vs_oracle), tests/soak_gpu.py.
#include <hip/hip_runtime.h>

#define BN_SPLIT_FP2 1
#define BN_PAIR_SQR_DPP_ASM 1
#define BN_INLINE_FP12_HOT 1
#define BN_INLINE_FE_HOT 1
#define bn254 bn254_lm        // own namespace: the pair layout's types and routines
#include "bn254_pairing.h"
#include "bn254_nonet.h"
#include "bn254_lmachine.h"

using namespace bn254;

#include "bn254_ws.h"

#define BN_LM_WG 256
#define BN_LM_LANES 18                             // lanes per verify and wave: nine lane pairs
#define BN_LM_PER_WG 3                             // verifies per workgroup (54 of a wave's 64 lanes; lanes 54..63 follow along on copies)
#define KERNEL_LM __global__ __launch_bounds__(BN_LM_WG) __attribute__((amdgpu_waves_per_eu(1, 1)))

// dynamic LDS (words): per verify and role the register file, then the step types (bytes)
#define LM_ROLE_STRIDE (LS_COUNT * BN_LIMBS + 1)
#define LM_VERIFY_STRIDE (2 * LM_ROLE_STRIDE)
#define LM_TY_OFF (BN_LM_PER_WG * LM_VERIFY_STRIDE)
#define LM_TY_WORDS 24                             // 87 step types + 2 x "past the end", as bytes
#define LM_LDS_WORDS (LM_TY_OFF + LM_TY_WORDS)
static_assert(BN_N_FIXED_LINES + 2 <= 4 * LM_TY_WORDS, "step types");
static_assert(LM_LDS_WORDS * sizeof(int32_t) <= 160 * 1024, "lane machine: register files exceed the 160 KB of LDS of a gfx950 CU");

extern __shared__ int32_t lm_lds[];

#define LM_FENCE() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
#define LM_TICK() __syncthreads()

struct LmLdsBox {
  typedef unsigned Ref;
  typedef unsigned Rel;
  typedef unsigned Base;
  unsigned base;                 // word offset of slot 0 of the lane's verify and role
  __device__ __forceinline__ Ref slot(uint32_t id) const { return base + id * BN_LIMBS; }
  __device__ __forceinline__ Ref coef(unsigned idx) const { return slot(idx < 6 ? LS_ACC + idx : (unsigned)LS_ZERO); }
  __device__ __forceinline__ Ref xp(unsigned q) const { return slot(LS_XP0 + q); }
  __device__ __forceinline__ Ref x1(unsigned q) const { return slot(q < 9 ? LS_X1 + q : (unsigned)LS_ZERO); }
  __device__ __forceinline__ Ref zero() const { return slot(LS_ZERO); }
  __device__ __forceinline__ static Rel rel(unsigned idx) { return idx * BN_LIMBS; }
  __device__ __forceinline__ static Ref at(Base b, Rel r) { return b + r; }
  __device__ __forceinline__ Fp2 get(Ref off) const {
    Fp2 r;
#pragma unroll
    for (int i = 0; i < BN_LIMBS; ++i) r.c[0].v[i] = lm_lds[off + i];
    return r;
  }
  __device__ __forceinline__ void put(Ref off, const Fp2& x) const {
#pragma unroll
    for (int i = 0; i < BN_LIMBS; ++i) lm_lds[off + i] = x.c[0].v[i];
  }
};
struct LmLane { LmLdsBox bx; unsigned pair; bool writer, skip_a, skip_b; };

__device__ __forceinline__ int lm_ty(int k) { return (int)((const signed char*)(lm_lds + LM_TY_OFF))[k]; }

// one level of a wave T / L program: products, publish, linear combinations, publish
__device__ __forceinline__ void lm_level(const LmLane& ln, const LmEntry& e, unsigned par) {
  LmLdsBox bx = ln.bx;
  {
    const Fp2 pr = lm_stage_product(e, bx, par);
    LM_FENCE();                                   // every pair has read its operands
    if (ln.writer && (e.w[1] & 255u) != (uint32_t)LS_DUMMY) bx.put(lm_product_out(e, bx, par), pr);
    LM_FENCE();
  }
  const Fp2 li = lm_stage_linear(e, bx, par, ln.skip_a, ln.skip_b);
  LM_FENCE();
  if (ln.writer && (e.w[4] & 255u) != (uint32_t)LS_DUMMY) bx.put(lm_linear_out(e, bx, par), li);
  LM_FENCE();
}
__device__ __forceinline__ void lm_copy(const LmLane& ln, unsigned to, unsigned from) {
  const Fp2 x = ln.bx.get(ln.bx.slot(from));
  if (ln.writer) ln.bx.put(ln.bx.slot(to), x);     // identical words from every pair of the verify
}
__device__ __forceinline__ LmEntry lm_load_entry(const LmEntry* table, unsigned pair) {
  LmEntry e;
#pragma unroll
  for (int j = 0; j < 5; ++j) e.w[j] = table[pair].w[j];
  return e;
}
#define LM_GLOBAL_STEPS_BEGIN (-2)

// ---- wave T: the twist point, two steps ahead of the accumulator
__device__ __noinline__ void lm_wave_t(const LmLane ln) {
  const LmEntry e_i = lm_load_entry(LM_T_INIT[0], ln.pair);
  const LmEntry e_d0 = lm_load_entry(LM_T_DBL[0], ln.pair), e_d1 = lm_load_entry(LM_T_DBL[1], ln.pair);
  const LmEntry e_a0 = lm_load_entry(LM_T_ADD[0], ln.pair), e_a1 = lm_load_entry(LM_T_ADD[1], ln.pair), e_a2 = lm_load_entry(LM_T_ADD[2], ln.pair);
  lm_level(ln, e_i, 0);
  LM_TICK();
#pragma clang loop unroll(disable)
  for (int g = LM_GLOBAL_STEPS_BEGIN; g < BN_N_FIXED_LINES; ++g) {
    const int ty = lm_ty(g + 2);
    const unsigned par = (unsigned)(g + 2) & 1u;
    if (ty == 0) {
      lm_level(ln, e_d0, par);
      LM_TICK();
      lm_level(ln, e_d1, par);
      LM_TICK();
    } else if (ty != 4) {
      lm_copy(ln, LS_TQX, (unsigned)lm_q_x(ty)); lm_copy(ln, LS_TQY, (unsigned)lm_q_y(ty));
      LM_FENCE();
      lm_level(ln, e_a0, par);
      LM_TICK();
      lm_level(ln, e_a1, par);
      LM_TICK();
      lm_level(ln, e_a2, par);
      LM_TICK();
    } else {
      LM_TICK();
      LM_TICK();
    }
  }
}
// ---- wave L: the step's line at pair A's G1 point times the table line at pair B's, one step ahead
// (measured: the tables copied into LDS once per workgroup instead of a fetch a step ahead — neutral for the keyed form, +8 us for this one)
__device__ __forceinline__ Fp2 lm_table_const(int k, int j) { return fp2_load_const(C_NEG_G2_LINES[k < BN_N_FIXED_LINES ? k : 0][j]); }
__device__ __noinline__ void lm_wave_l(const LmLane ln) {
  const LmEntry e_d = lm_load_entry(LM_L_DBL[0], ln.pair), e_a = lm_load_entry(LM_L_ADD[0], ln.pair), e_p = lm_load_entry(LM_L_PROD[0], ln.pair);
  Fp2 mc0 = lm_table_const(0, 0), mc1 = lm_table_const(0, 1);          // of the step it works on next (fetched a step early: global memory)
  LM_TICK();
#pragma clang loop unroll(disable)
  for (int g = LM_GLOBAL_STEPS_BEGIN; g < BN_N_FIXED_LINES; ++g) {
    const int ticks = lm_ticks(lm_ty(g + 2));
    const int ty = g + 1 >= 0 ? lm_ty(g + 1) : 4;
    const unsigned par = (unsigned)(g + 1) & 1u;
    if (ty != 4) {
      if (ln.writer) { ln.bx.put(ln.bx.slot(LS_MC0), mc0); ln.bx.put(ln.bx.slot(LS_MC1), mc1); }
      if (ty != 0) { lm_copy(ln, LS_LQX, (unsigned)lm_q_x(ty)); lm_copy(ln, LS_LQY, (unsigned)lm_q_y(ty)); }
      LM_FENCE();
      mc0 = lm_table_const(g + 2, 0); mc1 = lm_table_const(g + 2, 1);
      lm_level(ln, ty == 0 ? e_d : e_a, par);
      LM_TICK();
      lm_level(ln, e_p, par);
      LM_TICK();
    } else {
      LM_TICK();
      LM_TICK();
    }
    if (ticks == 3) LM_TICK();
  }
}
// ---- waves F0 / F1: the accumulator.  acc <- acc * b as the nonet layout's Karatsuba product: round R of its two rounds of nine products
// in this wave (`start`, published before the tick's barrier), the coefficient levels in both waves (`finish`, behind the barrier —
// identical words from both).  The products of consecutive multiplications alternate between two buffers: a wave may start the next
// product while the other still reads the last one's.
__device__ __forceinline__ void lm_f_shift(NnLane<LmLdsBox>& L, unsigned off) {
  L.m_pub[0] += off; L.m_pub[1] += off; L.l1_k += off; L.l1_a += off; L.l1_b += off; L.l1_d += off;
}
template <int R> __device__ __forceinline__ void lm_f_start(const LmLane& ln, const NnLane<LmLdsBox>& L, unsigned b) {
  LmLdsBox bx = ln.bx;
  const Fp2 pr = nn_mul_product(L, bx, bx.slot(LS_ACC), b, (unsigned)R);
  LM_FENCE();
  if (ln.writer) bx.put(L.m_pub[R], pr);
}
__device__ __forceinline__ void lm_f_finish(const LmLane& ln, const NnLane<LmLdsBox>& L) {
  LmLdsBox bx = ln.bx;
  {
    const Fp2 c = nn_mul_level1(L, bx);
    LM_FENCE();
    if (ln.writer) bx.put(L.l1_pub, c);
    LM_FENCE();
  }
  const Fp2 o = nn_mul_level2(L, bx);
  LM_FENCE();
  if (L.publishes_out) bx.put(L.out_coef, o);
  LM_FENCE();
}
template <int R, bool KEYED> __device__ __noinline__ void lm_wave_f(const LmLane ln) {
  LmLdsBox bx = ln.bx;
  NnLane<LmLdsBox> L0, L1;
  nn_lane_roles<LmLdsBox>(L0, bx, ln.pair, ln.writer);
  L1 = L0;
  lm_f_shift(L1, (unsigned)(LS_XP1 - LS_XP0) * BN_LIMBS);
  int pending = -1;                                 // buffer of the product waiting for its coefficient levels
  unsigned nmul = 0;
  LM_TICK();
#pragma clang loop unroll(disable)
  for (int g = LM_GLOBAL_STEPS_BEGIN; g < BN_N_FIXED_LINES; ++g) {
    const int ty = g >= 0 ? lm_ty(g) : 4;
    const int ticks = KEYED ? (ty == 0 ? 2 : 1) : lm_ticks(lm_ty(g + 2));
    const unsigned lp = bx.slot((unsigned)LS_LP + ((unsigned)g & 1u) * (unsigned)LS_REL_N);
#pragma clang loop unroll(disable)
    for (int tick = 0; tick < ticks; ++tick) {
      if (pending == 0) lm_f_finish(ln, L0); else if (pending == 1) lm_f_finish(ln, L1);
      pending = -1;
      const bool sq = ty == 0 && tick == 0, ml = (ty == 0 && tick == 1) || (ty != 0 && ty != 4 && tick == 0);
      if (sq || ml) {
        const unsigned b = sq ? bx.slot(LS_ACC) : lp;
        if (nmul & 1u) lm_f_start<R>(ln, L1, b); else lm_f_start<R>(ln, L0, b);
        pending = (int)(nmul & 1u);
        ++nmul;
      }
      LM_TICK();
    }
  }
  if (pending == 0) lm_f_finish(ln, L0); else if (pending == 1) lm_f_finish(ln, L1);
}

KERNEL_LM void k_miller_verify_lm(size_t n, Ws ws, int mode) {
  const unsigned w = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // the wave's role: F0, F1, T, L
  const unsigned l = threadIdx.x & (BN_WAVE - 1);
  const unsigned v = l / BN_LM_LANES;
  LmLane ln;
  ln.writer = v < BN_LM_PER_WG;
  const unsigned vslot = ln.writer ? v : BN_LM_PER_WG - 1;
  ln.pair = (l % BN_LM_LANES) >> 1;
  const unsigned role = l & 1u;
  size_t i = (size_t)blockIdx.x * BN_LM_PER_WG + vslot;
  const bool live = ln.writer && i < n;
  if (i >= n) i = n - 1;                                 // lanes without a verify of their own follow along on the last one
  ln.bx.base = vslot * LM_VERIFY_STRIDE + role * LM_ROLE_STRIDE;
  for (unsigned k = threadIdx.x; k < LM_LDS_WORDS; k += BN_LM_WG) lm_lds[k] = 0;
  __syncthreads();
  // inputs -> slots, every lane of a wave the words of its verify and role (identical from every pair); the step types
  G1Affine sig, h;
  ws_load_g1(ws, PL_P1X, BY_P1_INF, i, sig);
  if (mode == 1) { h.x = fp_load_const(C_G1_GEN[0]); h.y = fp_load_const(C_G1_GEN[1]); h.inf = false; }   // uniform
  else ws_load_g1(ws, PL_P2X, BY_P2_INF, i, h);
  const bool pk_inf = ws_byte(ws, BY_Q_INF, i) != 0;
  if (mode == 2) h = sig;                                // a single pair e(P1, Q) (bn254_batch_pairing*): the fixed pair is skipped
  ln.skip_a = h.inf || pk_inf; ln.skip_b = sig.inf || mode == 2;
  if (w == 0) {
    if (ln.writer) { ln.bx.put(ln.bx.slot(LS_ONE), fp2_one()); ln.bx.put(ln.bx.slot(LS_ACC), fp2_one()); ln.bx.put(ln.bx.slot(LS_B3), fp2_load_const(C_TWIST_3B)); }
    // step types: doubling d sits behind the d doublings and the additions before it
    const int dg = (int)C_ATE_NAF[l];
    const unsigned long long nz = __builtin_amdgcn_ballot_w64(dg != 0);
    const int at = (int)l + __builtin_popcountll(nz & ((1ull << l) - 1ull));
    signed char* ty = (signed char*)(lm_lds + LM_TY_OFF);
    ty[at] = 0;
    if (dg != 0) ty[at + 1] = (signed char)dg;
    if (l == 0) { ty[BN_N_FIXED_LINES - 2] = 2; ty[BN_N_FIXED_LINES - 1] = 3; ty[BN_N_FIXED_LINES] = 4; ty[BN_N_FIXED_LINES + 1] = 4; }
  } else if (w == 1) {
    if (ln.writer) {
      ln.bx.put(ln.bx.slot(LS_FX1), fp2_load_const(C_TW_FROB_X1)); ln.bx.put(ln.bx.slot(LS_FY1), fp2_load_const(C_TW_FROB_Y1));
      ln.bx.put(ln.bx.slot(LS_FX2), fp2_load_const(C_TW_FROB_X2));
    }
  } else if (w == 2) {
    Fp2 x, y;
    x.c[0] = ws_load_fp(ws, PL_QX0 + (int)role, i); y.c[0] = ws_load_fp(ws, PL_QY0 + (int)role, i);
    if (ln.writer) {
      ln.bx.put(ln.bx.slot(LS_PKX), x); ln.bx.put(ln.bx.slot(LS_PKY), y);
      ln.bx.put(ln.bx.slot(LS_CPKX), fp2_conj(x)); ln.bx.put(ln.bx.slot(LS_CPKY), fp2_conj(y));
    }
  } else {
    if (ln.writer) {
      ln.bx.put(ln.bx.slot(LS_PAX), fp2_from_fp(h.x)); ln.bx.put(ln.bx.slot(LS_PAY), fp2_from_fp(h.y));
      ln.bx.put(ln.bx.slot(LS_PBX), fp2_from_fp(sig.x)); ln.bx.put(ln.bx.slot(LS_PBY), fp2_from_fp(sig.y));
    }
  }
  __syncthreads();
  switch (w) {
    case 0: lm_wave_f<0, false>(ln); break;
    case 1: lm_wave_f<1, false>(ln); break;
    case 2: lm_wave_t(ln); break;
    default: lm_wave_l(ln); break;
  }
  if (w != 0 || !live || ln.pair >= 6) return;           // wave F0 writes the Miller value out: pair k coefficient k
  ws_store_fp(ws, PL_F0 + 2 * (int)ln.pair + (int)role, i, ln.bx.get(ln.bx.slot(LS_ACC + ln.pair)).c[0]);
}

// ---- KEYED form (registered public keys, bn254_ctx_register_keys: the 87 lines of a key's Miller loop tabulated in the c2 = 1 form): no
// twist point to walk — the waves are F0, F1, LA (both table lines scaled by their G1 points, two steps ahead) and LB (their product, one
// step ahead); every wave has at most one level per tick, so an addition step is ONE tick: 153 ticks instead of 201.
__device__ __forceinline__ Fp2 lm_key_const(const int32_t* key_lines, int k, int coef) {
  Fp2 r;
  const int32_t* w = key_lines + ((size_t)(k < BN_N_FIXED_LINES ? k : 0) * 2 + (size_t)coef) * 2 * BN_LIMBS + (threadIdx.x & 1u) * BN_LIMBS;
#pragma unroll
  for (int i = 0; i < BN_LIMBS; ++i) r.c[0].v[i] = w[i];
  return r;
}
__device__ __noinline__ void lm_wave_ka(const LmLane ln, const int32_t* key_lines) {
  const LmEntry e_ev = lm_load_entry(LM_K_EVAL[0], ln.pair);
  Fp2 kc0 = lm_key_const(key_lines, 0, 0), kc1 = lm_key_const(key_lines, 0, 1), mc0 = lm_table_const(0, 0), mc1 = lm_table_const(0, 1);
  LM_TICK();
#pragma clang loop unroll(disable)
  for (int g = LM_GLOBAL_STEPS_BEGIN; g < BN_N_FIXED_LINES; ++g) {
    const int k = g + 2;
    if (k < BN_N_FIXED_LINES) {
      if (ln.writer) { ln.bx.put(ln.bx.slot(LS_KC0), kc0); ln.bx.put(ln.bx.slot(LS_KC1), kc1); ln.bx.put(ln.bx.slot(LS_MC0), mc0); ln.bx.put(ln.bx.slot(LS_MC1), mc1); }
      LM_FENCE();
      kc0 = lm_key_const(key_lines, k + 1, 0); kc1 = lm_key_const(key_lines, k + 1, 1); mc0 = lm_table_const(k + 1, 0); mc1 = lm_table_const(k + 1, 1);
      lm_level(ln, e_ev, (unsigned)k & 1u);
    }
    LM_TICK();
    if (g >= 0 && lm_ty(g) == 0) LM_TICK();
  }
}
__device__ __noinline__ void lm_wave_kb(const LmLane ln) {
  const LmEntry e_pr = lm_load_entry(LM_K_PROD[0], ln.pair);
  LM_TICK();
#pragma clang loop unroll(disable)
  for (int g = LM_GLOBAL_STEPS_BEGIN; g < BN_N_FIXED_LINES; ++g) {
    const int k = g + 1;
    if (k >= 0 && k < BN_N_FIXED_LINES) lm_level(ln, e_pr, (unsigned)k & 1u);
    LM_TICK();
    if (g >= 0 && lm_ty(g) == 0) LM_TICK();
  }
}
KERNEL_LM void k_miller_verify_lmk(size_t n, Ws ws, const uint32_t* key_idx, KeyTable kt) {
  const unsigned w = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // the wave's role: F0, F1, LA, LB
  const unsigned l = threadIdx.x & (BN_WAVE - 1);
  const unsigned v = l / BN_LM_LANES;
  LmLane ln;
  ln.writer = v < BN_LM_PER_WG;
  const unsigned vslot = ln.writer ? v : BN_LM_PER_WG - 1;
  ln.pair = (l % BN_LM_LANES) >> 1;
  const unsigned role = l & 1u;
  size_t i = (size_t)blockIdx.x * BN_LM_PER_WG + vslot;
  const bool live = ln.writer && i < n;
  if (i >= n) i = n - 1;
  ln.bx.base = vslot * LM_VERIFY_STRIDE + role * LM_ROLE_STRIDE;
  for (unsigned k = threadIdx.x; k < LM_LDS_WORDS; k += BN_LM_WG) lm_lds[k] = 0;
  __syncthreads();
  // the key (as k_miller_verify_keyed_pair: an index out of range or a refused key sets the tuple's status and walks the loop as a skipped pair)
  uint32_t key = key_idx[i];
  uint8_t kst = ST_OK;
  if (key >= kt.n_keys) { kst = ST_INDEX_OOB; key = 0; }
  else kst = kt.st[key];
  const bool key_inf = kst != ST_OK || kt.inf[key] != 0;
  G1Affine sig, h;
  ws_load_g1(ws, PL_P1X, BY_P1_INF, i, sig);
  ws_load_g1(ws, PL_P2X, BY_P2_INF, i, h);
  ln.skip_a = h.inf || key_inf; ln.skip_b = sig.inf;
  if (w == 0) {
    if (live && ln.pair == 0 && role == 0) {
      const uint8_t prev = ws_byte(ws, BY_ST_DECODE, i);
      ws_byte(ws, BY_ST_DECODE, i) = prev != ST_OK ? prev : kst;
    }
    if (ln.writer) { ln.bx.put(ln.bx.slot(LS_ONE), fp2_one()); ln.bx.put(ln.bx.slot(LS_ACC), fp2_one()); ln.bx.put(ln.bx.slot(LS_XI), fp2_load_const(C_XI_MONT)); }
    const int dg = (int)C_ATE_NAF[l];
    const unsigned long long nz = __builtin_amdgcn_ballot_w64(dg != 0);
    const int at = (int)l + __builtin_popcountll(nz & ((1ull << l) - 1ull));
    signed char* ty = (signed char*)(lm_lds + LM_TY_OFF);
    ty[at] = 0;
    if (dg != 0) ty[at + 1] = (signed char)dg;
    if (l == 0) { ty[BN_N_FIXED_LINES - 2] = 2; ty[BN_N_FIXED_LINES - 1] = 3; ty[BN_N_FIXED_LINES] = 4; ty[BN_N_FIXED_LINES + 1] = 4; }
  } else if (w == 3) {
    if (ln.writer) {
      ln.bx.put(ln.bx.slot(LS_PAX), fp2_from_fp(h.x)); ln.bx.put(ln.bx.slot(LS_PAY), fp2_from_fp(h.y));
      ln.bx.put(ln.bx.slot(LS_PBX), fp2_from_fp(sig.x)); ln.bx.put(ln.bx.slot(LS_PBY), fp2_from_fp(sig.y));
    }
  }
  __syncthreads();
  switch (w) {
    case 0: lm_wave_f<0, true>(ln); break;
    case 1: lm_wave_f<1, true>(ln); break;
    case 2: lm_wave_ka(ln, kt.lines + (size_t)key * BN_N_FIXED_LINES * BN_KEY_LINE_WORDS); break;
    default: lm_wave_kb(ln); break;
  }
  if (w != 0 || !live || ln.pair >= 6) return;
  ws_store_fp(ws, PL_F0 + 2 * (int)ln.pair + (int)role, i, ln.bx.get(ln.bx.slot(LS_ACC + ln.pair)).c[0]);
}
int bn254_lm_miller_verify_keyed(size_t n, Ws ws, const uint32_t* key_idx, KeyTable kt, hipStream_t s) {
  const unsigned grid = (unsigned)((n + BN_LM_PER_WG - 1) / BN_LM_PER_WG);
  k_miller_verify_lmk<<<grid, BN_LM_WG, LM_LDS_WORDS * sizeof(int32_t), s>>>(n, ws, key_idx, kt);
  HIP_TRY(hipGetLastError());
  return 0;
}

// ---- the G2 subgroup test of a decode for the smallest batches (flag bit 0; bn254_curve.h: g2_in_subgroup — [u + 1]P + psi([u]P) +
// psi^2([u]P) == psi^3([2u]P)): its ladder [u]P runs wave T's level tables — 62 doublings of two levels, 22 additions of three, 0.66 ->
// 0.4 ms for one point — in ONE wave per three points (no other wave to meet: fences only); the tail of the test is the lane-pair code.
// Why the incomplete additions cannot give a wrong verdict: bn254_lmachine.h (lm_g2_subgroup_model).  Reads the point k_decode_g2 left in the
// Q planes; a point outside the subgroup gets the decoder's status and is replaced by the generator, as in k_decode_g2_pair.
#define KERNEL_LMS __global__ __launch_bounds__(BN_WAVE) __attribute__((amdgpu_waves_per_eu(1, 2)))
KERNEL_LMS void k_g2_subgroup_lm(size_t n, Ws ws, int fail_status) {
  const unsigned l = threadIdx.x & (BN_WAVE - 1);
  const unsigned v = l / BN_LM_LANES;
  LmLane ln;
  ln.writer = v < BN_LM_PER_WG;
  const unsigned vslot = ln.writer ? v : BN_LM_PER_WG - 1;
  ln.pair = (l % BN_LM_LANES) >> 1;
  ln.skip_a = false; ln.skip_b = false;
  const unsigned role = l & 1u;
  size_t i = (size_t)blockIdx.x * BN_LM_PER_WG + vslot;
  const bool live = ln.writer && i < n;
  if (i >= n) i = n - 1;
  ln.bx.base = vslot * LM_VERIFY_STRIDE + role * LM_ROLE_STRIDE;
  for (unsigned k = threadIdx.x; k < LM_LDS_WORDS; k += BN_WAVE) lm_lds[k] = 0;
  __syncthreads();
  G2Affine q;
  q.x.c[0] = ws_load_fp(ws, PL_QX0 + (int)role, i); q.y.c[0] = ws_load_fp(ws, PL_QY0 + (int)role, i);
  q.inf = ws_byte(ws, BY_Q_INF, i) != 0;
  if (ln.writer) lm_subgroup_init(ln.bx, q);             // identical words from every pair of the point
  LM_FENCE();
  const LmEntry e_d0 = lm_load_entry(LM_T_DBL[0], ln.pair), e_d1 = lm_load_entry(LM_T_DBL[1], ln.pair);
  const LmEntry e_a0 = lm_load_entry(LM_T_ADD[0], ln.pair), e_a1 = lm_load_entry(LM_T_ADD[1], ln.pair), e_a2 = lm_load_entry(LM_T_ADD[2], ln.pair);
#pragma clang loop unroll(disable)
  for (int k = 0; k < BN_U_NAF_LEN; ++k) {
    lm_level(ln, e_d0, 0);
    lm_level(ln, e_d1, 0);
    const int d = C_U_NAF[k];                              // wave-uniform: u is a public constant
    if (d != 0) {
      lm_copy(ln, LS_TQX, LS_PKX); lm_copy(ln, LS_TQY, d > 0 ? (unsigned)LS_PKY : (unsigned)LS_NPKY);
      LM_FENCE();
      lm_level(ln, e_a0, 0);
      lm_level(ln, e_a1, 0);
      lm_level(ln, e_a2, 0);
    }
  }
  const bool in = lm_subgroup_verdict(q, ln.bx.get(ln.bx.slot(LS_TX)), ln.bx.get(ln.bx.slot(LS_TY)), ln.bx.get(ln.bx.slot(LS_TZ)));
  if (!live || ln.pair != 0 || in) return;
  ws_store_fp(ws, PL_QX0 + (int)role, i, fp2_load_const(C_G2_GEN[0]).c[0]);
  ws_store_fp(ws, PL_QY0 + (int)role, i, fp2_load_const(C_G2_GEN[1]).c[0]);
  if (role == 0) {
    ws_byte(ws, BY_Q_INF, i) = 0;
    if (ws_byte(ws, BY_ST_DECODE, i) == ST_OK) ws_byte(ws, BY_ST_DECODE, i) = (uint8_t)fail_status;   // uncompressed decode: InvalidGroupPoint; compressed: NotMember
  }
}
int bn254_lm_g2_subgroup(size_t n, Ws ws, hipStream_t s, int fail_status) {
  k_g2_subgroup_lm<<<(unsigned)((n + BN_LM_PER_WG - 1) / BN_LM_PER_WG), BN_WAVE, LM_LDS_WORDS * sizeof(int32_t), s>>>(n, ws, fail_status);
  HIP_TRY(hipGetLastError());
  return 0;
}

bool bn254_lm_fits_device() {
  int blocks = 0;
  hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, k_miller_verify_lm, BN_LM_WG, LM_LDS_WORDS * sizeof(int32_t));
  if (e != hipSuccess) { (void)hipGetLastError(); return true; }
  return blocks > 0;
}
int bn254_lm_miller_verify(size_t n, Ws ws, hipStream_t s, int mode) {
  const unsigned grid = (unsigned)((n + BN_LM_PER_WG - 1) / BN_LM_PER_WG);
  k_miller_verify_lm<<<grid, BN_LM_WG, LM_LDS_WORDS * sizeof(int32_t), s>>>(n, ws, mode);
  HIP_TRY(hipGetLastError());
  return 0;
}
