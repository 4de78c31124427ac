// Device-side helpers shared by the translation units of the ONE-LANE-PER-ITEM layout (bn254_hip.hip, bn254_rand.hip, bn254_group.hip,
// bn254_devhooks.hip): workspace accessors for Fq12 / G2 values, the generators a failed decode walks on with, the LDS slot of the
// Miller accumulator.  Include after bn254_pairing.h / bn254_ws.h; not for the pair-layout units (their Fq2 holds one coefficient per lane).
#pragma once
__device__ __forceinline__ void ws_store_f12(const Ws& ws, size_t i, const Fp12& f) {
  const Fp2* c[6] = {&f.c0.c0, &f.c0.c1, &f.c0.c2, &f.c1.c0, &f.c1.c1, &f.c1.c2};
#pragma unroll
  for (int k = 0; k < 6; ++k) { ws_store_fp(ws, PL_F0 + 2 * k, i, c[k]->c0); ws_store_fp(ws, PL_F0 + 2 * k + 1, i, c[k]->c1); }
}
__device__ __forceinline__ void ws_load_f12(const Ws& ws, size_t i, Fp12& f) {
  Fp2* c[6] = {&f.c0.c0, &f.c0.c1, &f.c0.c2, &f.c1.c0, &f.c1.c1, &f.c1.c2};
#pragma unroll
  for (int k = 0; k < 6; ++k) { c[k]->c0 = ws_load_fp(ws, PL_F0 + 2 * k, i); c[k]->c1 = ws_load_fp(ws, PL_F0 + 2 * k + 1, i); }
}
__device__ __forceinline__ void ws_store_g2(const Ws& ws, size_t i, const G2Affine& q) {
  ws_store_fp(ws, PL_QX0, i, q.x.c0); ws_store_fp(ws, PL_QX1, i, q.x.c1);
  ws_store_fp(ws, PL_QY0, i, q.y.c0); ws_store_fp(ws, PL_QY1, i, q.y.c1);
  ws_byte(ws, BY_Q_INF, i) = q.inf;
}
__device__ __forceinline__ void ws_load_g2(const Ws& ws, size_t i, G2Affine& q) {
  q.x.c0 = ws_load_fp(ws, PL_QX0, i); q.x.c1 = ws_load_fp(ws, PL_QX1, i);
  q.y.c0 = ws_load_fp(ws, PL_QY0, i); q.y.c1 = ws_load_fp(ws, PL_QY1, i);
  q.inf = ws_byte(ws, BY_Q_INF, i) != 0;
}
// a lane whose input failed to decode walks the rest of the pipeline on the generators so that
// every wave stays convergent; its status byte keeps the decode error.
__device__ __forceinline__ void g1_set_generator(G1Affine& p) { p.x = fp_load_const(C_G1_GEN[0]); p.y = fp_load_const(C_G1_GEN[1]); p.inf = false; }
__device__ __forceinline__ void g2_set_generator(G2Affine& q) { q.x = fp2_load_const(C_G2_GEN[0]); q.y = fp2_load_const(C_G2_GEN[1]); q.inf = false; }
// the coordinates of the generator with the identity flag untouched (a stand-in for arithmetic that must not meet (0, 0))
__device__ __forceinline__ void g2_set_generator_keep_inf(G2Affine& q) { q.x = fp2_load_const(C_G2_GEN[0]); q.y = fp2_load_const(C_G2_GEN[1]); }

// The Miller accumulator f (12 field elements = 432 B per lane) is the hottest per-lane state: every
// Fq12 squaring / line multiplication reads and rewrites it.  It is staged in LDS, one padded slot per
// lane (109 words: an odd word stride keeps the 64 lanes of a wave on distinct banks), so those
// accesses never leave the CU.  28 KB per 64-lane workgroup -> 5 workgroups per 160 KB CU.
struct Fp12Slot { Fp12 v; int32_t pad; };
static_assert(sizeof(Fp12Slot) == (12 * BN_LIMBS + 1) * 4 && ((12 * BN_LIMBS + 1) & 1), "LDS slot: 12 x 9 limbs + 1 pad word (odd stride)");
