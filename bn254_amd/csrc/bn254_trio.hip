// Third device translation unit of libbn254hip.so: ECDSA::verify (/root/reference/src/ecdsa.rs:49-64) for SMALL batches,
// one verify per OCTET of lanes (four lane pairs of one wave).  Same tower / pairing source as bn254_pair.hip in the pair
// layout of the Fq2 values, but an Fq12 product runs as its four Fq6 products, one per lane pair (fp12_kmul4), a cyclotomic
// squaring as three Fq4 squarings in three pairs, each also forming the outputs that come from its square, and every Fq2
// product of a Miller-loop step in rounds of FOUR independent products, one per pair (bn254_pairing.h:
// miller_verify_rounds); results are exchanged through LDS (bn254_field.h: BN_TRIO_DEVICE); everything linear is replicated
// in the pairs.  A wave that has its SIMD to itself issues a multiply-add only every ~4.4 ns, so for a batch that cannot
// fill the chip latency is instructions per LANE.  256-thread workgroups (32 verifies), 91 KB of LDS -> one workgroup per
// CU, one wave per SIMD.  The final exponentiation of every small batch runs here; the Miller loop runs here only with
// BN254_OPT_TRIO_WAVE_ROLES = 0 — by default it runs as wave roles (bn254_quad.hip).  DESIGN.md section 4d.
#include <hip/hip_runtime.h>

#define BN_SPLIT_FP2 1
#define BN_PAIR_SQR_DPP_ASM 1
#define BN_INLINE_FP12_HOT 1
#define BN_INLINE_FE_HOT 1
#define BN_TRIO_FORMULAS 1
#define BN_TRIO_DEVICE 1
#define bn254 bn254_trio   // own namespace: same types as bn254_pair, different routines
#include "bn254_pairing.h"

using namespace bn254;

#include "bn254_ws.h"

#define BN_TRIO_FE_MACHINE_MIN_N 128               // final exponentiation of a small batch: accumulator machine from here on, straight-line chain below
#define KERNEL_TRIO __global__ __launch_bounds__(BN_TRIO_WG) __attribute__((amdgpu_waves_per_eu(1, 1)))

struct Fp12TrioSlot { Fp12 v; int32_t pad; };
static_assert(sizeof(Fp12TrioSlot) == BN_TRIO_F_WORDS * 4, "accumulator slot: 6 x 9 limbs + 1 pad word");

__device__ __forceinline__ Fp2 ws_load_fp2_role(const Ws& ws, int plane_re, size_t i) {
  Fp2 r;
  r.c[0] = ws_load_fp(ws, plane_re + (int)(threadIdx.x & 1u), i);
  return r;
}

// f = miller(H(m), pk) * miller(sig, -G2) for item = lane >> 3; the eight lanes of an octet take every branch together.
// mode 1 = check_public_keys (/root/reference/src/ecdsa.rs:80-86), as in k_miller_verify_pair.
KERNEL_TRIO void k_miller_verify_trio(size_t n, Ws ws, int mode) {
  const size_t i = ((size_t)blockIdx.x * BN_TRIO_WG + threadIdx.x) >> 3;
  if (i >= n) return;
  G1Affine sig, h;
  G2Affine pk;
  ws_load_g1(ws, PL_P1X, BY_P1_INF, i, sig);
  if (mode == 1) { h.x = fp_load_const(C_G1_GEN[0]); h.y = fp_load_const(C_G1_GEN[1]); h.inf = false; }   // wave-uniform
  else ws_load_g1(ws, PL_P2X, BY_P2_INF, i, h);
  pk.x = ws_load_fp2_role(ws, PL_QX0, i);
  pk.y = ws_load_fp2_role(ws, PL_QY0, i);
  pk.inf = ws_byte(ws, BY_Q_INF, i) != 0;
  Fp12& f = ((Fp12TrioSlot*)bn_trio_lds)[threadIdx.x].v;
  miller_verify_rounds<true>(f, h, pk, sig);        // steps and line preparation as rounds of four Fq2 products
  if (trio_pair() == 0) {                            // the four pairs hold the same f
    const Fp2* c[6] = {&f.c0.c0, &f.c0.c1, &f.c0.c2, &f.c1.c0, &f.c1.c1, &f.c1.c2};
#pragma unroll
    for (int k = 0; k < 6; ++k) ws_store_fp(ws, PL_F0 + 2 * k + (int)(threadIdx.x & 1u), i, c[k]->c[0]);
  }
}
// final exponentiation (status-only chain) + comparison with one for item = lane >> 3
// MACHINE: the chain as a program of the accumulator machine (bn254_pairing.h: fe_machine; accumulator in this lane's LDS slot,
// slots in the private segment) — 4 % faster from about a hundred verifies on, where other waves hide the slot traffic; a lone
// verify is 3 % quicker through the straight-line chain (profiles/r03_g_ab_trio_fe.log), so the launcher picks by batch size.
template <bool MACHINE>
KERNEL_TRIO void k_final_exp_trio(size_t n, Ws ws, int use_hash, uint8_t* status_out) {
  const size_t i = ((size_t)blockIdx.x * BN_TRIO_WG + threadIdx.x) >> 3;
  if (i >= n) return;
  Fp12 f;
  Fp2* c[6] = {&f.c0.c0, &f.c0.c1, &f.c0.c2, &f.c1.c0, &f.c1.c1, &f.c1.c2};
#pragma unroll
  for (int k = 0; k < 6; ++k) *c[k] = ws_load_fp2_role(ws, PL_F0 + 2 * k, i);
  uint8_t st = ws_byte(ws, BY_ST_DECODE, i);
  if (st == ST_OK && use_hash) st = ws_byte(ws, BY_ST_HASH, i);
  if constexpr (MACHINE) {
    Fp12& acc = ((Fp12TrioSlot*)bn_trio_lds)[threadIdx.x].v;
    acc = f;
    Fp12 slot[BN_FE_CHECK_SLOTS];
    fe_machine(acc, slot, C_FE_CHECK);
    f = acc;
  } else {
    final_exponentiation_check<true>(f, f, ((Fp12TrioSlot*)bn_trio_lds)[threadIdx.x].v);
  }
  const bool one = fp12_is_one(f);   // combined over the pair
  if ((threadIdx.x & 7u) == 0) status_out[i] = st != ST_OK ? st : (one ? (uint8_t)ST_OK : (uint8_t)ST_VERIFICATION_FAILED);
}

static_assert(BN_TRIO_LDS_WORDS * sizeof(int32_t) <= 160 * 1024, "octet kernels: accumulators + exchange areas exceed the 160 KB of LDS of a gfx950 CU");
bool bn254_trio_fits_device() {
  int blocks = 0;
  hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, k_final_exp_trio<true>, BN_TRIO_WG, BN_TRIO_LDS_WORDS * sizeof(int32_t));
  if (e != hipSuccess) { (void)hipGetLastError(); return true; }
  return blocks > 0;
}
int bn254_trio_miller_verify(size_t n, Ws ws, hipStream_t s, int mode) {
  k_miller_verify_trio<<<(unsigned)((8 * n + BN_TRIO_WG - 1) / BN_TRIO_WG), BN_TRIO_WG, BN_TRIO_LDS_WORDS * sizeof(int32_t), s>>>(n, ws, mode);
  HIP_TRY(hipGetLastError());
  return 0;
}
int bn254_trio_final_exp(size_t n, Ws ws, int use_hash, uint8_t* status_out, hipStream_t s) {
  const unsigned grid = (unsigned)((8 * n + BN_TRIO_WG - 1) / BN_TRIO_WG);
  if (n >= BN_TRIO_FE_MACHINE_MIN_N) k_final_exp_trio<true><<<grid, BN_TRIO_WG, BN_TRIO_LDS_WORDS * sizeof(int32_t), s>>>(n, ws, use_hash, status_out);
  else k_final_exp_trio<false><<<grid, BN_TRIO_WG, BN_TRIO_LDS_WORDS * sizeof(int32_t), s>>>(n, ws, use_hash, status_out);
  HIP_TRY(hipGetLastError());
  return 0;
}
