// Translation unit of libbn254hip.so: DEVELOPER HOOKS — the element-wise test entry points the parity tests compare layer by layer with the
// oracle (bn254_debug_*) and the measurement entry points behind bench.py's roofline figures (bn254_probe_*).  Not part of the drop-in ABI
// (include/bn254_hip.h, section BN254_DEV_HOOKS): no reference function corresponds to any of them.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "../../include/bn254_hip.h"
#include "bn254_hash.h"
#include "bn254_io.h"
#include "bn254_pairing.h"

using namespace bn254;

#include "bn254_ws.h"
#include "bn254_lane.h"
#include "bn254_host.h"

// --- test hooks ---------------------------------------------------------------------------
KERNEL_SMALL void k_debug_fp_op(int op, const uint8_t* a, const uint8_t* b, size_t n, uint8_t* out, uint8_t* status) {
  size_t i = (size_t)blockIdx.x * BN_WAVE + threadIdx.x;
  if (i >= n) return;
  uint32_t any = 0;
  Fp x, y, r;
  bool ok = fp_from_be(x, a + 32 * i, any);
  if (b) ok = fp_from_be(y, b + 32 * i, any) && ok; else y = fp_zero();
  uint8_t st = ok ? ST_OK : ST_NOT_MEMBER;
  switch (op) {
    case 0: r = fp_mul(x, y); break;
    case 1: r = fp_add(x, y); break;
    case 2: r = fp_sub(x, y); break;
    case 3: r = fp_inv(x); break;
    case 4: r = fp_sqr(x); break;
    default: if (!fp_sqrt(r, x) && st == ST_OK) st = ST_NOT_MEMBER; break;
  }
  fp_to_be(out + 32 * i, r);
  status[i] = st;
}
// The try loop's treatment of ONE chosen digest value (32 B big-endian): range rules + mod_u256, the Jacobi filter of
// k_hash_round and the square root of k_hash_finish.  status 0 = yields the point written to out, 1 = next counter;
// bit 7 set = filter and square root disagree (never expected).
KERNEL_SMALL void k_debug_hash_candidate(const uint8_t* h, size_t n, uint8_t* out, uint8_t* status) {
  size_t i = (size_t)blockIdx.x * BN_WAVE + threadIdx.x;
  if (i >= n) return;
  U256 x;
  const uint32_t* w = (const uint32_t*)(h + 32 * i);
#pragma unroll
  for (int k = 0; k < 8; ++k) x.w[7 - k] = __builtin_bswap32(w[k]);
  bool cand = hash_reduce_candidate(x);
  bool filt = false, ok = false;
  G1Affine p;
  g1_set_generator(p);
  if (cand) {
    Fp xm, rhs;
    hash_curve_rhs(xm, rhs, x);
    filt = u256_is_square_mod_q(fp_to_u256(rhs));
    ok = hash_point_from_candidate(p, x);
  }
  if (!ok) p.inf = true;
  encode_g1(out + 64 * i, p);
  status[i] = (uint8_t)((ok ? 0 : 1) | (filt != ok ? 0x80 : 0));
}
__device__ __forceinline__ void decode_fp12(Fp12& f, const uint8_t* b) {
  uint32_t any = 0;
  Fp2* c[6] = {&f.c0.c0, &f.c0.c1, &f.c0.c2, &f.c1.c0, &f.c1.c1, &f.c1.c2};
  for (int k = 0; k < 6; ++k) { fp_from_be(c[k]->c0, b + 64 * k, any); fp_from_be(c[k]->c1, b + 64 * k + 32, any); }
}
KERNEL void k_debug_fp12_op(int op, const uint8_t* a, const uint8_t* b, size_t n, uint8_t* out) {
  size_t i = (size_t)blockIdx.x * BN_WAVE + threadIdx.x;
  if (i >= n) return;
  Fp12 x, y, r;
  decode_fp12(x, a + 384 * i);
  if (b) decode_fp12(y, b + 384 * i); else fp12_set_one(y);
  switch (op) {
    case 0: fp12_mul(r, x, y); break;
    case 1: fp12_sqr(r, x); break;
    case 2: fp12_inv(r, x); break;
    case 3: fp12_conj(r, x); break;
    case 4: fp12_frob(r, x, 1); break;
    case 5: fp12_frob(r, x, 2); break;
    case 6: fp12_frob(r, x, 3); break;
    case 7: fp12_cyclotomic_sqr(r, x); break;
    default: { Fp12 acc; final_exponentiation(r, x, acc); } break;
  }
  encode_fp12(out + 384 * i, r);
}

// test hook: LIMB vectors straight into the F planes of the workspace (12 coefficients x 9 int32 limbs per item, Gt order) — the input of a
// final exponentiation with non-canonical / extreme-digit representatives that no byte decoder would produce
KERNEL_SMALL void k_debug_load_f(const int32_t* limbs, size_t n, Ws ws) {
  size_t i = (size_t)blockIdx.x * BN_WAVE + threadIdx.x;
  if (i >= n) return;
  for (int e = 0; e < 12; ++e) {
    Fp x;
#pragma unroll
    for (int k = 0; k < BN_LIMBS; ++k) x.v[k] = limbs[(i * 12 + e) * BN_LIMBS + k];
    ws_store_fp(ws, PL_F0 + e, i, x);
  }
  ws_byte(ws, BY_ST_DECODE, i) = ST_OK;
  ws_byte(ws, BY_ST_HASH, i) = ST_OK;
}

// ---- in-process issue-rate probe (bench.py's roofline calibration) --------------------------------------------
// 16 independent chains of one instruction, 4096 trips, on every SIMD of the device with `waves_per_simd` waves each
// (256-thread workgroups = one wave per SIMD of a CU, like the pair kernels).  op 0: v_mad_u64_u32, 1: v_add_u32,
// 2: v_mul_lo_u32.  The standalone sweep over more instructions is bn254_amd/csrc/microbench/valu_rates.hip.
#define PROBE_ITERS 4096
#define PROBE_CHAINS 16
template <int OP>
__global__ void __launch_bounds__(256) k_issue_probe(uint32_t* out, uint32_t seed, unsigned long long* clk) {
  unsigned long long clk0 = 0, wall0 = 0;
  if (clk && threadIdx.x == 0) { clk0 = clock64(); wall0 = wall_clock64(); }
  uint32_t a = seed + threadIdx.x * 2654435761u, b = seed ^ (threadIdx.x * 40503u + 977u);
  uint64_t acc[PROBE_CHAINS];
#pragma unroll
  for (int j = 0; j < PROBE_CHAINS; ++j) acc[j] = a + j;
  for (int i = 0; i < PROBE_ITERS; ++i) {
#pragma unroll
    for (int j = 0; j < PROBE_CHAINS; ++j) {
      if (OP == 0) {
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b) : "vcc");
      } else if (OP == 1) {
        uint32_t lo = (uint32_t)acc[j];
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(lo) : "v"(a));
        acc[j] = lo;
      } else {
        uint32_t lo = (uint32_t)acc[j];
        asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(lo) : "v"(b));
        acc[j] = lo;
      }
    }
  }
  uint64_t sum = 0;
#pragma unroll
  for (int j = 0; j < PROBE_CHAINS; ++j) sum += acc[j];
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = (uint32_t)sum ^ (uint32_t)(sum >> 32);
  if (clk && threadIdx.x == 0 && blockIdx.x < BN_CLK_MAX_WG) {     // slot 2 of the clock probe (bn254_ws.h): this kernel's own clock
    unsigned long long* p = clk + ((size_t)2 * BN_CLK_MAX_WG + blockIdx.x) * 2;
    p[0] += clock64() - clk0; p[1] += wall_clock64() - wall0;
  }
}


extern "C" {

// the routing table of the context (bn254_ws.h: bn_route_table) — tests iterate its boundaries
int bn254_debug_route_table(bn254_ctx* c, uint64_t* max_n, int* miller, int* fe, int cap) {
  if (!c || !max_n || !miller || !fe || cap < 5) return BN254_E_BAD_ARGUMENT;
  size_t m[5];
  BnRoute r[5];
  const int rows = bn_route_table(route_limits(c), m, r, 5);
  for (int i = 0; i < rows; ++i) { max_n[i] = m[i] == (size_t)-1 ? UINT64_MAX : (uint64_t)m[i]; miller[i] = r[i].miller; fe[i] = r[i].fe; }
  return rows;
}

// issue-rate probe: wave-instructions per second of `op` with `waves_per_simd` waves on every SIMD, timed with HIP events
int bn254_probe_issue_rate(bn254_ctx* c, int op, int waves_per_simd, double* wave_inst_per_s, int* n_simd) {
  if (!c || !wave_inst_per_s || op < 0 || op > 2 || waves_per_simd < 1 || waves_per_simd > 8) return BN254_E_BAD_ARGUMENT;
  HIP_TRY(hipSetDevice(c->device));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, c->device));
  const int n_cu = prop.multiProcessorCount, blocks = n_cu * waves_per_simd;
  int rc;
  if ((rc = stage_reserve(c, 0, sizeof(uint32_t) * 256 * (size_t)blocks))) return rc;
  uint32_t* out = (uint32_t*)c->stage[0];
  ScopedEvents ev;                                   // destroyed on every path out, the early error returns included
  HIP_TRY(ev.create());
  hipEvent_t e0 = ev.e0, e1 = ev.e1;
  float best = 0;
  for (int rep = 0; rep < 3; ++rep) {       // first repetition warms up; keep the fastest
    HIP_TRY(hipEventRecord(e0, c->stream));
    if (op == 0) k_issue_probe<0><<<blocks, 256, 0, c->stream>>>(out, 12345u, c->ws.clk);
    else if (op == 1) k_issue_probe<1><<<blocks, 256, 0, c->stream>>>(out, 12345u, c->ws.clk);
    else k_issue_probe<2><<<blocks, 256, 0, c->stream>>>(out, 12345u, c->ws.clk);
    HIP_TRY(hipEventRecord(e1, c->stream));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    if (rep > 0 && (best == 0 || ms < best)) best = ms;
  }
  *wave_inst_per_s = (double)PROBE_ITERS * PROBE_CHAINS * 4.0 * blocks / (best * 1e-3);
  if (n_simd) *n_simd = n_cu * 4;
  return 0;
}

// Measurement: the product leaves of one verify's Miller loop alone (k_leaf_floor_pair, bn254_pair.hip) on the planes the last verify
// left in the workspace (n <= the size of that batch); ms = the kernel's duration (HIP events), best of 3 after a warm-up launch.
int bn254_probe_leaf_floor(bn254_ctx* c, size_t n, int mode, float* ms) {
  if (!c || !ms || n == 0 || n > c->ws.stride || mode < 0 || mode > 7) return BN254_E_BAD_ARGUMENT;
  HIP_TRY(hipSetDevice(c->device));
  ScopedEvents ev;                                   // destroyed on every path out, the early error returns included
  HIP_TRY(ev.create());
  hipEvent_t e0 = ev.e0, e1 = ev.e1;
  float best = 0;
  int rc = 0;
  for (int rep = 0; rep < 4 && rc == 0; ++rep) {
    HIP_TRY(hipEventRecord(e0, c->stream));
    rc = bn254_pair_leaf_floor(n, c->ws, c->stream, mode);
    HIP_TRY(hipEventRecord(e1, c->stream));
    HIP_TRY(hipEventSynchronize(e1));
    float t = 0;
    HIP_TRY(hipEventElapsedTime(&t, e0, e1));
    if (rep > 0 && (best == 0 || t < best)) best = t;
  }
  *ms = best;
  return rc;
}

// Measurement: the final exponentiation's accumulator machine on a caller-supplied program (pairs of bytes (opcode, argument), ended by
// (0, 0); opcodes 1 LOAD s, 2 STORE s, 3 CSQR, 4 MUL s, 5 CONJ, 6 FROB k, 7 INV — bn254_pairing.h) for n lane pairs, on whatever the
// F planes of the workspace hold (run a verify first).  ms = the kernel's duration, best of 3 after a warm-up launch.  The values are
// meaningless (a cyclotomic squaring of a non-cyclotomic element): this times the routines in place, it does not check them.
int bn254_probe_fe_program(bn254_ctx* c, size_t n, const uint8_t* prog, size_t n_steps, float* ms) {
  if (!c || !ms || !prog || n == 0 || n > c->ws.stride || n_steps == 0 || n_steps > 4096) return BN254_E_BAD_ARGUMENT;
  for (size_t k = 0; k < n_steps; ++k) {
    const uint8_t op = prog[2 * k], arg = prog[2 * k + 1];
    if (op == 0 || op > 7) return BN254_E_BAD_ARGUMENT;
    if ((op == 1 || op == 2 || op == 4) && arg >= (BN_FE_EXACT_SLOTS > BN_FE_CHECK_SLOTS ? BN_FE_EXACT_SLOTS : BN_FE_CHECK_SLOTS)) return BN254_E_BAD_ARGUMENT;
    if (op == 6 && (arg < 1 || arg > 3)) return BN254_E_BAD_ARGUMENT;
  }
  HIP_TRY(hipSetDevice(c->device));
  int rc;
  if ((rc = stage_reserve(c, 7, 2 * n_steps + 2))) return rc;
  static const uint8_t fe_end[2] = {0, 0};          // n_steps <= 4096: the program and its END pair go over in two small copies
  HIP_TRY(hipMemcpy(c->stage[7], prog, 2 * n_steps, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(c->stage[7] + 2 * n_steps, fe_end, 2, hipMemcpyHostToDevice));
  ScopedEvents ev;                                   // destroyed on every path out, the early error returns included
  HIP_TRY(ev.create());
  hipEvent_t e0 = ev.e0, e1 = ev.e1;
  float best = 0;
  for (int rep = 0; rep < 4 && rc == 0; ++rep) {
    HIP_TRY(hipEventRecord(e0, c->stream));
    rc = bn254_pair_fe_program(n, c->ws, c->stage[7], c->stream);
    HIP_TRY(hipEventRecord(e1, c->stream));
    HIP_TRY(hipEventSynchronize(e1));
    float t = 0;
    HIP_TRY(hipEventElapsedTime(&t, e0, e1));
    if (rep > 0 && (best == 0 || t < best)) best = t;
  }
  *ms = best;
  return rc;
}

// ---- test hooks --------------------------------------------------------------------------
int bn254_debug_fp_op(bn254_ctx* c, int op, const uint8_t* a, const uint8_t* b, size_t n, uint8_t* out, uint8_t* status) {
  if (!c || (n && (!a || !out || !status))) return BN254_E_BAD_ARGUMENT;
  if (n == 0) return 0;
  HIP_TRY(hipSetDevice(c->device));
  int rc;
  if ((rc = stage_in(c, 0, a, n * 32))) return rc;
  if (b && (rc = stage_in(c, 1, b, n * 32))) return rc;
  if ((rc = stage_reserve(c, 2, n * 32))) return rc;
  if ((rc = stage_reserve(c, 3, n))) return rc;
  k_debug_fp_op<<<grid_for(n), BN_WAVE, 0, c->stream>>>(op, c->stage[0], b ? c->stage[1] : nullptr, n, c->stage[2], c->stage[3]);
  HIP_TRY(hipGetLastError());
  if ((rc = stage_out(c, 2, out, n * 32))) return rc;
  if ((rc = stage_out(c, 3, status, n))) return rc;
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}
int bn254_debug_hash_candidate(bn254_ctx* c, const uint8_t* h, size_t n, uint8_t* out, uint8_t* status) {
  if (!c || (n && (!h || !out || !status))) return BN254_E_BAD_ARGUMENT;
  if (n == 0) return 0;
  HIP_TRY(hipSetDevice(c->device));
  int rc;
  if ((rc = stage_in(c, 0, h, n * 32))) return rc;
  if ((rc = stage_reserve(c, 2, n * 64))) return rc;
  if ((rc = stage_reserve(c, 3, n))) return rc;
  k_debug_hash_candidate<<<grid_for(n), BN_WAVE, 0, c->stream>>>(c->stage[0], n, c->stage[2], c->stage[3]);
  HIP_TRY(hipGetLastError());
  if ((rc = stage_out(c, 2, out, n * 64))) return rc;
  if ((rc = stage_out(c, 3, status, n))) return rc;
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}
// layout: 0 one lane per item, exact chain (Gt out) | 1 lane pairs, program C_FE_EXACT (Gt out) | 2 lane pairs, program C_FE_CHECK |
// 3 octet (straight-line chains below 128 items, accumulator machine from 128 on) | 4 nonet | 5 one lane per item, check chain | 6 nonet, wide form (18 lane pairs)
int bn254_debug_final_exp_limbs(bn254_ctx* c, int layout, const int32_t* limbs, size_t n, uint8_t* gt, uint8_t* status) {
  if (!c || layout < 0 || layout > 6 || (n && (!limbs || !status)) || (gt && layout > 1)) return BN254_E_BAD_ARGUMENT;
  if (n == 0) return 0;
  if ((layout == 3 && !c->fits_trio) || ((layout == 4 || layout == 6) && !bn254_nonet_fits_device())) return BN254_E_BAD_ARGUMENT;
  HIP_TRY(hipSetDevice(c->device));
  int rc;
  if ((rc = ws_reserve(c, n))) return rc;
  if ((rc = stage_in(c, 0, limbs, n * 12 * BN_LIMBS * sizeof(int32_t)))) return rc;
  if ((rc = stage_reserve(c, 1, n * 384))) return rc;
  if ((rc = stage_reserve(c, 2, n))) return rc;
  hipStream_t s = c->stream;
  uint8_t* d_gt = gt ? c->stage[1] : nullptr;
  k_debug_load_f<<<grid_for(n), BN_WAVE, 0, s>>>((const int32_t*)c->stage[0], n, c->ws);
  switch (layout) {
    case 0: { int rc_ = launch_final_exp_lane(c, s, n, 1, 1, 1, 0, d_gt ? d_gt : c->stage[1], c->stage[2], 0, 0, nullptr, nullptr); if (rc_) return rc_; } break;
    case 1: rc = bn254_pair_final_exp_product(n, 1, c->ws, d_gt ? d_gt : c->stage[1], c->stage[2], 0, s); break;
    case 2: rc = bn254_pair_final_exp(n, c->ws, 0, c->stage[2], nullptr, nullptr, s); break;
    case 3: rc = bn254_trio_final_exp(n, c->ws, 0, c->stage[2], s); break;
    case 4: rc = bn254_nonet_final_exp(n, c->ws, 0, c->stage[2], s); break;
    case 6: rc = bn254_nonet_final_exp(n, c->ws, 0, c->stage[2], s, 1); break;
    default: { int rc_ = launch_final_exp_lane(c, s, n, 1, 1, 1, 0, nullptr, c->stage[2], 0, 0, nullptr, nullptr); if (rc_) return rc_; } break;
  }
  if (rc) return rc;
  HIP_TRY(hipGetLastError());
  if (gt && (rc = stage_out(c, 1, gt, n * 384))) return rc;
  if ((rc = stage_out(c, 2, status, n))) return rc;
  HIP_TRY(hipStreamSynchronize(s));
  return 0;
}
int bn254_debug_fp12_op(bn254_ctx* c, int op, const uint8_t* a, const uint8_t* b, size_t n, uint8_t* out) {
  if (!c || (n && (!a || !out))) return BN254_E_BAD_ARGUMENT;
  if (n == 0) return 0;
  HIP_TRY(hipSetDevice(c->device));
  int rc;
  if ((rc = stage_in(c, 0, a, n * 384))) return rc;
  if (b && (rc = stage_in(c, 1, b, n * 384))) return rc;
  if ((rc = stage_reserve(c, 2, n * 384))) return rc;
  k_debug_fp12_op<<<grid_for(n), BN_WAVE, 0, c->stream>>>(op, c->stage[0], b ? c->stage[1] : nullptr, n, c->stage[2]);
  HIP_TRY(hipGetLastError());
  if ((rc = stage_out(c, 2, out, n * 384))) return rc;
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}


}  // extern "C"
