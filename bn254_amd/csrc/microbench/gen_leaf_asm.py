#!/usr/bin/env python3
"""Emits the two product leaves of the lane-pair layout (bn254_fp2_pair.h: fp_pair_mul_impl, fp_pair_sqr_impl) as gfx950 assembly with a
FIXED register map — the same algorithm as BN_MONT_DUAL_BODY / BN_MONT_PRODUCT_BODY (product-scanning Montgomery product over nine
balanced 29-bit limbs, one 64-bit column accumulator), instruction for instruction what the compiler emits, except for the things the
compiler cannot be told:
  * the (unused) carry-out of v_mad_i64_i32 ROTATES over four SGPR pairs instead of one pair for the whole chain (two waves per SIMD:
    2.06 -> 1.91 ns per multiply-add in a synthetic stream, profiles/r01_mad_latency_microbench.jsonl);
  * no s_waitcnt vmcnt(0) lgkmcnt(0) at function entry (the callers pass everything in registers).

Calling convention = the AMDGPU C convention of the functions they replace: a in v0..v8, b in v9..v17, result in v0..v8, return address in
s[30:31]; only caller-saved registers are touched (v0-v39, v48-v55, v64-v71; s4-s29, vcc untouched).

    gen_leaf_asm.py header  > bn254_leaf_asm.h      (the leaf as one inline-asm statement: the round-4 A/B build -DBN_ASM_MUL_LEAF included it from
                                                     bn254_fp2_pair.h — bit-exact, no faster (profiles/r04_h_ab_asm_mul_leaf.log); the hook was removed
                                                     from the library sources in round 5, this generator and its output are kept as the record)
    gen_leaf_asm.py bench   > leaf_variants.hip     (the same bodies in a timing loop, carry-out rotation on / off)
"""
import sys

LIMBS, W = 9, 29
Q = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
N0 = (-pow(Q, -1, 1 << W)) % (1 << W)


def balanced_limbs(x):
    out = []
    for _ in range(LIMBS - 1):
        d = x & ((1 << W) - 1)
        if d >= 1 << (W - 1):
            d -= 1 << W
        out.append(d)
        x = (x - d) >> W
    out.append(x)
    return out


QL = balanced_limbs(Q)

# register map (all caller-saved)
A = list(range(0, 9))            # own a (argument), result r overwrites it from column 9 on
B = list(range(9, 18))           # own b (argument)
AP = list(range(18, 27))         # partner's a
X = list(range(27, 36))          # b0 in both lanes of the pair (real part of b)
Y = [36, 37, 38, 39, 48, 49, 50, 51, 52]      # +-b1
M = [53, 54, 55, 64, 65, 66, 67, 68, 69]      # Montgomery digits
ACC = 70                         # v[70:71]
ONE, MASK = 33, 34               # (sqr leaf only uses its own scratch; see below)
SQ = {i: 4 + i for i in range(LIMBS)}          # s4..s12 = limbs of q
S_N0 = 13                        # N0 << 3
S_HALF = 14                      # s[14:15] = 2^28 as a 64-bit constant
SINKS = ["s[16:17]", "s[18:19]", "s[20:21]", "s[22:23]", "s[24:25]", "s[26:27]", "s[28:29]", "vcc"]
QUAD_PARTNER = "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1"
QUAD_RE = "quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf bound_ctrl:1"
QUAD_IM = "quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1"


def reg(r):
    """a register operand: an int is a fixed VGPR, a string (an inline-asm placeholder like %3) is used as it is"""
    return r if isinstance(r, str) else "v%d" % r


class Emit:
    def __init__(self, rot):
        """rot: number of carry-out sinks the multiply-adds rotate over (True = 4, False = 1)"""
        self.lines, self.rot, self.n = [], (4 if rot is True else 1 if rot is False else int(rot)), 0
        self.A = A

    def op(self, s):
        self.lines.append("  " + s)

    def mac(self, x, y):
        """acc += x * y   (x: vgpr index; y: 'vN' or 'sN')"""
        sink = SINKS[self.n % self.rot]
        self.n += 1
        self.op("v_mad_i64_i32 v[%d:%d], %s, %s, %s, v[%d:%d]" % (ACC, ACC + 1, sink, reg(x), y if isinstance(y, str) and y[0] in "sv%" else reg(y), ACC, ACC + 1))

    def constants(self):
        for i in range(LIMBS):
            self.op("s_mov_b32 s%d, 0x%x" % (SQ[i], QL[i] & 0xFFFFFFFF))
        self.op("s_mov_b32 s%d, 0x%x" % (S_N0, (N0 << 3) & 0xFFFFFFFF))
        self.op("s_mov_b32 s%d, 0x%x" % (S_HALF, 1 << (W - 1)))
        self.op("s_mov_b32 s%d, 0" % (S_HALF + 1))

    def columns(self, terms):
        """terms(k) -> list of (x vgpr, y operand string) limb products of column k (it may emit instructions of its own first); then the
        reduction, digits / outputs"""
        self.op("v_mov_b32 v%d, 0" % ACC)
        self.op("v_mov_b32 v%d, 0" % (ACC + 1))
        for k in range(2 * LIMBS - 1):
            for x, y in terms(k):
                self.mac(x, y)
            for i in range(LIMBS):
                j = k - i
                if j < 0 or j >= LIMBS or (k < LIMBS and i >= k):
                    continue
                self.mac(M[i], "s%d" % SQ[j])
            if k < LIMBS:
                # m_k = balanced digit of (acc * N0): (acc.lo * (N0 << 3)) >> 3 arithmetic
                self.op("v_mul_lo_u32 v%d, v%d, s%d" % (M[k], ACC, S_N0))
                self.op("v_ashrrev_i32 v%d, 3, v%d" % (M[k], M[k]))
                self.mac(M[k], "s%d" % SQ[0])
                self.op("v_ashrrev_i64 v[%d:%d], %d, v[%d:%d]" % (ACC, ACC + 1, W, ACC, ACC + 1))
            else:
                r = self.A[k - LIMBS]
                self.op("v_bfe_i32 %s, v%d, 0, %d" % (reg(r), ACC, W))
                self.op("v_lshl_add_u64 v[%d:%d], v[%d:%d], 0, s[%d:%d]" % (ACC, ACC + 1, ACC, ACC + 1, S_HALF, S_HALF + 1))
                self.op("v_ashrrev_i64 v[%d:%d], %d, v[%d:%d]" % (ACC, ACC + 1, W, ACC, ACC + 1))
        self.op("v_mov_b32 %s, v%d" % (reg(self.A[LIMBS - 1]), ACC))


def pair_mul(rot, a=A, b=B, lazy=False):
    """re lane: a0*b0 + a1*(-b1)    im lane: a1*b0 + a0*b1   (own = a, partner = ap; x = b0 broadcast, y = +-b1 broadcast).
    a, b: the argument registers (fixed VGPR numbers, or inline-asm placeholders); the result overwrites a.
    lazy: the partner fetches of limb k are issued in front of column k (the first column that needs them) instead of all at the top —
    what the compiler's scheduler does with the C++ version: the plain instructions then sit BETWEEN the runs of multiply-adds"""
    e = Emit(rot)
    e.A = a
    e.op("s_nop 1")                                           # a DPP operand must not have been written by the two preceding VALU instructions
    e.constants()
    # lane parity: one = 1 - (lane & 1), mask = -one   (v_mbcnt: lane id within the wave)
    t_one, t_mask = M[8], M[7]                                # the last two digit registers are free until columns 7 and 8
    e.op("v_mbcnt_lo_u32_b32 v%d, -1, 0" % t_one)
    e.op("v_mbcnt_hi_u32_b32 v%d, -1, v%d" % (t_one, t_one))   # lane id within the wave
    e.op("v_and_b32 v%d, 1, v%d" % (t_one, t_one))
    e.op("v_sub_u32 v%d, 1, v%d" % (t_one, t_one))
    e.op("v_sub_u32 v%d, 0, v%d" % (t_mask, t_one))

    def fetch(i):
        e.op("v_mov_b32_dpp v%d, %s %s" % (AP[i], reg(a[i]), QUAD_PARTNER))
        e.op("v_mov_b32_dpp v%d, %s %s" % (X[i], reg(b[i]), QUAD_RE))
        e.op("v_mov_b32_dpp v%d, %s %s" % (Y[i], reg(b[i]), QUAD_IM))
        if lazy:
            e.op("v_xad_u32 v%d, v%d, v%d, v%d" % (Y[i], Y[i], t_mask, t_one))
    if lazy:
        # limbs 0..6 just in time; 7 and 8 before the digits m7, m8 take the registers of the parity values
        def terms(k):
            if k <= 6:
                fetch(k)
                if k == 6:
                    fetch(7); fetch(8)
            out = []
            for i in range(LIMBS):
                j = k - i
                if 0 <= j < LIMBS:
                    out.append((a[i], "v%d" % X[j]))
                    out.append((AP[i], "v%d" % Y[j]))
            return out
    else:
        for i in range(LIMBS):
            fetch(i)
        for i in range(LIMBS):
            e.op("v_xad_u32 v%d, v%d, v%d, v%d" % (Y[i], Y[i], t_mask, t_one))          # (y ^ mask) + one

        def terms(k):
            out = []
            for i in range(LIMBS):
                j = k - i
                if 0 <= j < LIMBS:
                    out.append((a[i], "v%d" % X[j]))
                    out.append((AP[i], "v%d" % Y[j]))
            return out
    e.columns(terms)
    return e.lines


def pair_sqr(rot):
    """re lane (a0 + a1)(a0 - a1), im lane (2 a1) a0:  u = own + a1,  v = a0 - (a1 in the re lane, 0 in the im lane); one product u * v"""
    e = Emit(rot)
    e.op("s_nop 1")
    e.constants()
    U, V = AP, X
    t_mask = ACC
    e.op("v_mbcnt_lo_u32_b32 v%d, -1, 0" % t_mask)
    e.op("v_mbcnt_hi_u32_b32 v%d, -1, v%d" % (t_mask, t_mask))
    e.op("v_and_b32 v%d, 1, v%d" % (t_mask, t_mask))
    e.op("v_add_u32 v%d, -1, v%d" % (t_mask, t_mask))        # re_mask = (lane & 1) - 1: all ones in a real-part lane
    for i in range(LIMBS):
        e.op("v_add_u32_dpp v%d, v%d, v%d %s" % (U[i], A[i], A[i], QUAD_IM))          # own + a1
        e.op("v_and_b32_dpp v%d, v%d, v%d %s" % (V[i], A[i], t_mask, QUAD_IM))        # a1 & re_mask (the DPP operand is always the argument a)
        e.op("v_sub_u32_dpp v%d, v%d, v%d %s" % (V[i], A[i], V[i], QUAD_RE))          # a0 - (a1 & re_mask)

    def terms(k):
        return [(U[i], "v%d" % V[k - i]) for i in range(LIMBS) if 0 <= k - i < LIMBS]
    e.columns(terms)
    return e.lines


def function(name, body):
    out = ["\t.text", "\t.p2align 8", "\t.type %s,@function" % name, "%s:" % name]
    out += body
    out += ["  s_setpc_b64 s[30:31]", ".Lend_%s:" % name, "\t.size %s, .Lend_%s-%s" % (name, name, name)]
    return out


def header():
    """bn254_leaf_asm.h: the body of fp_pair_mul_impl as ONE inline-asm statement — operands %0..%8 = a (in) / result (out), %9..%17 = b;
    every temporary is a fixed caller-saved register named in the clobber list, so the function keeps the C calling convention and the
    compiler's own bookkeeping of what a call destroys"""
    body = pair_mul(4, ["%%%d" % i for i in range(LIMBS)], ["%%%d" % (LIMBS + i) for i in range(LIMBS)], lazy=True)
    temps = AP + X + Y + M + [ACC, ACC + 1]
    print("// GENERATED by gen_leaf_asm.py header — do not edit.  The dual product of the lane-pair layout (BN_MONT_DUAL_BODY with the role")
    print("// prologue of fp_pair_mul_impl) as gfx950 assembly with a fixed register map: the same instructions the compiler emits, with the")
    print("// unused carry-out of v_mad_i64_i32 rotating over four SGPR pairs (profiles/r04_h_leaf_variants_microbench.jsonl: 681-716 -> 615 ns")
    print("// per product and SIMD with two waves; the squaring leaf gains nothing and stays compiled).")
    print("#pragma once")
    print("#define BN_LEAF_PAIR_MUL_TEXT \\")
    for ln in body:
        print('  "%s\\n" \\' % ln.strip())
    print('  ""')
    print("#define BN_LEAF_PAIR_MUL_CLOBBERS " + ", ".join('"v%d"' % r for r in temps) + ", " + ", ".join('"s%d"' % r for r in range(4, 24)))


def bench():
    print("// GENERATED by gen_leaf_asm.py bench — leaf bodies in a timing loop: does rotating the carry-out SGPR of v_mad_i64_i32 pay inside the real product?")
    print("#include <hip/hip_runtime.h>\n#include <cstdio>\n#include <vector>")
    clob = ", ".join('"v%d"' % r for r in range(4, 72) if r not in (40, 41, 42, 43, 44, 45, 46, 47, 56, 57, 58, 59, 60, 61, 62, 63)) + ", " + \
        ", ".join('"s%d"' % r for r in range(4, 30)) + ', "vcc"'
    variants = tuple(("mul_%s_%d_sinks" % ("lazy_fetch" if lz else "top_fetch", k), pair_mul(k, lazy=lz)) for lz in (False, True) for k in (1, 2, 4)) + \
        tuple(("sqr_%d_sinks" % k, pair_sqr(k)) for k in (1, 4))
    for name, body in variants:
        print("__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) k_%s(int* p, int iters) {" % name)
        print("  extern __shared__ int lds[];")
        print("  int a0 = p[threadIdx.x], a1 = p[threadIdx.x + 256], a2 = lds[threadIdx.x & 15], a3 = threadIdx.x;")
        print("  for (int it = 0; it < iters; ++it) {")
        print('    asm volatile("v_mov_b32 v0, %0\\n v_mov_b32 v1, %1\\n v_mov_b32 v2, %2\\n v_mov_b32 v3, %3\\n"')
        for ln in body:
            print('                 "%s\\n"' % ln.strip())
        print('                 "v_mov_b32 %0, v0\\n v_mov_b32 %1, v1\\n v_mov_b32 %2, v2\\n v_mov_b32 %3, v3\\n"')
        print('                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "v0", "v1", "v2", "v3", %s);' % clob)
        print("  }")
        print("  p[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3;\n}")
    print("""int main() {
  int* d; hipMalloc(&d, 4 * 256 * 2048); hipMemset(d, 1, 4 * 256 * 2048);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4000, blocks = 512;                      // 256 CUs x 2 workgroups of 4 waves: two waves per SIMD (56 KB of LDS each)
  const size_t lds = 56 * 1024;""")
    for name, body in variants:
        nmac = sum(1 for ln in body if "v_mad_i64_i32" in ln)
        print("""  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0); k_%s<<<blocks, 256, lds>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep) printf("{\\"variant\\": \\"%s\\", \\"ms\\": %%.3f, \\"ns_per_leaf_per_simd\\": %%.1f, \\"multiply_adds\\": %d, \\"instructions\\": %d}\\n", ms, ms * 1e6 / iters / 2.0);
  }""" % (name, name, nmac, len(body)))
    print("  return 0;\n}")


if __name__ == "__main__":
    {"header": header, "bench": bench}[sys.argv[1]]()
