#!/usr/bin/env python3
"""Generate bn254_constants.h: every numeric table the HIP kernels need.  Field elements are
9 little-endian BALANCED limbs of 29 bits (digits in [-2^28, 2^28), the top limb absorbs; stored in
int32) in Montgomery form with R = 2^261; plain 256-bit integers (moduli multiples, exponents, group
order) stay 8 x 32-bit words.

Self-contained (plain Python integers; imports nothing from oracle/): derived from the curve
definition in SURVEY.md Appendix A.1 only — u, the polynomials q(u), r(u), xi = 9+i, the
EIP-197 G2 generator.

The line table for the constant second pair of every verify, Q = -G2::one()
(/root/reference/src/ecdsa.rs:56), is produced with exactly the projective doubling/addition
formulas the kernels use for a variable Q (pairing.h: dbl_step/add_step), so a verify through
the table and a generic 2-pair Miller loop give the same un-exponentiated f.

Run:  python bn254_amd/csrc/gen_constants.py   (rewrites bn254_constants.h next to it)
"""
import os

U = 4965661367192848881
Q = 36 * U**4 + 36 * U**3 + 24 * U**2 + 6 * U + 1
R_ORDER = 36 * U**4 + 36 * U**3 + 18 * U**2 + 6 * U + 1
LIMB_BITS = 29
N_LIMBS = 9
MONT_R = 1 << (LIMB_BITS * N_LIMBS)
XI = (9, 1)
G2X = (10857046999023057135944570762232829481370756359578518086990519993285655852781,
       11559732032986387107991004021392285783925812861821192530917403151452391805634)
G2Y = (8495653923123431417604973247489272438418190587263600148770280649306958101930,
       4082367875863433681332203403145435568316851327593401208105741076214120093531)


def add(a, b): return ((a[0] + b[0]) % Q, (a[1] + b[1]) % Q)
def sub(a, b): return ((a[0] - b[0]) % Q, (a[1] - b[1]) % Q)
def neg(a): return ((-a[0]) % Q, (-a[1]) % Q)
def mul(a, b): return ((a[0] * b[0] - a[1] * b[1]) % Q, (a[0] * b[1] + a[1] * b[0]) % Q)
def sqr(a): return mul(a, a)
def smul(a, k): return ((a[0] * k) % Q, (a[1] * k) % Q)
def conj(a): return (a[0], (-a[1]) % Q)


def inv(a):
    n = pow((a[0] * a[0] + a[1] * a[1]) % Q, -1, Q)
    return ((a[0] * n) % Q, (-a[1] * n) % Q)


def fpow(a, e):
    out = (1, 0)
    while e:
        if e & 1:
            out = mul(out, a)
        a = mul(a, a)
        e >>= 1
    return out


# ---- final exponentiation as an ACCUMULATOR-MACHINE program (bn254_pairing.h: fe_machine) ---------------------------------
# One Fq12 accumulator (an LDS slot in the kernels) and a small file of Fq12 slots in the lane's private segment; every
# Fq12 routine then has ONE inlined call site in the kernel (a switch inside the interpreter loop) instead of ~26 out-of-loop
# calls that each save and restore up to 112 VGPRs.  Instructions are (opcode, slot) byte pairs:
FE_END, FE_LOAD, FE_STORE, FE_CSQR, FE_MUL, FE_CONJ, FE_FROB, FE_INV = range(8)
FE_OPNAMES = ["END", "LOAD", "STORE", "CSQR", "MUL", "CONJ", "FROB", "INV"]


class FeProgram:
    """builds a program and, alongside, the EXPONENT every value carries: after the easy part all values are powers of
    g = f^((q^6-1)(q^2+1)), an element of the cyclotomic subgroup of order Phi12(q) = q^4 - q^2 + 1, where conjugation
    is inversion and Frobenius is the q-th power — so a program is correct iff the exponent left in the accumulator is the
    wanted one modulo Phi12(q).  (The easy part itself is checked symbolically: see fe_easy.)"""

    def __init__(self):
        self.code = []
        self.slots = {}
        self.phi = Q**4 - Q**2 + 1
        self.acc = None               # exponent of the accumulator (None before the easy part)
        self.exp = {}                 # slot -> exponent

    def slot(self, name):
        return self.slots.setdefault(name, len(self.slots))

    def emit(self, op, name=None, arg=0):
        self.code.append((op, self.slot(name) if name is not None else arg))

    def load(self, n): self.emit(FE_LOAD, n); self.acc = self.exp[n]
    def store(self, n): self.emit(FE_STORE, n); self.exp[n] = self.acc
    def csqr(self): self.emit(FE_CSQR); self.acc = self.acc * 2 % self.phi
    def mul(self, n): self.emit(FE_MUL, n); self.acc = (self.acc + self.exp[n]) % self.phi
    def conj(self): self.emit(FE_CONJ); self.acc = -self.acc % self.phi
    def frob(self, k): self.emit(FE_FROB, None, k); self.acc = self.acc * pow(Q, k, self.phi) % self.phi

    def mulc(self, n):
        """acc * conj(slot) = conj(conj(acc) * slot): the conjugations run on the accumulator (27 negations in LDS) so that
        the multiplication keeps its one site and reads its second operand straight from the slot"""
        self.conj(); self.mul(n); self.conj()

    def easy(self):
        """f^((q^6-1)(q^2+1)) with one inversion: f1 = conj(f) * f^-1 = f^(q^6-1); then frob2(f1) * f1"""
        self.emit(FE_STORE, "F")      # fin
        self.emit(FE_INV)
        self.emit(FE_CONJ); self.emit(FE_MUL, "F"); self.emit(FE_CONJ)   # conj(conj(inv) * fin) = inv * conj(fin)  [conj is a ring automorphism]
        self.emit(FE_STORE, "F")      # f1
        self.emit(FE_FROB, None, 2)
        self.emit(FE_MUL, "F")
        self.acc = 1                  # g

    def pow_u(self, w4):
        """acc <- acc^u over the signed digits {1, 15, 19}: table a (T0), a^15 (T1), a^19 (T2)"""
        self.store("T0")
        self.csqr(); self.csqr(); self.store("T2")          # a^4
        self.csqr(); self.csqr()                              # a^16
        self.mulc("T0"); self.store("T1")                    # a^15
        self.mul("T2"); self.store("T2")                     # a^19
        tab = {1: "T0", 15: "T1", 19: "T2"}
        base = self.exp["T0"]
        assert self.exp["T1"] == 15 * base % self.phi and self.exp["T2"] == 19 * base % self.phi
        self.load(tab[w4[0]])
        for d in w4[1:]:
            self.csqr()
            if d > 0:
                self.mul(tab[d])
            elif d < 0:
                self.mulc(tab[-d])
        assert self.acc == base * U % self.phi

    def finish(self, want):
        assert self.acc == want % self.phi, "final exponentiation program computes the wrong power"
        self.code.append((FE_END, 0))
        assert len(self.slots) <= 16 and all(0 <= a < 256 for _, a in self.code)
        return self.code


def fe_program_check(w4):
    """the == one test of verify: Fuentes-Castaneda et al. hard part, g^(m h) with h = Phi12(q)/r, m = 2u(6u^2+3u+1)
    (10 multiplications, 3 squarings, 3 Frobenius maps beside the three exponentiations by u)"""
    P = FeProgram()
    P.easy()
    P.store("F")
    P.pow_u(w4); P.conj(); P.csqr(); P.store("Y1")          # g^-2u
    P.csqr(); P.mul("Y1"); P.store("Y3")                    # g^-6u
    P.pow_u(w4); P.conj(); P.store("Y4")                    # g^(6u^2)
    P.csqr(); P.pow_u(w4)                                   # g^(12u^3)
    P.mul("Y4"); P.mulc("Y3"); P.store("Y8")                # g^(12u^3+6u^2+6u)
    P.mul("Y1"); P.store("Y9")                              # g^(12u^3+6u^2+4u)
    P.load("Y8"); P.mul("Y4"); P.mul("F"); P.store("Y11")   # g^(12u^3+12u^2+6u+1)
    P.load("Y9"); P.frob(1); P.mul("Y11"); P.store("Y11")
    P.load("Y8"); P.frob(2); P.mul("Y11"); P.store("Y11")
    P.load("Y9"); P.mulc("F"); P.frob(3); P.mul("Y11")
    lam = (12 * U**3 + 12 * U**2 + 6 * U + 1) + (12 * U**3 + 6 * U**2 + 4 * U) * Q + (12 * U**3 + 6 * U**2 + 6 * U) * Q**2 + \
          (12 * U**3 + 6 * U**2 + 4 * U - 1) * Q**3
    m = 2 * U * (6 * U**2 + 3 * U + 1)
    h = (Q**4 - Q**2 + 1) // R_ORDER
    assert (Q**4 - Q**2 + 1) % R_ORDER == 0 and lam % P.phi == m * h % P.phi and 0 < m < R_ORDER and R_ORDER % m != 0
    return P.finish(m * h), P.slots


def fe_program_exact(w4):
    """the canonical Gt of the pairing API: the exact hard part Phi12(q)/r by the vectorial addition chain
    y0 y1^2 y2^6 y3^12 y4^18 y5^30 y6^36 (13 multiplications, 4 squarings, 6 Frobenius maps)"""
    P = FeProgram()
    P.easy()
    P.store("F"); P.pow_u(w4); P.store("FU"); P.pow_u(w4); P.store("FU2"); P.pow_u(w4); P.store("FU3")
    P.load("FU2"); P.frob(1); P.mul("FU"); P.conj(); P.store("Y4")                  # y4 = conj(frob(fu2) fu)
    P.load("FU3"); P.frob(1); P.mul("FU3"); P.conj()                                  # y6 = conj(frob(fu3) fu3)
    P.csqr(); P.mul("Y4"); P.mulc("FU2"); P.store("A")                              # t0 = y6^2 y4 y5,  y5 = conj(fu2)
    P.load("FU"); P.frob(1); P.conj(); P.mulc("FU2"); P.mul("A"); P.store("B")     # t1 = y3 y5 t0,  y3 = conj(frob(fu))
    P.load("FU2"); P.frob(2); P.mul("A"); P.store("A")                              # t0 = t0 y2,  y2 = frob2(fu2)
    P.load("B"); P.csqr(); P.mul("A"); P.csqr(); P.store("B")                      # t1 = (t1^2 t0)^2
    P.mulc("F"); P.csqr(); P.store("A")                                              # t0 = (t1 y1)^2,  y1 = conj(f)
    P.load("F"); P.frob(1); P.store("Y4"); P.load("F"); P.frob(2); P.mul("Y4"); P.store("Y4"); P.load("F"); P.frob(3); P.mul("Y4")   # y0
    P.mul("B"); P.mul("A")
    return P.finish((Q**4 - Q**2 + 1) // R_ORDER), P.slots


TWIST_B = smul(inv(XI), 3)
TWIST_3B = smul(TWIST_B, 3)


def naf_digits(s):
    """signed-digit (NAF) expansion, most-significant first, without the leading 1"""
    d = []
    while s:
        if s & 1:
            k = 2 - (s & 3)
            s -= k
        else:
            k = 0
        d.append(k)
        s >>= 1
    assert d[-1] == 1
    if d[-3:] == [-1, 0, 1]:      # top digits 1,0,-1 -> 1,1 (same value, one doubling fewer)
        d = d[:-3] + [1, 1]
    return d[-2::-1]


def pow_schedule(e, w):
    """left-to-right sliding window over the odd powers below 2^w: list of (squarings, multiplier)"""
    bits = bin(e)[2:]
    i, nsq, ops = 0, 0, []
    while i < len(bits):
        if bits[i] == "0":
            nsq += 1
            i += 1
            continue
        j = min(i + w, len(bits))
        while bits[j - 1] == "0":
            j -= 1
        ops.append((0 if not ops else nsq + (j - i), int(bits[i:j], 2)))
        nsq, i = 0, j
    if nsq:
        ops.append((nsq, 0))
    acc = 0
    for n, v in ops:
        acc = (acc << n) + v
    assert acc == e and all(v < (1 << w) and (v & 1 or v == 0) for _, v in ops)
    return ops


def naf_plain(s):
    """canonical NAF, most-significant first, without the leading 1"""
    d = []
    while s:
        if s & 1:
            k = 2 - (s & 3)
            s -= k
        else:
            k = 0
        d.append(k)
        s >>= 1
    assert d[-1] == 1
    v = 1
    for x in d[-2::-1]:
        v = 2 * v + x
    assert v == U
    return d[-2::-1]


def dbl_step(T):
    """homogeneous projective doubling on the twist + line coefficients (pairing.h: dbl_step).
    line = c0*yP + c1*xP*w + c2*w^3 with c0 = 2YZ, c1 = -3X^2, c2 = Y^2 - 3b'Z^2"""
    X, Y, Z = T
    xy = mul(X, Y); b = sqr(Y); c = sqr(Z)
    e = mul(c, TWIST_3B); f = smul(e, 3)
    h = sub(sub(sqr(add(Y, Z)), b), c)
    x2 = sqr(X)
    X3 = smul(mul(xy, sub(b, f)), 2)
    Y3 = sub(sqr(add(b, f)), smul(sqr(e), 12))
    Z3 = smul(mul(b, h), 4)
    return (X3, Y3, Z3), (h, neg(smul(x2, 3)), sub(b, e))


def add_step(T, Qa):
    """mixed addition T + Q (Q affine) + line coefficients (pairing.h: add_step):
    c0 = mu, c1 = -theta, c2 = theta*x2 - mu*y2"""
    X, Y, Z = T
    x2, y2 = Qa
    theta = sub(Y, mul(y2, Z)); mu = sub(X, mul(x2, Z))
    c = sqr(theta); d = sqr(mu); e = mul(mu, d)
    f = mul(Z, c); g = mul(X, d)
    h = sub(sub(add(e, f), g), g)
    X3 = mul(mu, h)
    Y3 = sub(mul(theta, sub(g, h)), mul(e, Y))
    Z3 = mul(Z, e)
    return (X3, Y3, Z3), (mu, neg(theta), sub(mul(theta, x2), mul(mu, y2)))


def words32(x):
    return [(x >> (32 * i)) & 0xFFFFFFFF for i in range(8)]


def limbs_balanced(x):
    """digits d_i in [-2^(W-1), 2^(W-1)) for i < N-1, the top limb takes the rest: x = sum d_i 2^(W i)"""
    out = []
    for _ in range(N_LIMBS - 1):
        d = x & ((1 << LIMB_BITS) - 1)
        if d >= 1 << (LIMB_BITS - 1):
            d -= 1 << LIMB_BITS
        out.append(d)
        x = (x - d) >> LIMB_BITS
    out.append(x)
    assert abs(x) < 1 << 31
    return out


def limbs_floor(x):
    """plain non-negative digits in [0, 2^W)"""
    return [(x >> (LIMB_BITS * i)) & ((1 << LIMB_BITS) - 1) for i in range(N_LIMBS)]


def mont(x):
    return (x * MONT_R) % Q


def c_u256(x):
    """plain 256-bit integer as 8 x u32"""
    return "{" + ", ".join("0x%08xu" % w for w in words32(x)) + "}"


def c_fp(x, m=True):
    """field element as balanced limbs (Montgomery form unless m=False)"""
    return "{" + ", ".join("%d" % w for w in limbs_balanced(mont(x) if m else x)) + "}"


def c_fp2(a):
    return "{" + c_fp(a[0]) + ", " + c_fp(a[1]) + "}"


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    naf = naf_digits(6 * U + 2)
    assert len(naf) == 64
    # fixed-Q line table for Q = -G2
    qn = (G2X, neg(G2Y))
    qpos = (G2X, G2Y)   # -(-G2)
    T = (qn[0], qn[1], (1, 0))
    lines = []
    for d in naf:
        T, ln = dbl_step(T); lines.append(ln)
        if d:
            T, ln = add_step(T, qn if d > 0 else qpos); lines.append(ln)
    g_x1 = fpow(XI, (Q - 1) // 3); g_y1 = fpow(XI, (Q - 1) // 2); g_x2 = fpow(XI, (Q * Q - 1) // 3)
    q1 = (mul(conj(qn[0]), g_x1), mul(conj(qn[1]), g_y1))
    q2 = (mul(qn[0], g_x2), qn[1])
    T, ln = add_step(T, q1); lines.append(ln)
    T, ln = add_step(T, q2); lines.append(ln)
    # scale every line by 1/c2 (an Fq2 factor, killed by the final exponentiation) so that c2 == 1:
    # the product of a variable line with a table line then takes 5 Fq2 products instead of 6
    for i, ln in enumerate(lines):
        assert ln[2] != (0, 0)
        k = fpow(ln[2], Q * Q - 2)
        lines[i] = (mul(ln[0], k), mul(ln[1], k), (1, 0))
        assert mul(ln[2], k) == (1, 0)

    frob = {}
    for j in (1, 2, 3):
        g = fpow(XI, (Q**j - 1) // 6)
        tab = [(1, 0)]
        for k in range(1, 6):
            tab.append(mul(tab[-1], g))
        frob[j] = tab
    assert all(t[1] == 0 for t in frob[2])

    o = []
    o.append("// GENERATED by bn254_amd/csrc/gen_constants.py — do not edit.")
    o.append("// BN254 constants.  Field elements: %d balanced limbs of %d bits (int32), Montgomery form, R = 2^%d." % (N_LIMBS, LIMB_BITS, N_LIMBS * LIMB_BITS))
    o.append("// Plain integers (U256): 8 x 32-bit words, little-endian.")
    o.append("#pragma once")
    o.append("")
    o.append("#define BN_LIMBS %d" % N_LIMBS)
    o.append("#define BN_W %d" % LIMB_BITS)
    o.append("#define BN_MASK 0x%xu" % ((1 << LIMB_BITS) - 1))
    o.append("#define BN_HALF 0x%x   /* 2^(W-1): digits are in [-BN_HALF, BN_HALF) */" % (1 << (LIMB_BITS - 1)))
    for i, w in enumerate(limbs_balanced(Q)):
        o.append("#define BN_QL%d (%d)" % (i, w))
    o.append("#define BN_QL_ARRAY {%s}" % ", ".join("BN_QL%d" % i for i in range(N_LIMBS)))
    o.append("#define BN_N0 0x%07xu   /* -q^-1 mod 2^%d */" % ((-pow(Q, -1, 1 << LIMB_BITS)) % (1 << LIMB_BITS), LIMB_BITS))
    top_unit = Q / float(1 << (LIMB_BITS * (N_LIMBS - 1)))           # q in units of the top limb
    o.append("#define BN_TOP_PER_Q %.1f   /* q / 2^%d: the top limb of a tight element of value v*q is ~ v * this */" % (top_unit, LIMB_BITS * (N_LIMBS - 1)))
    o.append("#define BN_R_OVER_Q %.3f   /* R / q: a Montgomery product shrinks |a||b| (in units of q^2) by this */" % (MONT_R / float(Q)))
    kmul = round((1 << 32) / top_unit)
    o.append("#define BN_WEAK_KMUL %d   /* round(2^32 / (q / 2^%d)): k = mulhi(top + BN_WEAK_HALF, KMUL) ~ round(value / q) */" % (kmul, LIMB_BITS * (N_LIMBS - 1)))
    o.append("#define BN_WEAK_HALF %d" % int(top_unit / 2))
    o.append("#define BN_U_LO 0x%08xu" % (U & 0xFFFFFFFF))
    o.append("#define BN_U_HI 0x%08xu" % (U >> 32))
    o.append("#define BN_N_FIXED_LINES %d" % len(lines))
    o.append("")
    o.append("BN_CONST uint32_t C_Q[8] = %s;            /* q, plain U256 */" % c_u256(Q))
    o.append("BN_CONST uint32_t C_QMULT[5][8] = {%s};   /* k*q, k = 1..5, plain U256 (5q = hash.rs:11-14) */" %
             ", ".join(c_u256(k * Q) for k in range(1, 6)))
    o.append("BN_CONST uint32_t C_ORDER_R[8] = %s;      /* group order r, plain U256 */" % c_u256(R_ORDER))
    o.append("BN_CONST uint32_t C_EXP_QM2[8] = %s;      /* q-2, plain U256 */" % c_u256(Q - 2))
    o.append("BN_CONST uint32_t C_EXP_QP1D4[8] = %s;    /* (q+1)/4, plain U256 */" % c_u256((Q + 1) // 4))
    o.append("BN_CONST uint32_t C_EXP_QM3D4[8] = %s;    /* (q-3)/4, plain U256 */" % c_u256((Q - 3) // 4))
    o.append("BN_CONST uint32_t C_EXP_QM1D2[8] = %s;    /* (q-1)/2, plain U256 */" % c_u256((Q - 1) // 2))
    # GLV endomorphism of G1: phi(x, y) = (beta x, y) = lambda (x, y), beta^3 = 1 mod q, lambda^2 + lambda + 1 = 0 mod r
    glv_beta = 0x59E26BCEA0D48BACD4F263F1ACDB5C4F5763473177FFFFFE
    glv_lambda = 0xB3C4D79D41A917585BFC41088D8DAAA78B17EA66B99C90DD
    assert pow(glv_beta, 3, Q) == 1 and glv_beta != 1 and (glv_lambda * glv_lambda + glv_lambda + 1) % R_ORDER == 0

    def g1_add(P, T):
        if P is None:
            return T
        if T is None:
            return P
        if P[0] == T[0]:
            if (P[1] + T[1]) % Q == 0:
                return None
            lam = 3 * P[0] * P[0] * pow(2 * P[1], Q - 2, Q) % Q
        else:
            lam = (T[1] - P[1]) * pow(T[0] - P[0], Q - 2, Q) % Q
        x3 = (lam * lam - P[0] - T[0]) % Q
        return (x3, (lam * (P[0] - x3) - P[1]) % Q)
    acc = None
    for bit in bin(glv_lambda)[2:]:
        acc = g1_add(acc, acc)
        if bit == "1":
            acc = g1_add(acc, (1, 2))
    assert acc == (glv_beta % Q, 2), "lambda * G1 != (beta * x, y)"
    o.append("BN_CONST int32_t C_GLV_BETA[BN_LIMBS] = %s;   /* phi(x, y) = (beta x, y) = lambda (x, y) on G1 */" % c_fp(glv_beta))
    o.append("BN_CONST uint32_t C_GLV_LAMBDA[8] = %s;       /* lambda (192 bits), plain U256 */" % c_u256(glv_lambda))
    # decomposition of a FULL scalar k in [0, r) (round 6: g1_mul_glv_full — sign and variable-base G1 multiplication): the lattice
    # {(x, y): x + y lambda = 0 mod r} has the reduced basis v1 = (a1, -b1n), v2 = (a2, a1) (extended Euclid on (r, lambda), det = r);
    # (k, 0) = beta1 v1 + beta2 v2 with beta1 = k a1 / r, beta2 = k b1n / r; c_i = floor(k g_i / 2^256) <= floor(beta_i) with
    # g1 = floor(2^256 a1 / r), g2 = floor(2^256 b1n / r); k1 = k - c1 a1 - c2 a2 in [0, a1 + a2) (< 2^128),
    # k2 = c1 b1n - c2 a1 in (-b1n, a1] (|k2| < 2^127): k = k1 + k2 lambda mod r
    glv_a1, glv_b1n, glv_a2 = 0x89d3256894d213e3, 0x6f4d8248eeb859fc8211bbeb7d4f1128, 0x6f4d8248eeb859fd0be4e1541221250b
    assert (glv_a1 - glv_b1n * glv_lambda) % R_ORDER == 0 and (glv_a2 + glv_a1 * glv_lambda) % R_ORDER == 0
    assert glv_a1 * glv_a1 + glv_a2 * glv_b1n == R_ORDER
    glv_g1, glv_g2 = (glv_a1 << 256) // R_ORDER, (glv_b1n << 256) // R_ORDER
    assert glv_g1 < 1 << 96 and glv_g2 < 1 << 160 and glv_a1 + glv_a2 < 1 << 128

    def c_words(x, n):
        return "{%s}" % ", ".join("0x%08xu" % ((x >> (32 * i)) & 0xFFFFFFFF) for i in range(n))
    o.append("BN_CONST uint32_t C_GLV_A1[2] = %s, C_GLV_B1N[4] = %s, C_GLV_A2[4] = %s;   /* reduced basis (a1, -b1n), (a2, a1) of the GLV lattice */" %
             (c_words(glv_a1, 2), c_words(glv_b1n, 4), c_words(glv_a2, 4)))
    o.append("BN_CONST uint32_t C_GLV_G1[3] = %s, C_GLV_G2[5] = %s;   /* floor(2^256 a1 / r), floor(2^256 b1n / r) */" % (c_words(glv_g1, 3), c_words(glv_g2, 5)))
    o.append("/* width-4 sliding-window schedules for the fixed exponents: {squarings, odd multiplier} steps, MSB first;")
    o.append("   the first step only selects its multiplier, a multiplier of 0 means squarings only */")
    for name, e in (("QM2", Q - 2), ("QP1D4", (Q + 1) // 4), ("QM3D4", (Q - 3) // 4), ("QM1D2", (Q - 1) // 2)):
        ops = pow_schedule(e, 4)
        o.append("#define BN_SCHED_%s_LEN %d" % (name, len(ops)))
        o.append("BN_CONST unsigned char C_SCHED_%s[BN_SCHED_%s_LEN][2] = {%s};  /* %d squarings, %d multiplications */" %
                 (name, name, ", ".join("{%d, %d}" % op for op in ops), sum(op[0] for op in ops), sum(1 for op in ops[1:] if op[1])))
    o.append("BN_CONST int32_t C_QL[BN_LIMBS] = %s;           /* q as balanced limbs (plain) */" % c_fp(Q, False))
    o.append("BN_CONST int32_t C_R2[BN_LIMBS] = %s;           /* R^2 mod q, plain limbs: to_mont(x) = mul(x, R2) */" % c_fp(MONT_R * MONT_R % Q, False))
    o.append("BN_CONST int32_t C_R3[BN_LIMBS] = %s;           /* R^3 mod q, plain limbs: mul(plain inverse of a Montgomery value, R3) = its Montgomery inverse */" % c_fp(pow(MONT_R, 3, Q), False))
    o.append("BN_CONST int32_t C_QF[BN_LIMBS] = {%s};         /* q as plain digits in [0, 2^%d) (fp_inv: division steps) */" % (", ".join("%d" % w for w in limbs_floor(Q)), LIMB_BITS))
    for i, w in enumerate(limbs_floor(Q)):
        o.append("#define C_QF_%d %d" % (i, w))
    o.append("#define BN_QINV 0x%07xu   /* q^-1 mod 2^%d */" % (pow(Q, -1, 1 << LIMB_BITS), LIMB_BITS))
    o.append("BN_CONST int32_t C_ONE[BN_LIMBS] = %s;          /* 1 (Montgomery) */" % c_fp(1))
    o.append("BN_CONST int32_t C_THREE[BN_LIMBS] = %s;        /* curve b = 3 */" % c_fp(3))
    o.append("BN_CONST int32_t C_TWIST_B[2][BN_LIMBS] = %s;   /* 3/xi */" % c_fp2(TWIST_B))
    o.append("BN_CONST int32_t C_TWIST_3B[2][BN_LIMBS] = %s;  /* 9/xi */" % c_fp2(TWIST_3B))
    o.append("BN_CONST int32_t C_XI_MONT[2][BN_LIMBS] = %s;  /* xi = 9 + i (w^6; the product of two table lines with c2 = 1) */" % c_fp2(XI))
    for j in (1, 2, 3):
        o.append("BN_CONST int32_t C_FROB%d[6][2][BN_LIMBS] = {%s};  /* xi^(k(q^%d-1)/6), k=0..5 */" % (j, ", ".join(c_fp2(t) for t in frob[j]), j))
    o.append("BN_CONST int32_t C_TW_FROB_X1[2][BN_LIMBS] = %s;" % c_fp2(g_x1))
    o.append("BN_CONST int32_t C_TW_FROB_Y1[2][BN_LIMBS] = %s;" % c_fp2(g_y1))
    o.append("BN_CONST int32_t C_TW_FROB_X2[2][BN_LIMBS] = %s;" % c_fp2(g_x2))
    o.append("BN_CONST int32_t C_G1_GEN[2][BN_LIMBS] = {%s, %s};" % (c_fp(1), c_fp(2)))
    o.append("BN_CONST int32_t C_G2_GEN[2][2][BN_LIMBS] = {%s, %s};" % (c_fp2(G2X), c_fp2(G2Y)))
    o.append("BN_CONST signed char C_ATE_NAF[64] = {%s};  /* digits of 6u+2 after the leading 1, MSB first */" % ", ".join(str(d) for d in naf))
    unaf = naf_plain(U)
    o.append("#define BN_U_NAF_LEN %d" % len(unaf))
    o.append("BN_CONST signed char C_U_NAF[BN_U_NAF_LEN] = {%s};  /* NAF digits of u after the leading 1, MSB first (weight %d) */" %
             (", ".join(str(d) for d in unaf), sum(1 for d in unaf if d)))
    # exponentiation by u in the cyclotomic subgroup (inverse = conjugate): signed digits from the set {1, 15, 19} found by a
    # search over small odd digit sets (cost = table multiplications + non-zero digits): 2 + 12 - 1 = 13 multiplications
    # against 3 + 14 - 1 = 16 for width-4 signed windows.  Table: a^2, a^4, a^8, a^16 (4 squarings), a^15 = a^16 / a,
    # a^19 = a^15 * a^4.  Digits by dynamic programming over u = sum d_i 2^i, d_i in {0, +-1, +-15, +-19}, fewest non-zero.
    import functools
    import sys
    sys.setrecursionlimit(10000)
    DSET = (1, 15, 19)

    @functools.lru_cache(None)
    def best(m):
        """(non-zero digits, digit list LSB first) for m"""
        if m == 0:
            return (0, ())
        if m % 2 == 0:
            c, ds = best(m // 2)
            return (c, (0,) + ds)
        cand = None
        for d in DSET:
            for sd in (d, -d):
                if m == sd:
                    r = (1, (sd,))
                elif abs((m - sd) // 2) < abs(m):
                    c, ds = best((m - sd) // 2)
                    r = (c + 1, (sd,) + ds)
                else:
                    continue
                if cand is None or r[0] < cand[0] or (r[0] == cand[0] and len(r[1]) < len(cand[1])):
                    cand = r
        return cand
    nz, digs = best(U)
    w4 = list(reversed(digs))
    while w4[0] == 0:
        w4.pop(0)
    assert sum(d << (len(w4) - 1 - i) for i, d in enumerate(w4)) == U and w4[0] > 0 and nz == 12
    o.append("#define BN_U_W4_LEN %d" % len(w4))
    o.append("BN_CONST signed char C_U_W4[BN_U_W4_LEN] = {%s};  /* signed digits of u from {0, +-1, +-15, +-19}, MSB first, %d non-zero */" %
             (", ".join(str(d) for d in w4), sum(1 for d in w4 if d)))
    for name, (code, slots) in (("CHECK", fe_program_check(w4)), ("EXACT", fe_program_exact(w4))):
        o.append("#define BN_FE_%s_LEN %d" % (name, len(code)))
        o.append("#define BN_FE_%s_SLOTS %d" % (name, len(slots)))
        o.append("BN_CONST unsigned char C_FE_%s[BN_FE_%s_LEN][2] = {%s};  /* accumulator-machine program of the final exponentiation (gen_constants.py: fe_program_%s; %s) */" %
                 (name, name, ", ".join("{%d, %d}" % ins for ins in code), name.lower(),
                  ", ".join("%d %s" % (sum(1 for op, _ in code if op == k), FE_OPNAMES[k]) for k in range(1, 8))))
    o.append("/* line coefficients (c0 -> *yP, c1 -> *xP, c2 == 1) for Q = -G2::one(), in order of use */")
    o.append("BN_CONST int32_t C_NEG_G2_LINES[BN_N_FIXED_LINES][3][2][BN_LIMBS] = {")
    for ln in lines:
        o.append("  {%s, %s, %s}," % (c_fp2(ln[0]), c_fp2(ln[1]), c_fp2(ln[2])))
    o.append("};")
    o.append("")
    path = os.path.join(here, "bn254_constants.h")
    with open(path, "w") as f:
        f.write("\n".join(o))
    print("wrote", path, "lines:", len(lines), "naf weight:", sum(1 for d in naf if d))


if __name__ == "__main__":
    main()
