"""Slow, independent big-integer model of the sedaprotocol/bn254 verify path.

TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may import anything under oracle/.  The product path (bn254_amd/, libbn254hip.so) never does.

This is the *second* CPU model: deliberately naive and shaped differently from both the C
oracle (oracle/bn254_oracle.c) and the HIP kernels so that a shared mistake is unlikely:
  * Python integers, no Montgomery form, no limbs;
  * affine curve arithmetic with modular inverses;
  * Fq12 as a degree-6 polynomial ring  Fq2[w]/(w^6 - xi)  (not a 2-3-2 tower);
  * plain binary (not signed-digit) Miller loop;
  * final exponentiation as one naive pow(f, (q^12-1)/r).

Parity status: PINNED.  The arithmetic lives in the third-party crate `zeropool-bn 0.5.11`
(`/root/reference/Cargo.toml:24`), which is not vendored, so this file restates the published
alt_bn128 / EIP-196/197 definitions and is anchored on every known-answer vector the
reference's own tests hold (tests/golden/reference_kats.json, see tests/test_oracle_model.py).

Wrapper semantics follow the reference line by line where cited.
"""
import hashlib

# --- curve constants --------------------------------------------
U = 4965661367192848881
Q = 36 * U**4 + 36 * U**3 + 24 * U**2 + 6 * U + 1
R = 36 * U**4 + 36 * U**3 + 18 * U**2 + 6 * U + 1
assert Q == 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
assert R == 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
B1 = 3
XI = (9, 1)  # xi = 9 + i
ATE_LOOP = 6 * U + 2
G1_GEN = (1, 2)
G2_GEN = (
    (10857046999023057135944570762232829481370756359578518086990519993285655852781,
     11559732032986387107991004021392285783925812861821192530917403151452391805634),
    (8495653923123431417604973247489272438418190587263600148770280649306958101930,
     4082367875863433681332203403145435568316851327593401208105741076214120093531),
)
# /root/reference/src/hash.rs:11-14
LAST_MULTIPLE_OF_FQ_MODULUS_LOWER_THAN_2_256 = 5 * Q

# status codes = 1 + index of the variant in /root/reference/src/error.rs:6-29 (0 = Ok)
OK = 0
ERR_HASH_TO_POINT = 1
ERR_INDEX_OUT_OF_BOUNDS = 2
ERR_INVALID_ENCODING = 3
ERR_INVALID_GROUP_POINT = 4
ERR_INVALID_LENGTH = 5
ERR_NOT_MEMBER = 6
ERR_TO_AFFINE = 7
ERR_POINT_IN_JACOBIAN = 8
ERR_VERIFICATION_FAILED = 9
ERR_SERIALIZATION = 10
ERR_HEX_DECODE = 11


class Bn254Error(Exception):
    def __init__(self, code):
        super().__init__("bn254 error %d" % code)
        self.code = code


# --- Fq2 = Fq[i]/(i^2+1), elements are tuples (re, im) ------------------------------------
def f2_add(a, b):
    return ((a[0] + b[0]) % Q, (a[1] + b[1]) % Q)


def f2_sub(a, b):
    return ((a[0] - b[0]) % Q, (a[1] - b[1]) % Q)


def f2_neg(a):
    return ((-a[0]) % Q, (-a[1]) % Q)


def f2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % Q, (a[0] * b[1] + a[1] * b[0]) % Q)


def f2_smul(a, k):
    return ((a[0] * k) % Q, (a[1] * k) % Q)


def f2_conj(a):
    return (a[0], (-a[1]) % Q)


def f2_inv(a):
    n = pow((a[0] * a[0] + a[1] * a[1]) % Q, -1, Q)
    return ((a[0] * n) % Q, (-a[1] * n) % Q)


def f2_pow(a, e):
    out = (1, 0)
    while e:
        if e & 1:
            out = f2_mul(out, a)
        a = f2_mul(a, a)
        e >>= 1
    return out


def f2_sqrt(a):
    """Square root in Fq2 for q = 3 mod 4 (Adj & Rodriguez-Henriquez, alg. 9); None if none."""
    if a == (0, 0):
        return (0, 0)
    a1 = f2_pow(a, (Q - 3) // 4)
    alpha = f2_mul(f2_mul(a1, a1), a)
    a0 = f2_mul(f2_pow(alpha, Q), alpha)
    if a0 == ((-1) % Q, 0):
        return None
    x0 = f2_mul(a1, a)
    if alpha == ((-1) % Q, 0):
        x = f2_mul((0, 1), x0)
    else:
        b = f2_pow(f2_add((1, 0), alpha), (Q - 1) // 2)
        x = f2_mul(b, x0)
    return x if f2_mul(x, x) == a else None


F2_ZERO = (0, 0)
F2_ONE = (1, 0)
B2 = f2_smul(f2_inv(XI), B1)  # twist coefficient 3/xi

# --- Fq12 = Fq2[w]/(w^6 - xi): list of 6 Fq2 coefficients, index = power of w -------------
F12_ONE = [F2_ONE] + [F2_ZERO] * 5


def f12_mul(a, b):
    t = [F2_ZERO] * 11
    for i in range(6):
        if a[i] == F2_ZERO:
            continue
        for j in range(6):
            t[i + j] = f2_add(t[i + j], f2_mul(a[i], b[j]))
    out = list(t[:6])
    for k in range(6, 11):
        out[k - 6] = f2_add(out[k - 6], f2_mul(t[k], XI))
    return out


def f12_pow(a, e):
    out = F12_ONE
    for bit in bin(e)[2:]:
        out = f12_mul(out, out)
        if bit == "1":
            out = f12_mul(out, a)
    return out


def f12_to_tower(c):
    """polynomial coefficients -> tower order used by the C oracle / HIP code:
    Fq12 = Fq6[w]/(w^2 - v), Fq6 = Fq2[v]/(v^3 - xi), v = w^2:
    ((a0,a1,a2),(b0,b1,b2)) with a_i = c[2i], b_i = c[2i+1]."""
    return ((c[0], c[2], c[4]), (c[1], c[3], c[5]))


def f12_to_bytes(c):
    """Canonical Gt serialisation of this build (the reference never serialises Gt,
    /root/reference/src/lib.rs:60-63): 12 big-endian 32-byte words in tower order
    a0.re a0.im a1.re a1.im a2.re a2.im b0.re ... b2.im  (384 bytes)."""
    (a, b) = f12_to_tower(c)
    out = b""
    for f2 in list(a) + list(b):
        out += f2[0].to_bytes(32, "big") + f2[1].to_bytes(32, "big")
    return out


# --- G1: y^2 = x^3 + 3 over Fq; None = identity -------------------------------------------
def g1_on_curve(p):
    if p is None:
        return True
    x, y = p
    return (y * y - x * x * x - B1) % Q == 0


def g1_neg(p):
    return None if p is None else (p[0], (-p[1]) % Q)


def g1_add(p, s):
    if p is None:
        return s
    if s is None:
        return p
    if p[0] == s[0]:
        if (p[1] + s[1]) % Q == 0:
            return None
        lam = (3 * p[0] * p[0]) * pow(2 * p[1], -1, Q) % Q
    else:
        lam = (s[1] - p[1]) * pow(s[0] - p[0], -1, Q) % Q
    x3 = (lam * lam - p[0] - s[0]) % Q
    return (x3, (lam * (p[0] - x3) - p[1]) % Q)


def g1_mul(p, k):
    """k is used as-is (NOT reduced mod r): /root/reference/src/bn256.json:54-159 has
    scalars up to 2^256-1."""
    out = None
    while k:
        if k & 1:
            out = g1_add(out, p)
        p = g1_add(p, p)
        k >>= 1
    return out


# --- G2: y^2 = x^3 + 3/xi over Fq2 ---------------------------------------------------------
def g2_on_curve(p):
    if p is None:
        return True
    x, y = p
    return f2_sub(f2_mul(y, y), f2_add(f2_mul(f2_mul(x, x), x), B2)) == F2_ZERO


def g2_neg(p):
    return None if p is None else (p[0], f2_neg(p[1]))


def g2_add(p, s):
    if p is None:
        return s
    if s is None:
        return p
    if p[0] == s[0]:
        if f2_add(p[1], s[1]) == F2_ZERO:
            return None
        lam = f2_mul(f2_smul(f2_mul(p[0], p[0]), 3), f2_inv(f2_smul(p[1], 2)))
    else:
        lam = f2_mul(f2_sub(s[1], p[1]), f2_inv(f2_sub(s[0], p[0])))
    x3 = f2_sub(f2_sub(f2_mul(lam, lam), p[0]), s[0])
    return (x3, f2_sub(f2_mul(lam, f2_sub(p[0], x3)), p[1]))


def g2_mul(p, k):
    out = None
    while k:
        if k & 1:
            out = g2_add(out, p)
        p = g2_add(p, p)
        k >>= 1
    return out


def g2_in_subgroup(p):
    return g2_mul(p, R) is None


# --- pairing --------------------------------------------------------------------------------
def _line(t, s, p):
    """Line through twist points t,s (t==s: tangent) evaluated at P in G1, as a sparse Fq12
    polynomial.  Untwist (x',y') -> (x' w^2, y' w^3); slope on E is lam' * w, so
    l(P) = yP - lam' xP w + (lam' xT' - yT') w^3.  A vertical line (t == -s) evaluates to
    xP - xT' w^2."""
    xp, yp = p
    if t[0] == s[0] and f2_add(t[1], s[1]) == F2_ZERO:
        return [(xp, 0), F2_ZERO, f2_neg(t[0]), F2_ZERO, F2_ZERO, F2_ZERO]
    if t == s:
        lam = f2_mul(f2_smul(f2_mul(t[0], t[0]), 3), f2_inv(f2_smul(t[1], 2)))
    else:
        lam = f2_mul(f2_sub(s[1], t[1]), f2_inv(f2_sub(s[0], t[0])))
    return [(yp, 0), f2_neg(f2_smul(lam, xp)), F2_ZERO, f2_sub(f2_mul(lam, t[0]), t[1]), F2_ZERO, F2_ZERO]


def _frob_twist(p, power):
    """q^power Frobenius of an untwisted point, in twist coordinates."""
    x, y = p
    for _ in range(power):
        x = f2_mul(f2_conj(x), f2_pow(XI, (Q - 1) // 3))
        y = f2_mul(f2_conj(y), f2_pow(XI, (Q - 1) // 2))
    return (x, y)


def miller_loop(p, qt):
    """f_{6u+2,Q}(P) * l_{[6u+2]Q, pi(Q)}(P) * l_{[6u+2]Q+pi(Q), -pi^2(Q)}(P), plain binary."""
    if p is None or qt is None:
        return F12_ONE
    f = F12_ONE
    t = qt
    for bit in bin(ATE_LOOP)[3:]:
        f = f12_mul(f12_mul(f, f), _line(t, t, p))
        t = g2_add(t, t)
        if bit == "1":
            f = f12_mul(f, _line(t, qt, p))
            t = g2_add(t, qt)
    q1 = _frob_twist(qt, 1)
    q2 = g2_neg(_frob_twist(qt, 2))
    f = f12_mul(f, _line(t, q1, p))
    t = g2_add(t, q1)
    f = f12_mul(f, _line(t, q2, p))
    return f


FINAL_EXP = (Q**12 - 1) // R


def final_exponentiation(f):
    return f12_pow(f, FINAL_EXP)


def pairing(p, qt):
    """Canonical reduced optimal-ate pairing e(P,Q) = miller(P,Q)^((q^12-1)/r)."""
    return final_exponentiation(miller_loop(p, qt))


def pairing_batch(pairs):
    """bn::pairing_batch as called at /root/reference/src/ecdsa.rs:57,86: product of pairings,
    pairs containing the identity contribute 1."""
    f = F12_ONE
    for (p, qt) in pairs:
        f = f12_mul(f, miller_loop(p, qt))
    return final_exponentiation(f)


# --- hash-to-G1, /root/reference/src/hash.rs:29-63 -------------------------------------------
def mod_u256(num, modulus):
    """/root/reference/src/utils.rs:27-37 (strict '>': a value equal to the modulus stays)."""
    reduced = num
    while reduced > modulus:
        reduced -= modulus
    return reduced


def fq_sqrt(a):
    y = pow(a, (Q + 1) // 4, Q)
    return y if (y * y) % Q == a % Q else None


def g1_from_compressed(data):
    """bn::G1::from_compressed as used at /root/reference/src/utils.rs:60 and types.rs:234."""
    if len(data) != 33:
        raise Bn254Error(ERR_INVALID_ENCODING)
    sign = data[0]
    x = int.from_bytes(data[1:], "big")
    if x >= Q:
        raise Bn254Error(ERR_NOT_MEMBER)
    y = fq_sqrt((x * x * x + B1) % Q)
    if y is None:
        raise Bn254Error(ERR_NOT_MEMBER)
    if sign == 2:
        if y & 1:
            y = Q - y
    elif sign == 3:
        if not (y & 1):
            y = Q - y
    else:
        raise Bn254Error(ERR_INVALID_ENCODING)
    return (x, y)


def hash_to_try_and_increment_ex(message):
    """returns (point, tries).  /root/reference/src/hash.rs:29-63."""
    v = bytearray(message) + b"\x00"                                  # :33
    for ctr in range(255):                                            # :40  (0..255 = 0..=254)
        v[-1] = ctr                                                   # :41
        h = int.from_bytes(hashlib.sha256(bytes(v)).digest(), "big")  # :42-44
        if h >= LAST_MULTIPLE_OF_FQ_MODULUS_LOWER_THAN_2_256:         # :49-51
            continue
        x = mod_u256(h, Q)                                            # :53
        try:
            return g1_from_compressed(b"\x02" + x.to_bytes(32, "big")), ctr + 1   # :54-58, utils.rs:56-63
        except Bn254Error:
            continue
    raise Bn254Error(ERR_HASH_TO_POINT)                               # :62


def hash_to_try_and_increment(message):
    return hash_to_try_and_increment_ex(message)[0]


# --- byte formats, /root/reference/src/utils.rs ----------------------------------------------
def g1_to_compressed(p):
    if p is None:
        raise Bn254Error(ERR_POINT_IN_JACOBIAN)       # utils.rs:86
    return bytes([3 if p[1] & 1 else 2]) + p[0].to_bytes(32, "big")


def g1_to_uncompressed(p):
    if p is None:
        raise Bn254Error(ERR_POINT_IN_JACOBIAN)       # utils.rs:184
    return p[0].to_bytes(32, "big") + p[1].to_bytes(32, "big")


def g1_from_uncompressed(data):
    if len(data) != 64:
        raise Bn254Error(ERR_INVALID_LENGTH)          # utils.rs:120
    x = int.from_bytes(data[:32], "big")
    y = int.from_bytes(data[32:], "big")
    if x >= Q or y >= Q:
        raise Bn254Error(ERR_NOT_MEMBER)              # Fq::from_slice -> FieldError::NotMember
    if not g1_on_curve((x, y)):
        raise Bn254Error(ERR_INVALID_GROUP_POINT)     # AffineG1::new -> GroupError
    return (x, y)


def _u512(c):
    return c[1] * Q + c[0]                            # utils.rs:40-45


def g2_to_compressed(p):
    if p is None:
        raise Bn254Error(ERR_POINT_IN_JACOBIAN)       # utils.rs:133
    x, y = p
    sign = 0x0B if _u512(y) > _u512(f2_neg(y)) else 0x0A   # utils.rs:142
    return bytes([sign]) + _u512(x).to_bytes(64, "big")


def g2_to_uncompressed(p):
    if p is None:
        raise Bn254Error(ERR_POINT_IN_JACOBIAN)       # utils.rs:163
    x, y = p
    return b"".join(v.to_bytes(32, "big") for v in (x[0], x[1], y[0], y[1]))


def g2_from_uncompressed(data, subgroup_check=True):
    if len(data) != 128:
        raise Bn254Error(ERR_INVALID_LENGTH)          # utils.rs:108
    w = [int.from_bytes(data[i:i + 32], "big") for i in range(0, 128, 32)]
    if any(v >= Q for v in w):
        raise Bn254Error(ERR_NOT_MEMBER)
    p = ((w[0], w[1]), (w[2], w[3]))
    if not g2_on_curve(p) or (subgroup_check and not g2_in_subgroup(p)):
        raise Bn254Error(ERR_INVALID_GROUP_POINT)     # AffineG2::new -> GroupError
    return p


def g2_from_compressed(data):
    """bn::G2::from_compressed as used at /root/reference/src/types.rs:92."""
    if len(data) != 65:
        raise Bn254Error(ERR_INVALID_ENCODING)
    sign = data[0]
    v = int.from_bytes(data[1:], "big")
    c1, c0 = divmod(v, Q)
    if c1 >= Q:
        # unpinned (no reference vector; the dependency is not vendored): upstream's Fq2::from_slice is recalled as
        # divrem(..).0.ok_or(FieldError::NotMember) -> Error::NotMemberError (/root/reference/src/error.rs:44-51);
        # InvalidU512Encoding is the wrong-LENGTH fault of U512::from_slice, unreachable behind the 65-byte check
        raise Bn254Error(ERR_NOT_MEMBER)
    x = (c0, c1)
    y = f2_sqrt(f2_add(f2_mul(f2_mul(x, x), x), B2))
    if y is None:
        raise Bn254Error(ERR_NOT_MEMBER)
    y_gt = _u512(y) > _u512(f2_neg(y))
    if sign == 0x0A:
        y = f2_neg(y) if y_gt else y
    elif sign == 0x0B:
        y = y if y_gt else f2_neg(y)
    else:
        raise Bn254Error(ERR_INVALID_ENCODING)
    p = (x, y)
    if not g2_on_curve(p) or not g2_in_subgroup(p):
        raise Bn254Error(ERR_NOT_MEMBER)
    return p


def private_key_from_bytes(data):
    """Fr::from_slice: length must be 32; the value is reduced mod r
    (/root/reference/src/types.rs:36-38, examples/bn254.rs:7-12 use keys > r)."""
    if len(data) != 32:
        raise Bn254Error(ERR_INVALID_LENGTH)
    return int.from_bytes(data, "big") % R


# --- scheme, /root/reference/src/ecdsa.rs ----------------------------------------------------
def sign(message, sk):
    return g1_mul(hash_to_try_and_increment(message), sk)             # ecdsa.rs:26-35


def public_key(sk):
    return g2_mul(G2_GEN, sk)                                         # types.rs:85-87


def public_key_g1(sk):
    return g1_mul(G1_GEN, sk)                                         # types.rs:155-157


def verify_status(message, sig, pk):
    """ecdsa.rs:49-64 -> status code (0 Ok, 9 VerificationFailed, 1 HashToPointError)."""
    try:
        h = hash_to_try_and_increment(message)                        # :53
    except Bn254Error as e:
        return e.code
    gt = pairing_batch([(h, pk), (sig, g2_neg(G2_GEN))])              # :54-57
    return OK if gt == F12_ONE else ERR_VERIFICATION_FAILED           # :59-63


def check_public_keys_status(pk_g2, pk_g1):
    """ecdsa.rs:78-93."""
    gt = pairing_batch([(G1_GEN, pk_g2), (pk_g1, g2_neg(G2_GEN))])
    return OK if gt == F12_ONE else ERR_VERIFICATION_FAILED
