#!/usr/bin/env python3
"""Generates tests/golden/adversarial_fe_vectors.json: inputs of the final exponentiation of ECDSA::verify
(/root/reference/src/ecdsa.rs:57-59) as LIMB vectors that sit at the edge of what the interval tracker's contract for a Miller value
allows (tests/hostsim: hp_miller_output_bounds — limbs 0..7 in [-2^28, 2^28], |top limb| <= 1.62 .. 1.65 M, |value| <= 0.511 .. 0.5215 q
depending on the coefficient; the vectors stay inside the smallest of them), i.e.
non-canonical Montgomery representatives with extreme balanced digits that no byte decoder produces, together with what the
independent big-integer model (oracle/bn254_model.py: one pow(f, (q^12 - 1) / r)) says the result is.

    python tests/golden/gen_adversarial_fe.py          # ~2 minutes of pure-Python field arithmetic

Two families:
  * "full": all 12 coefficients adversarial; the result is some Gt element != 1 (status 9); canonical Gt bytes recorded.
  * "one":  f = g^r * s with s in Fq6 chosen so that the six coefficients of f's c0 half ARE chosen adversarial limb vectors
            (s = t0 / (g^r).c0; elements of the subfield Fq6 die in the easy part, g^r in the hard part): the result is 1 (status 0)
            whatever the arithmetic does in between — a status-only kernel that slips anywhere answers 9.
Test infrastructure only; consumed by tests/test_bounds.py (host emulations) and tests/test_gpu_parity.py (every device layout).
"""
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import bn254_model as M  # noqa: E402

Q = M.Q if hasattr(M, "Q") else 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
R_ORDER = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
W, LIMBS = 29, 9
T = 1 << 28
MONT_R = 1 << (W * LIMBS)
MONT_RINV = pow(MONT_R, -1, Q)
VMAX = 0.511           # |value| / q: below the SMALLEST per-coefficient bound the tracker reports for a Miller output (0.5111 .. 0.5215)
TOPMAX = 1620000       # below the smallest per-coefficient top bound the tracker reports (1 621 006)


def value_of(limbs):
    return sum(d << (W * i) for i, d in enumerate(limbs))


def to_tight_limbs(v):
    """the unique tight representation of the integer v: digits 0..7 in [-2^28, 2^28), the top limb absorbs"""
    out = []
    for _ in range(LIMBS - 1):
        d = v & ((1 << W) - 1)
        if d >= T:
            d -= 1 << W
        out.append(d)
        v = (v - d) >> W
    out.append(v)
    return out


def adversarial_limbs(rnd, kind):
    hi, lo = T - 1, -T
    if kind == 0:
        d = [hi] * 8
    elif kind == 1:
        d = [lo] * 8
    elif kind == 2:
        d = [hi if i & 1 else lo for i in range(8)]
    elif kind == 3:
        d = [lo if i & 1 else hi for i in range(8)]
    elif kind == 4:
        d = [rnd.choice((hi, lo)) for _ in range(8)]
    elif kind == 5:
        d = [rnd.choice((T, lo, hi)) for _ in range(8)]          # +2^28 itself: inside the tracker's closed interval
    elif kind == 6:
        d = [rnd.choice((hi, lo, 0, 1, -1)) for _ in range(8)]
    else:
        d = [rnd.randrange(lo, hi + 1) for _ in range(8)]
    for _ in range(100):
        top = rnd.choice((TOPMAX, -TOPMAX, TOPMAX - rnd.randrange(1000), -TOPMAX + rnd.randrange(1000), 0, 1, -1, rnd.randrange(-TOPMAX, TOPMAX)))
        limbs = d + [top]
        if abs(value_of(limbs)) <= VMAX * Q:
            return limbs
    raise AssertionError("no top limb fits the value bound")


def field_of(limbs):
    """the field element a Montgomery limb vector stands for"""
    return value_of(limbs) * MONT_RINV % Q


def tower_to_poly(t12):
    """12 Fq in Gt order (a0.re a0.im a1.. b2.im) -> the model's polynomial basis (coefficient of w^k)"""
    f2 = [(t12[2 * k], t12[2 * k + 1]) for k in range(6)]
    a, b = f2[:3], f2[3:]
    return [a[0], b[0], a[1], b[1], a[2], b[2]]


def poly_to_tower(c):
    (a, b) = M.f12_to_tower(c)
    out = []
    for f2 in list(a) + list(b):
        out += [f2[0], f2[1]]
    return out


def main():
    rnd = random.Random(20261004)
    vectors = []
    # family "full"
    for k in range(20):
        limbs = [adversarial_limbs(rnd, (k + e) % 8 if k < 8 else rnd.randrange(8)) for e in range(12)]
        f = tower_to_poly([field_of(x) for x in limbs])
        g = M.final_exponentiation(f)
        vectors.append({"family": "full", "limbs": [d for x in limbs for d in x], "gt": M.f12_to_bytes(g).hex(), "status": 0 if g == M.F12_ONE else 9})
        print("full", k, vectors[-1]["status"], flush=True)
    # family "one"
    for k in range(12):
        while True:
            g = [(rnd.randrange(Q), rnd.randrange(Q)) for _ in range(6)]
            h = M.f12_pow(g, R_ORDER)
            h0 = [h[0], M.F2_ZERO, h[2], M.F2_ZERO, h[4], M.F2_ZERO]         # the c0 half, embedded (v = w^2)
            if any(c != M.F2_ZERO for c in h0):
                break
        t_limbs = [adversarial_limbs(rnd, (k + e) % 8) for e in range(6)]
        t0 = [field_of(x) for x in t_limbs]
        t0p = [(t0[0], t0[1]), M.F2_ZERO, (t0[2], t0[3]), M.F2_ZERO, (t0[4], t0[5]), M.F2_ZERO]
        h0_inv = M.f12_pow(h0, Q ** 6 - 2)
        assert M.f12_mul(h0, h0_inv) == M.F12_ONE
        s = M.f12_mul(t0p, h0_inv)
        assert all(s[i] == M.F2_ZERO for i in (1, 3, 5))                       # s is in Fq6
        f = M.f12_mul(h, s)
        tower = poly_to_tower(f)
        assert tower[:6] == t0
        limbs = list(t_limbs)
        for e in range(6, 12):
            v = tower[e] * MONT_R % Q
            if v > Q // 2:
                v -= Q                                                            # the symmetric representative, tight digits
            limbs.append(to_tight_limbs(v))
        for x in limbs:
            assert abs(value_of(x)) <= VMAX * Q and all(-T <= d <= T for d in x[:8])
        assert [field_of(x) for x in limbs] == tower
        gt = M.final_exponentiation(f)
        assert gt == M.F12_ONE
        vectors.append({"family": "one", "limbs": [d for x in limbs for d in x], "gt": M.f12_to_bytes(gt).hex(), "status": 0})
        print("one", k, flush=True)
    out = {"comment": "generated by tests/golden/gen_adversarial_fe.py (seed 20261004); limbs: 12 coefficients (Gt order) x 9 int32, Montgomery form R = 2^261",
           "contract": {"limb_abs_max": T, "top_abs_max": TOPMAX, "value_over_q_abs_max": VMAX}, "vectors": vectors}
    with open(os.path.join(ROOT, "tests", "golden", "adversarial_fe_vectors.json"), "w") as fh:
        json.dump(out, fh, indent=0)
    print("wrote", len(vectors), "vectors")


if __name__ == "__main__":
    main()
