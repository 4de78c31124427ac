"""Randomised batch verification — CPU side: the oracle's restatement against
the exact per-item oracle and against the Python big-int model, and the host compilation of the kernels'
source (tests/hostsim) against the oracle.  GPU parity: tests/test_gpu_parity.py."""
import hashlib
import random

from oracle import bn254_model as m
from oracle import c_oracle as c
from tests import hostsim_binding as hs

SEED = hashlib.sha256(b"bn254/rand-seed").digest()


def D(tag, i):
    return hashlib.sha256(tag + i.to_bytes(8, "little")).digest()


def make_batch(n, n_keys=3, tag=b"rnd/msg"):
    sks = [D(b"rnd/sk", j) for j in range(n_keys)]
    pks = [c.public_key_g2(sk) for sk in sks]
    msgs = [D(tag, i)[: 1 + i % 32] for i in range(n)]
    sigs = [c.sign(msgs[i], sks[i % n_keys]) for i in range(n)]
    return msgs, sigs, [pks[i % n_keys] for i in range(n)]


def test_random_scalar_derivation_matches_model():
    """r_i = first 16 bytes of SHA-256(seed || le64(i)), little-endian: check the combined equation of one
    small group with the Python model's own arithmetic."""
    msgs, sigs, pks = make_batch(3)
    st, gr = c.batch_verify_randomized(msgs, b"".join(sigs), b"".join(pks), SEED, flags=0)
    assert st == bytes(3) and gr == b"\x01"
    rs = [int.from_bytes(hashlib.sha256(SEED + i.to_bytes(8, "little")).digest()[:16], "little") for i in range(3)]
    # e(sum r_i sig_i, -G2) * prod e(r_i H_i, pk_i) == 1 as one 4-pair pairing check with independently derived r_i
    g1s, g2s = b"", b""
    total = bytes(64)
    hs_pts = [c.hash_to_g1(msg)[1] for msg in msgs]
    for i in range(3):
        g1s += c.g1_mul(hs_pts[i], rs[i].to_bytes(32, "big"))
        g2s += pks[i]
        total = c.g1_add(total, c.g1_mul(sigs[i], rs[i].to_bytes(32, "big")))
    neg_g2 = c.g2_mul(c.g2_generator(), (m.R - 1).to_bytes(32, "big"))
    assert c.pairing_check(g1s + total, g2s + neg_g2, 4) == 0
    # a wrong scalar for one item breaks it
    bad = c.g1_mul(hs_pts[0], (rs[0] + 1).to_bytes(32, "big")) + g1s[64:]
    assert c.pairing_check(bad + total, g2s + neg_g2, 4) == 9


def test_oracle_randomized_equals_exact():
    n = 70                                            # one full group + a ragged one
    msgs, sigs, pks = make_batch(n)
    sigs = list(sigs)
    pks = list(pks)
    exact = lambda: c.batch_verify(msgs, b"".join(sigs), b"".join(pks), flags=0)[0]
    st, gr = c.batch_verify_randomized(msgs, b"".join(sigs), b"".join(pks), SEED, flags=0)
    assert st == bytes(n) and gr == b"\x01\x01"
    # cancelling pair: sig_a + d, sig_b - d keeps the plain sum of signatures intact
    d = c.g1_mul(c.g1_generator(), (12345).to_bytes(32, "big"))
    dn = c.g1_mul(c.g1_generator(), (m.R - 12345).to_bytes(32, "big"))
    sigs[3] = c.g1_add(sigs[3], d)
    sigs[9] = c.g1_add(sigs[9], dn)
    st, gr = c.batch_verify_randomized(msgs, b"".join(sigs), b"".join(pks), SEED, flags=0)
    assert gr == b"\x00\x01" and st == exact() and st[3] == 9 and st[9] == 9 and sum(st) == 18
    # decode error + identity signature + swapped key in the ragged group; first group valid again
    sigs[3], sigs[9] = make_batch(n)[1][3], make_batch(n)[1][9]
    sigs[65] = b"\xff" * 64
    sigs[66] = bytes(64)
    pks[67] = pks[68]
    st, gr = c.batch_verify_randomized(msgs, b"".join(sigs), b"".join(pks), SEED, flags=0)
    assert gr == b"\x01\x00" and st == exact() and st[65] == 6 and st[66] == 9 and st[67] == 9
    # 64-bit scalars, GLV scalars
    for fl in (c.FLAG_RAND64, c.FLAG_RAND_GLV):
        st, gr = c.batch_verify_randomized(msgs, b"".join(sigs), b"".join(pks), SEED, flags=fl)
        assert st == exact() and gr == b"\x01\x00"


def test_windowed_scalar_mul_device_source():
    rng = random.Random(11)
    g = c.g1_generator()
    pts = [g, c.g1_mul(g, rng.getrandbits(250).to_bytes(32, "big"))]
    ks = [0, 1, 8, 9, 16, 2**128 - 1, 2**127, int("8" * 32, 16), int("9" * 32, 16), int("7" * 32, 16)] + [rng.getrandbits(128) for _ in range(12)]
    for p in pts:
        for k in ks:
            out, st = hs.g1_mul_u128(p, k)
            assert st == 0 and out == c.g1_mul(p, k.to_bytes(32, "big")), hex(k)


def test_hostsim_randomized_matches_oracle():
    n = 70
    msgs, sigs, pks = make_batch(n)
    sigs = list(sigs)
    for flags in (0, c.FLAG_RAND64, c.FLAG_RAND_GLV, 0x80000000, 0x80000000 | c.FLAG_RAND_GLV):   # bit 31: hostsim-only knob (two items per lane)
        st, gr = hs.verify_randomized(msgs, b"".join(sigs), b"".join(pks), SEED, flags)
        assert (st, gr) == c.batch_verify_randomized(msgs, b"".join(sigs), b"".join(pks), SEED, flags=flags & 0xFFFF) == (bytes(n), b"\x01\x01")
    sigs[5] = sigs[4]
    sigs[66] = b"\xff" * 64
    sigs[69] = bytes(64)
    st, gr = hs.verify_randomized(msgs, b"".join(sigs), b"".join(pks), SEED, 0)
    assert (st, gr) == c.batch_verify_randomized(msgs, b"".join(sigs), b"".join(pks), SEED, flags=0)
    assert (st, gr) == hs.verify_randomized(msgs, b"".join(sigs), b"".join(pks), SEED, 0x80000000)
    assert (st, gr) == hs.verify_randomized(msgs, b"".join(sigs), b"".join(pks), SEED, c.FLAG_RAND_GLV)
    assert gr == b"\x00\x00" and [i for i in range(n) if st[i]] == [5, 66, 69]
    # a different seed gives the same verdicts
    assert hs.verify_randomized(msgs, b"".join(sigs), b"".join(pks), bytes(32), 0)[0] == st


def test_glv_endomorphism_constant():
    """lambda * P == (beta x, y): the oracle multiplies by lambda without the endomorphism, the device source uses it"""
    lam = 0xB3C4D79D41A917585BFC41088D8DAAA78B17EA66B99C90DD
    beta = 0x59E26BCEA0D48BACD4F263F1ACDB5C4F5763473177FFFFFE
    assert (lam * lam + lam + 1) % m.R == 0 and pow(beta, 3, m.Q) == 1
    for i in range(3):
        p = c.hash_to_g1(b"glv%d" % i)[1]
        x, y = int.from_bytes(p[:32], "big"), p[32:]
        assert c.g1_mul(p, lam.to_bytes(32, "big")) == (beta * x % m.Q).to_bytes(32, "big") + y
