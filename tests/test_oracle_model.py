"""The slow big-integer model (oracle/bn254_model.py) against every known-answer vector the
reference's own tests hold (tests/golden/reference_kats.json).  CPU only."""
import pytest

from oracle import bn254_model as m

H = bytes.fromhex


def sk(hexstr):
    return m.private_key_from_bytes(H(hexstr))


def test_curve_constants(kats):
    assert m.g1_on_curve(m.G1_GEN) and m.g2_on_curve(m.G2_GEN) and m.g2_in_subgroup(m.G2_GEN)
    assert m.g1_mul(m.G1_GEN, m.R) is None
    # /root/reference/src/hash_test.rs:33-43
    assert 5 * m.Q == int(kats["last_multiple_of_fq_modulus_lower_than_2_256"]["hex"], 16)
    assert 6 * m.Q >= 2**256 and (6 * m.Q) % 2**256 < m.Q


def test_hash_to_g1_kats(kats):
    for v in kats["hash_to_g1"]:
        assert m.g1_to_compressed(m.hash_to_try_and_increment(H(v["message_hex"]))).hex() == v["compressed"]


def test_sign_kat(kats):
    for v in kats["sign"]:
        sig = m.sign(H(v["message_hex"]), sk(v["private_key"]))
        assert m.g1_to_compressed(sig).hex() == v["signature_compressed"]


def test_verify_kat(kats):
    for v in kats["verify_ok"]:
        sig = m.g1_from_compressed(H(v["signature_compressed"]))
        assert m.verify_status(H(v["message_hex"]), sig, m.public_key(sk(v["private_key"]))) == m.OK
        # uncompressed round trip then verify, /root/reference/src/ecdsa_test.rs:131-153
        sig2 = m.g1_from_uncompressed(m.g1_to_uncompressed(sig))
        assert m.verify_status(H(v["message_hex"]), sig2, m.public_key(sk(v["private_key"]))) == m.OK


def test_aggregate(kats):
    a = kats["aggregate"]
    msg = H(a["message_hex"])
    sks = [sk(x) for x in a["private_keys"]]
    sigs = [m.sign(msg, k) for k in sks]
    pks = [m.public_key(k) for k in sks]
    for s, p in zip(sigs, pks):
        assert m.verify_status(msg, s, p) == m.OK
    assert m.verify_status(msg, m.g1_add(*sigs), m.g2_add(*pks)) == m.OK


def test_check_public_keys(kats):
    for v in kats["check_public_keys"]:
        assert m.check_public_keys_status(m.public_key(sk(v["sk_g2"])), m.public_key_g1(sk(v["sk_g1"]))) == v["status"]


def test_private_key_bytes(kats):
    hx = kats["private_key_roundtrip"]["hex"]
    assert sk(hx).to_bytes(32, "big").hex() == hx
    for bad in kats["private_key_invalid_length"]["hex"]:
        with pytest.raises(m.Bn254Error) as e:
            m.private_key_from_bytes(H(bad))
        assert e.value.code == m.ERR_INVALID_LENGTH


def test_g2_codecs(kats):
    c = H(kats["g2_compressed_roundtrip"]["hex"])
    assert m.g2_to_compressed(m.g2_from_compressed(c)) == c
    u = H(kats["g2_uncompressed_roundtrip"]["hex"])
    assert m.g2_to_uncompressed(m.g2_from_uncompressed(u)) == u
    for v in kats["public_key_from_private_key"]:
        assert m.g2_to_uncompressed(m.public_key(sk(v["private_key"]))).hex() == v["uncompressed"]
    assert m.g2_to_compressed(m.g2_add(m.G2_GEN, m.G2_GEN)).hex() == kats["g2_double_generator_compressed"]["hex"]
    assert m.g1_to_compressed(m.g1_add(m.G1_GEN, m.G1_GEN)).hex() == kats["g1_double_generator_compressed"]["hex"]


def _pt(xh, yh):
    x, y = int(xh, 16), int(yh, 16)
    return None if x == 0 and y == 0 else (x, y)


def _enc(p):
    return bytes(64) if p is None else m.g1_to_uncompressed(p)


def test_bn256_add_mul(kats):
    for v in kats["g1_add"]:
        assert _enc(m.g1_add(_pt(v["x1"], v["y1"]), _pt(v["x2"], v["y2"]))).hex() == v["result"]
    for v in kats["g1_mul"]:
        assert _enc(m.g1_mul(_pt(v["x"], v["y"]), int(v["scalar"], 16))).hex() == v["result"]


def test_example_scenario(kats, derived):
    ex = kats["example"]
    ks = [sk(x) for x in ex["private_keys"]]
    assert [hex(k) for k in ks] == derived["example"]["sk_reduced"]
    msg = ex["message"].encode()
    s = m.g1_add(m.sign(msg, ks[0]), m.sign(msg, ks[1]))
    p = m.g2_add(m.public_key(ks[0]), m.public_key(ks[1]))
    assert m.verify_status(msg, s, p) == m.OK


def test_final_exponentiation_fast_path_agrees():
    """naive pow(f,(q^12-1)/r) == easy part via conj/inverse-free identity: f^(q^6-1) = conj(f)/f."""
    f = m.miller_loop(m.g1_mul(m.G1_GEN, 3), m.g2_mul(m.G2_GEN, 11))
    e1 = m.final_exponentiation(f)
    assert m.f12_pow(e1, m.R) == m.F12_ONE and e1 != m.F12_ONE
    # bilinearity
    assert e1 == m.f12_pow(m.pairing(m.G1_GEN, m.G2_GEN), 33)


def test_g2_endomorphism_subgroup_test_is_equivalent():
    """[u+1]P + psi([u]P) + psi^2([u]P) == psi^3([2u]P)  <=>  [r]P == O, on twist points in and out of G2
    (the product's decoder uses the left-hand test; this pins it to the definition)."""
    import random
    gx, gy = m.f2_pow(m.XI, (m.Q - 1) // 3), m.f2_pow(m.XI, (m.Q - 1) // 2)

    def psi(p):
        return None if p is None else (m.f2_mul(m.f2_conj(p[0]), gx), m.f2_mul(m.f2_conj(p[1]), gy))

    def fast(p):
        up = m.g2_mul(p, m.U)
        lhs = m.g2_add(m.g2_add(m.g2_add(up, p), psi(up)), psi(psi(up)))
        return lhs == psi(psi(psi(m.g2_add(up, up))))

    rnd = random.Random(5)
    for _ in range(3):
        assert fast(m.g2_mul(m.G2_GEN, rnd.randrange(1, m.R)))
    n = 0
    while n < 6:
        x = (rnd.randrange(m.Q), rnd.randrange(m.Q))
        y = m.f2_sqrt(m.f2_add(m.f2_mul(m.f2_mul(x, x), x), m.B2))
        if y is None:
            continue
        n += 1
        p = (x, y)
        assert fast(p) == m.g2_in_subgroup(p)
        cleared = m.g2_mul(p, 2 * m.Q - m.R)
        assert fast(cleared) and m.g2_in_subgroup(cleared)
        mixed = m.g2_add(cleared, m.g2_mul(p, m.R))
        assert fast(mixed) == m.g2_in_subgroup(mixed)
