"""The C restatement (oracle/bn254_oracle.c) against the reference's known-answer vectors, the
derived golden vectors and the independent big-integer model.  CPU only."""
import hashlib

import pytest

from oracle import bn254_model as m
from oracle import c_oracle as c

H = bytes.fromhex


def test_generators():
    assert c.g1_generator() == m.g1_to_uncompressed(m.G1_GEN)
    assert c.g2_generator() == m.g2_to_uncompressed(m.G2_GEN)


def test_hash_kats(kats, derived):
    for v in kats["hash_to_g1"]:
        st, pt, _ = c.hash_to_g1(H(v["message_hex"]))
        assert st == 0 and c.g1_compress(pt).hex() == v["compressed"]
    for v in derived["hash_to_g1"]:
        st, pt, tries = c.hash_to_g1(H(v["message_hex"]))
        assert st == 0 and pt.hex() == v["uncompressed"] and tries == v["tries"]


def test_sign_and_keys(kats):
    for v in kats["sign"]:
        assert c.g1_compress(c.sign(H(v["message_hex"]), H(v["private_key"]))).hex() == v["signature_compressed"]
    for v in kats["public_key_from_private_key"]:
        assert c.public_key_g2(H(v["private_key"])).hex() == v["uncompressed"]


def test_verify_kats(kats):
    for v in kats["verify_ok"]:
        sig = c.g1_decompress(H(v["signature_compressed"]))
        assert c.verify(H(v["message_hex"]), sig, c.public_key_g2(H(v["private_key"]))) == 0
    a = kats["aggregate"]
    msg = H(a["message_hex"])
    sigs = [c.sign(msg, H(k)) for k in a["private_keys"]]
    pks = [c.public_key_g2(H(k)) for k in a["private_keys"]]
    for s, p in zip(sigs, pks):
        assert c.verify(msg, s, p) == 0
    assert c.verify(msg, c.g1_add(*sigs), c.g2_add(*pks)) == 0
    assert c.verify(msg, sigs[0], pks[1]) == 9
    for v in kats["check_public_keys"]:
        assert c.check_public_keys(c.public_key_g2(H(v["sk_g2"])), c.public_key_g1(H(v["sk_g1"]))) == v["status"]


def test_example(kats, derived):
    ex = kats["example"]
    msg = ex["message"].encode()
    sigs = [c.sign(msg, H(k)) for k in ex["private_keys"]]
    pks = [c.public_key_g2(H(k)) for k in ex["private_keys"]]
    s, p = c.g1_add(*sigs), c.g2_add(*pks)
    assert s.hex() == derived["example"]["agg_sig"] and p.hex() == derived["example"]["agg_pk"]
    assert c.verify(msg, s, p) == 0


def test_bn256_vectors(kats):
    for v in kats["g1_add"]:
        assert c.g1_add(H(v["x1"] + v["y1"]), H(v["x2"] + v["y2"])).hex() == v["result"]
    for v in kats["g1_mul"]:
        assert c.g1_mul(H(v["x"] + v["y"]), H(v["scalar"])).hex() == v["result"]
    assert c.g1_compress(c.g1_add(c.g1_generator(), c.g1_generator())).hex() == kats["g1_double_generator_compressed"]["hex"]


def test_gt_golden(derived):
    for v in derived["pairing_gt"]:
        assert c.pairing(H(v["g1"]), H(v["g2"])).hex() == v["gt"]
    # identity members contribute one
    assert c.pairing(bytes(64), H(derived["g2_generator"])).hex() == derived["gt_one"]
    assert c.pairing(c.g1_generator(), bytes(128)).hex() == derived["gt_one"]


def test_verify_cases(derived):
    for v in derived["verify_cases"]:
        assert c.verify(H(v["message_hex"]), H(v["sig"]), H(v["pk"])) == v["status"], v["name"]
    # without the subgroup flag the off-subgroup key is accepted by the decoder and simply fails the pairing check
    v = [x for x in derived["verify_cases"] if "not-in-subgroup" in x["name"]][0]
    assert c.verify(H(v["message_hex"]), H(v["sig"]), H(v["pk"]), 0) == 9
    # strict decoding rejects the all-zero identity encoding like from_uncompressed does
    assert c.verify(b"x", bytes(64), bytes(128), c.FLAG_REJECT_IDENTITY) == 4


def test_random_pairings_match_model():
    """C oracle vs the independent model on seeded random points: canonical Gt bytes identical."""
    for i in range(3):
        a = int.from_bytes(hashlib.sha256(b"a%d" % i).digest(), "big") % m.R
        b = int.from_bytes(hashlib.sha256(b"b%d" % i).digest(), "big") % m.R
        P, Q2 = m.g1_mul(m.G1_GEN, a), m.g2_mul(m.G2_GEN, b)
        assert c.g1_mul(c.g1_generator(), a.to_bytes(32, "big")) == m.g1_to_uncompressed(P)
        assert c.g2_mul(c.g2_generator(), b.to_bytes(32, "big")) == m.g2_to_uncompressed(Q2)
        assert c.pairing(m.g1_to_uncompressed(P), m.g2_to_uncompressed(Q2)) == m.f12_to_bytes(m.pairing(P, Q2))


def test_batch_verify_threads(derived):
    cases = derived["verify_cases"]
    msgs = [H(v["message_hex"]) for v in cases]
    sigs = b"".join(H(v["sig"]) for v in cases)
    pks = b"".join(H(v["pk"]) for v in cases)
    want = bytes(v["status"] for v in cases)
    for nt in (1, 3):
        st, cnt = c.batch_verify(msgs, sigs, pks, nthreads=nt)
        assert st == want and cnt > 0
