#!/usr/bin/env python3
"""Generate tests/golden/derived_vectors.json from the slow big-integer model
(oracle/bn254_model.py).  These are NOT from the reference's tests — they are "derived"
vectors that pin branches the reference's own KATs do not reach (multi-try hashes, the >= 5q
rejection branch, negative verifies, identity inputs, decode failures, canonical Gt bytes).
The model itself is pinned on every reference KAT (tests/test_oracle_model.py).

Run:  python tests/golden/gen_golden.py      (takes ~1 minute; pure Python)
"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import bn254_model as m  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "derived_vectors.json")


def D(tag, i):
    """deterministic synthetic-data derivation used everywhere: SHA256(tag || le64(i))"""
    return hashlib.sha256(tag.encode() + i.to_bytes(8, "little")).digest()


def sk_of(j):
    return (int.from_bytes(D("bn254/sk", j), "big") % (m.R - 1)) + 1


def main():
    out = {"_comment": "Derived vectors from oracle/bn254_model.py (tests/golden/gen_golden.py); not from the reference's tests."}

    # --- hash-to-G1: find messages with 1..6 and a long try count, and one that hits the >=5q branch
    hashes = []
    seen_tries = set()
    hit_reject = None
    i = 0
    while len(seen_tries) < 7 or hit_reject is None:
        msg = b"msg-%d" % i
        p, tries = m.hash_to_try_and_increment_ex(msg)
        h0 = int.from_bytes(hashlib.sha256(msg + b"\x00").digest(), "big")
        rej = h0 >= 5 * m.Q
        want = (tries not in seen_tries and tries <= 6) or (tries >= 9 and not any(t >= 9 for t in seen_tries))
        if want or (rej and hit_reject is None):
            hashes.append({"message_hex": msg.hex(), "tries": tries, "first_hash_ge_5q": rej,
                           "uncompressed": m.g1_to_uncompressed(p).hex(), "compressed": m.g1_to_compressed(p).hex()})
            seen_tries.add(tries)
            if rej:
                hit_reject = i
        i += 1
    # survey Appendix E vectors, re-derived
    for name in (b"msg-23", b"msg-19", b"msg-967", b"sample", b""):
        p, tries = m.hash_to_try_and_increment_ex(name)
        hashes.append({"message_hex": name.hex(), "tries": tries, "uncompressed": m.g1_to_uncompressed(p).hex(),
                       "compressed": m.g1_to_compressed(p).hex()})
    # long messages (multi-block SHA-256 padding edges: 54,55,56,63,64,119,120 bytes)
    for ln in (54, 55, 56, 63, 64, 119, 120, 200):
        msg = bytes((7 * k + ln) & 0xFF for k in range(ln))
        p, tries = m.hash_to_try_and_increment_ex(msg)
        hashes.append({"message_hex": msg.hex(), "tries": tries, "uncompressed": m.g1_to_uncompressed(p).hex(),
                       "compressed": m.g1_to_compressed(p).hex()})
    out["hash_to_g1"] = hashes

    # --- canonical Gt values
    gts = []
    for (a, b) in ((1, 1), (5, 7), (sk_of(0), sk_of(1))):
        P = m.g1_mul(m.G1_GEN, a)
        Qp = m.g2_mul(m.G2_GEN, b)
        gts.append({"a": hex(a), "b": hex(b), "g1": m.g1_to_uncompressed(P).hex(), "g2": m.g2_to_uncompressed(Qp).hex(),
                    "gt": m.f12_to_bytes(m.pairing(P, Qp)).hex()})
    out["pairing_gt"] = gts
    out["gt_one"] = m.f12_to_bytes(m.F12_ONE).hex()

    # --- verify cases (status codes)
    cases = []
    keys = [sk_of(j) for j in range(4)]
    pks = [m.public_key(k) for k in keys]
    zero_g1, zero_g2 = bytes(64), bytes(128)
    for i in range(6):
        msg = D("bn254/msg2", i)
        sig = m.sign(msg, keys[i % 4])
        cases.append({"name": "valid-%d" % i, "message_hex": msg.hex(), "sig": m.g1_to_uncompressed(sig).hex(),
                      "pk": m.g2_to_uncompressed(pks[i % 4]).hex(), "status": 0})
    msg = D("bn254/msg2", 100)
    sig = m.sign(msg, keys[0])
    good_sig, good_pk = m.g1_to_uncompressed(sig), m.g2_to_uncompressed(pks[0])
    cases.append({"name": "wrong-key", "message_hex": msg.hex(), "sig": good_sig.hex(), "pk": m.g2_to_uncompressed(pks[1]).hex(), "status": 9})
    cases.append({"name": "wrong-message", "message_hex": D("bn254/msg2", 101).hex(), "sig": good_sig.hex(), "pk": good_pk.hex(), "status": 9})
    cases.append({"name": "negated-sig", "message_hex": msg.hex(), "sig": m.g1_to_uncompressed(m.g1_neg(sig)).hex(), "pk": good_pk.hex(), "status": 9})
    cases.append({"name": "identity-sig-identity-pk (both pairs skipped -> Ok, SURVEY D-7)", "message_hex": msg.hex(), "sig": zero_g1.hex(), "pk": zero_g2.hex(), "status": 0})
    cases.append({"name": "identity-sig only", "message_hex": msg.hex(), "sig": zero_g1.hex(), "pk": good_pk.hex(), "status": 9})
    cases.append({"name": "identity-pk only", "message_hex": msg.hex(), "sig": good_sig.hex(), "pk": zero_g2.hex(), "status": 9})
    bad = bytearray(good_sig); bad[63] ^= 1
    cases.append({"name": "sig-off-curve", "message_hex": msg.hex(), "sig": bytes(bad).hex(), "pk": good_pk.hex(), "status": 4})
    cases.append({"name": "sig-x-ge-q", "message_hex": msg.hex(), "sig": (m.Q.to_bytes(32, "big") + good_sig[32:]).hex(), "pk": good_pk.hex(), "status": 6})
    cases.append({"name": "sig-y-ge-q", "message_hex": msg.hex(), "sig": (good_sig[:32] + b"\xff" * 32).hex(), "pk": good_pk.hex(), "status": 6})
    badpk = bytearray(good_pk); badpk[127] ^= 1
    cases.append({"name": "pk-off-curve", "message_hex": msg.hex(), "sig": good_sig.hex(), "pk": bytes(badpk).hex(), "status": 4})
    cases.append({"name": "pk-coord-ge-q", "message_hex": msg.hex(), "sig": good_sig.hex(), "pk": (good_pk[:32] + m.Q.to_bytes(32, "big") + good_pk[64:]).hex(), "status": 6})
    # a point on the twist but outside the order-r subgroup: pick x until on curve
    x = (1, 0)
    while True:
        y = m.f2_sqrt(m.f2_add(m.f2_mul(m.f2_mul(x, x), x), m.B2))
        if y is not None and not m.g2_in_subgroup((x, y)):
            break
        x = (x[0] + 1, 0)
    off_sub = m.g2_to_uncompressed((x, y))
    cases.append({"name": "pk-on-curve-not-in-subgroup (flags bit0 set)", "message_hex": msg.hex(), "sig": good_sig.hex(), "pk": off_sub.hex(), "status": 4})
    # aggregate (same message)
    msg = b"sample"
    sigs = [m.sign(msg, k) for k in keys[:3]]
    agg_sig = None
    agg_pk = None
    for s, p in zip(sigs, pks[:3]):
        agg_sig = m.g1_add(agg_sig, s)
        agg_pk = m.g2_add(agg_pk, p)
    cases.append({"name": "aggregate-3", "message_hex": msg.hex(), "sig": m.g1_to_uncompressed(agg_sig).hex(), "pk": m.g2_to_uncompressed(agg_pk).hex(), "status": 0})
    cases.append({"name": "aggregate-3-missing-one-key", "message_hex": msg.hex(), "sig": m.g1_to_uncompressed(agg_sig).hex(),
                  "pk": m.g2_to_uncompressed(m.g2_add(pks[0], pks[1])).hex(), "status": 9})
    out["verify_cases"] = cases
    out["g2_not_in_subgroup"] = off_sub.hex()

    # --- example scenario (keys > r), SURVEY E7
    ex1 = m.private_key_from_bytes(bytes.fromhex("c9afa9d845ba75166b5c215767b1d6934e50c3db36e89b127b8a622b120f6721"))
    ex2 = m.private_key_from_bytes(bytes.fromhex("a55e93edb1350916bf5beea1b13d8f198ef410033445bcb645b65be5432722f1"))
    s = m.g1_add(m.sign(b"sample", ex1), m.sign(b"sample", ex2))
    p = m.g2_add(m.public_key(ex1), m.public_key(ex2))
    out["example"] = {"sk_reduced": [hex(ex1), hex(ex2)], "agg_sig_compressed": m.g1_to_compressed(s).hex(),
                      "agg_pk_compressed": m.g2_to_compressed(p).hex(), "agg_sig": m.g1_to_uncompressed(s).hex(),
                      "agg_pk": m.g2_to_uncompressed(p).hex(), "status": m.verify_status(b"sample", s, p)}
    out["neg_g2_generator"] = m.g2_to_uncompressed(m.g2_neg(m.G2_GEN)).hex()
    out["g2_generator"] = m.g2_to_uncompressed(m.G2_GEN).hex()
    out["g2_generator_compressed"] = m.g2_to_compressed(m.G2_GEN).hex()

    # every case's status is what the model says (decode errors are defined by the decoders)
    for c in out["verify_cases"]:
        try:
            sg = None if c["sig"] == zero_g1.hex() else m.g1_from_uncompressed(bytes.fromhex(c["sig"]))
            pk = None if c["pk"] == zero_g2.hex() else m.g2_from_uncompressed(bytes.fromhex(c["pk"]))
            st = m.verify_status(bytes.fromhex(c["message_hex"]), sg, pk)
        except m.Bn254Error as e:
            st = e.code
        assert st == c["status"], (c["name"], st)

    with open(OUT, "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    print("wrote", OUT)


if __name__ == "__main__":
    main()
