"""GPU parity tests: the HIP path, through the C ABI (libbn254hip.so), against the oracle on the
same inputs — bit-exact (integer/byte work, no tolerance).  Run on the MI355X box: -m gpu."""
import hashlib
import random

import pytest

pytestmark = pytest.mark.gpu

H = bytes.fromhex
Q = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001


@pytest.fixture(scope="module")
def eng():
    import bn254_amd
    from bn254_amd.engine import OPT_RAND_MIN_BATCH
    e = bn254_amd.Engine(0)
    e.set_option(OPT_RAND_MIN_BATCH, 0)      # the randomised-mode tests must run the randomised kernels at every size
    return e


@pytest.fixture(scope="module")
def c():
    from oracle import c_oracle
    return c_oracle


def test_native_library_loaded(eng):
    assert "gfx950" in eng.version()
    with open("/proc/self/maps") as f:
        assert "libbn254hip.so" in f.read()


# ---- layer by layer ------------------------------------------------------------------------
def test_fp_ops(eng):
    rnd = random.Random(11)
    edge = [0, 1, 2, Q - 1, Q - 2, (1 << 256) % Q, 0xFFFFFFFF, 1 << 32, (1 << 224) - 1]
    n = 4096
    a = [edge[i % len(edge)] if i < 81 else rnd.randrange(Q) for i in range(n)]
    b = [edge[(i // len(edge)) % len(edge)] if i < 81 else rnd.randrange(Q) for i in range(n)]
    ab = b"".join(x.to_bytes(32, "big") for x in a)
    bb = b"".join(x.to_bytes(32, "big") for x in b)
    for op, fn in ((0, lambda x, y: x * y % Q), (1, lambda x, y: (x + y) % Q), (2, lambda x, y: (x - y) % Q), (4, lambda x, y: x * x % Q)):
        out, st = eng.debug_fp_op(op, ab, bb, n)
        assert st == bytes(n)
        got = [int.from_bytes(out[32 * i:32 * i + 32], "big") for i in range(n)]
        assert got == [fn(x, y) for x, y in zip(a, b)], "fp op %d" % op
    out, _ = eng.debug_fp_op(3, ab, None, n)
    for i in range(n):
        assert int.from_bytes(out[32 * i:32 * i + 32], "big") == (pow(a[i], -1, Q) if a[i] else 0)
    sq = b"".join((x * x % Q).to_bytes(32, "big") for x in a)
    out, st = eng.debug_fp_op(5, sq, None, n)
    assert st == bytes(n)
    for i in range(n):
        assert pow(int.from_bytes(out[32 * i:32 * i + 32], "big"), 2, Q) == a[i] * a[i] % Q
    # x >= q is NotMemberError(6)
    _, st = eng.debug_fp_op(0, Q.to_bytes(32, "big"), (1).to_bytes(32, "big"), 1)
    assert st[0] == 6


def _rand_points(c, n, seed):
    g1, g2 = c.g1_generator(), c.g2_generator()
    ps, qs = [], []
    for i in range(n):
        a = int.from_bytes(hashlib.sha256(b"%s-a%d" % (seed, i)).digest(), "big") % R
        b = int.from_bytes(hashlib.sha256(b"%s-b%d" % (seed, i)).digest(), "big") % R
        ps.append(c.g1_mul(g1, a.to_bytes(32, "big")))
        qs.append(c.g2_mul(g2, b.to_bytes(32, "big")))
    return ps, qs


def test_miller_loop_raw(eng, c):
    ps, qs = _rand_points(c, 70, b"ml")
    got = eng.debug_miller_loop(b"".join(ps), b"".join(qs), 70)
    for i in range(70):
        assert got[384 * i:384 * i + 384] == c.miller_loop(ps[i], qs[i]), i


def test_fp12_ops(eng, c):
    ps, qs = _rand_points(c, 66, b"f12")
    fs = [c.miller_loop(ps[i], qs[i]) for i in range(66)]       # generic Fq12 elements
    gts = [c.pairing(ps[i], qs[i]) for i in range(66)]          # cyclotomic-subgroup elements
    a = b"".join(fs)
    # final exponentiation of raw Miller values == oracle Gt
    out = eng.debug_fp12_op(8, a, None, 66)
    assert out == b"".join(gts)
    # algebraic identities checked on the device results themselves
    sq = eng.debug_fp12_op(1, a, None, 66)
    assert sq == eng.debug_fp12_op(0, a, a, 66)
    inv = eng.debug_fp12_op(2, a, None, 66)
    one = eng.debug_fp12_op(0, a, inv, 66)
    from tests.conftest import GOLDEN  # noqa: F401
    gt_one = (1).to_bytes(32, "big") + bytes(352)
    assert one == gt_one * 66
    g = b"".join(gts)
    assert eng.debug_fp12_op(7, g, None, 66) == eng.debug_fp12_op(1, g, None, 66)     # cyclotomic sqr == sqr on Gt
    assert eng.debug_fp12_op(0, g, eng.debug_fp12_op(3, g, None, 66), 66) == gt_one * 66   # conj = inverse on Gt
    f1 = eng.debug_fp12_op(4, a, None, 66)
    f2 = eng.debug_fp12_op(5, a, None, 66)
    f3 = eng.debug_fp12_op(6, a, None, 66)
    assert eng.debug_fp12_op(4, f1, None, 66) == f2 and eng.debug_fp12_op(4, f2, None, 66) == f3


def test_adversarial_limbs_in_every_final_exponentiation_layout(eng):
    """Every layout of the final exponentiation of ECDSA::verify (/root/reference/src/ecdsa.rs:57-59) — one lane (exact and == one chains),
    lane pairs (both programs of the accumulator machine), lane octets (straight-line chains below 128 items, machine from 128 on), nine
    lane pairs and eighteen (one verify per wave: the default for every single verify) — on LIMB vectors at the edge of the interval tracker's contract for a Miller value:
    non-canonical representatives, extreme balanced digits and top limbs (tests/golden/adversarial_fe_vectors.json; expected results
    from the independent big-integer model).  Family "full" (all 12 coefficients adversarial): canonical Gt bytes where a layout writes
    them, status 9 everywhere; family "one" (f = g^r s, s in Fq6 forcing six adversarial coefficients): status 0 from every layout — a
    column overflow anywhere in a status-only kernel would turn it into 9."""
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "adversarial_fe_vectors.json")) as f:
        adv = json.load(f)
    vec = adv["vectors"]
    want_st = bytes(v["status"] for v in vec)
    want_gt = b"".join(bytes.fromhex(v["gt"]) for v in vec)
    assert set(want_st) == {0, 9}
    limbs = [x for v in vec for x in v["limbs"]]
    n = len(vec)
    for layout in (0, 1):
        gt, st = eng.debug_final_exp_limbs(layout, limbs, n, want_gt=True)
        assert st == want_st, (layout, st, want_st)
        assert gt == want_gt, layout
    for layout in (2, 3, 4, 5, 6):
        _, st = eng.debug_final_exp_limbs(layout, limbs, n)
        assert st == want_st, (layout, st, want_st)
    # sizes that change the kernels' own arrangement: the octet layout's accumulator machine (>= 128 items), several workgroups of the
    # nonet kernel (12 verifies each) incl. a ragged last one, lane pairs over several waves
    reps = 5
    big = limbs * reps + limbs[:108 * 7]
    for layout in (1, 2, 3, 4, 6):
        gt, st = eng.debug_final_exp_limbs(layout, big, reps * n + 7, want_gt=(layout == 1))
        assert st == want_st * reps + want_st[:7], layout
        if gt is not None:
            assert gt == want_gt * reps + want_gt[:384 * 7]
    # the hook refuses what it cannot do
    import ctypes
    arr = (ctypes.c_int32 * 108)()
    out = ctypes.create_string_buffer(384)
    assert eng._lib.bn254_debug_final_exp_limbs(eng._h, 7, arr, 1, None, out) == -10001
    assert eng._lib.bn254_debug_final_exp_limbs(eng._h, 4, arr, 1, out, out) == -10001      # Gt bytes only from layouts 0 and 1


# ---- golden vectors --------------------------------------------------------------------------
def test_hash_to_g1_golden(eng, kats, derived):
    vs = [(H(v["message_hex"]), v["uncompressed"], v["tries"]) for v in derived["hash_to_g1"]]
    pts, st, tries = eng.batch_hash_to_g1([m for m, _, _ in vs])
    assert st == bytes(len(vs))
    for i, (_, want, t) in enumerate(vs):
        assert pts[64 * i:64 * i + 64].hex() == want and tries[i] == t
    for v in kats["hash_to_g1"]:
        pts, st, _ = eng.batch_hash_to_g1([H(v["message_hex"])])
        assert st == b"\x00" and pts[:32].hex() == v["compressed"][2:] and (pts[63] & 1) == 0


def test_hash_to_g1_vs_oracle_ragged(eng, c):
    msgs = [hashlib.sha256(b"rag%d" % i).digest()[: i % 33] * (1 + i % 7) for i in range(1000)] + [b"", b"\x00" * 200]
    pts, st, tries = eng.batch_hash_to_g1(msgs)
    for i, m in enumerate(msgs):
        wst, wpt, wtries = c.hash_to_g1(m)
        assert (st[i], pts[64 * i:64 * i + 64], tries[i]) == (wst, wpt, wtries), i


def test_hash_small_batch_direct_path_and_its_survivors(eng, c):
    """batches of up to 4096 messages try the first counters of every message at once with the square root itself
    (k_hash_direct, 32 counters by default); a message that needs more falls through to the ordinary rounds.  Messages with
    17 and 18 tries (found by a search with the oracle) with the direct width at 32, at 16 (they become survivors), 1 and 0
    (rounds only), in a large batch (rounds only), and with the counter budget cut to 17"""
    from bn254_amd.engine import OPT_HASH_DIRECT_WIDTH, OPT_HASH_MAX_TRIES
    late = [b"late-21102", b"late-32370", b"late-82852"]
    want_late = [c.hash_to_g1(m) for m in late]
    assert [w[2] for w in want_late] == [18, 17, 17]
    small = [b"d%d" % i for i in range(61)] + late + [b""]
    want = [c.hash_to_g1(m) for m in small]
    try:
        for width in (32, 16, 4, 1, 0):
            eng.set_option(OPT_HASH_DIRECT_WIDTH, width)
            for msgs in (small, late[:1], small + [b"pad%d" % i for i in range(5000)]):
                pts, st, tries = eng.batch_hash_to_g1(msgs)
                for i in range(min(len(msgs), len(small))):
                    wst, wpt, wtries = want[i] if len(msgs) > 1 else want_late[0]
                    assert (st[i], pts[64 * i:64 * i + 64], tries[i]) == (wst, wpt, wtries), (width, len(msgs), i)
            eng.set_option(OPT_HASH_MAX_TRIES, 17)
            pts, st, tries = eng.batch_hash_to_g1(small)
            eng.set_option(OPT_HASH_MAX_TRIES, 0)
            k = len(small) - 4
            assert (st[k], tries[k], pts[64 * k:64 * k + 64]) == (1, 17, bytes(64)), width               # 18 tries needed: HashToPointError
            assert (st[k + 1], tries[k + 1], pts[64 * (k + 1):64 * (k + 2)]) == (0, 17, want_late[1][1]), width
    finally:
        eng.set_option(OPT_HASH_MAX_TRIES, 0)
        eng.set_option(OPT_HASH_DIRECT_WIDTH, 32)
    with pytest.raises(Exception):
        eng.set_option(OPT_HASH_DIRECT_WIDTH, 24)


def test_hash_to_point_error_path(eng, derived):
    """src/hash.rs:62: HashToPointError once the counters are exhausted.  255 failures in a row cannot be
    provoked with real data (p = 0.53^255), so the test knob shrinks the counter budget to 3."""
    from bn254_amd.engine import OPT_HASH_MAX_TRIES
    vs = derived["hash_to_g1"]
    msgs = [H(v["message_hex"]) for v in vs]
    eng.set_option(OPT_HASH_MAX_TRIES, 3)
    try:
        pts, st, tries = eng.batch_hash_to_g1(msgs)
        sigs, st_sign = eng.batch_sign(msgs, b"".join((i + 1).to_bytes(32, "big") for i in range(len(msgs))))
    finally:
        eng.set_option(OPT_HASH_MAX_TRIES, 0)
    for i, v in enumerate(vs):
        if v["tries"] <= 3:
            assert st[i] == 0 and pts[64 * i:64 * i + 64].hex() == v["uncompressed"] and tries[i] == v["tries"]
            assert st_sign[i] == 0
        else:
            assert st[i] == 1 and pts[64 * i:64 * i + 64] == bytes(64) and tries[i] == 3
            assert st_sign[i] == 1 and sigs[64 * i:64 * i + 64] == bytes(64)     # ecdsa.rs:28 propagates the error


def test_hash_candidate_range_rules_vs_oracle(eng, c):
    """the branch no SHA-256 preimage reaches: digest values h = k*q stop at q under mod_u256's
    strict '>' (/root/reference/src/utils.rs:27-37) and are rejected, h >= 5q is skipped (src/hash.rs:49-51).  Chosen
    values of h go through the device's range / reduction / Jacobi filter / square-root code via the debug hook and are
    compared with the oracle's treatment of the same values: q, 2q, 3q, 4q, 5q-1, 5q, neighbours, edges, random."""
    rnd = random.Random(7)
    hs = [0, 1, 2, Q - 1, Q, Q + 1, 2 * Q - 1, 2 * Q, 2 * Q + 1, 3 * Q - 1, 3 * Q, 3 * Q + 1, 4 * Q - 1, 4 * Q, 4 * Q + 1, 5 * Q - 2, 5 * Q - 1, 5 * Q,
          5 * Q + 1, 2**256 - 1, 2**255, Q + 2, 2 * Q + 2, 3 * Q + 2, 4 * Q + 2]
    hs += [rnd.randrange(2**256) for _ in range(400)] + [k * Q + rnd.randrange(1, 50) for k in range(5) for _ in range(20)]
    blob = b"".join(h.to_bytes(32, "big") for h in hs)
    pts, st = eng.debug_hash_candidate(blob, len(hs))
    n_point = 0
    for i, h in enumerate(hs):
        ok, want = c.hash_candidate(h.to_bytes(32, "big"))
        assert st[i] == (0 if ok else 1), (hex(h), st[i])          # bit 7 (filter vs square root disagree) never set
        assert pts[64 * i:64 * i + 64] == want, hex(h)
        n_point += ok
    for k in range(1, 6):
        assert st[hs.index(k * Q)] == 1                              # exact multiples of q never yield a point
    assert st[hs.index(5 * Q - 1)] in (0, 1) and st[hs.index(5 * Q + 1)] == 1 and st[hs.index(2**256 - 1)] == 1
    assert 150 < n_point < 400


def test_hash_large_batch_statistics(eng, c):
    """200 000 messages: every round shape (speculative and one-try rounds) is exercised; spot-check
    against the oracle and check the try-count distribution (mean 2.116 = 1/0.4726)."""
    n = 200000
    msgs = [hashlib.sha256(b"big%d" % i).digest() for i in range(n)]
    pts, st, tries = eng.batch_hash_to_g1(msgs)
    assert st == bytes(n)
    assert abs(sum(tries) / n - 2.116) < 0.02
    for i in list(range(0, n, 997)) + [max(range(n), key=lambda k: tries[k])]:
        assert (0, pts[64 * i:64 * i + 64], tries[i]) == c.hash_to_g1(msgs[i])


def test_pairing_gt_golden(eng, derived):
    vs = derived["pairing_gt"]
    gt, st = eng.batch_pairing(b"".join(H(v["g1"]) for v in vs), b"".join(H(v["g2"]) for v in vs), len(vs))
    assert st == bytes([9]) * len(vs)           # e(P,Q) != 1
    for i, v in enumerate(vs):
        assert gt[384 * i:384 * i + 384].hex() == v["gt"]
    gt, st = eng.batch_pairing(bytes(64), H(derived["g2_generator"]), 1)
    assert gt.hex() == derived["gt_one"] and st == b"\x00"


def test_verify_cases_golden(eng, derived):
    cs = derived["verify_cases"]
    st = eng.batch_verify([H(v["message_hex"]) for v in cs], b"".join(H(v["sig"]) for v in cs), b"".join(H(v["pk"]) for v in cs), flags=1)
    assert list(st) == [v["status"] for v in cs], [v["name"] for v, s in zip(cs, st) if s != v["status"]]
    v = [x for x in cs if "not-in-subgroup" in x["name"]][0]
    assert eng.batch_verify([H(v["message_hex"])], H(v["sig"]), H(v["pk"]), flags=0) == b"\x09"
    assert eng.batch_verify([b"x"], bytes(64), bytes(128), flags=2) == b"\x04"
    assert eng.batch_verify([], b"", b"") == b""


def test_reference_kats_through_api(eng, c, kats):
    import bn254_amd as bn
    # src/ecdsa_test.rs:5-17 (sign KAT) and :20-38 (verify)
    for v in kats["sign"]:
        sk = bn.PrivateKey.try_from(v["private_key"])
        sig = bn.ECDSA.sign(H(v["message_hex"]), sk)
        assert sig.to_compressed().hex() == v["signature_compressed"]
        bn.ECDSA.verify(H(v["message_hex"]), sig, bn.PublicKey.from_private_key(sk))
        again = bn.Signature.from_uncompressed(sig.to_uncompressed())      # :131-153
        bn.ECDSA.verify(H(v["message_hex"]), again, bn.PublicKey.from_private_key(sk))
    # src/ecdsa_test.rs:41-78 aggregate
    a = kats["aggregate"]
    sks = [bn.PrivateKey.try_from(k) for k in a["private_keys"]]
    msg = H(a["message_hex"])
    sigs = [bn.ECDSA.sign(msg, k) for k in sks]
    pks = [bn.PublicKey.from_private_key(k) for k in sks]
    for s, p in zip(sigs, pks):
        bn.ECDSA.verify(msg, s, p)
    bn.ECDSA.verify(msg, sigs[0] + sigs[1], pks[0] + pks[1])
    with pytest.raises(bn.Error) as e:
        bn.ECDSA.verify(msg, sigs[0], pks[1])
    assert e.value.kind == bn.ErrorKind.VerificationFailed
    # src/ecdsa_test.rs:81-112 check_public_keys
    for v in kats["check_public_keys"]:
        pk2 = bn.PublicKey.from_private_key(bn.PrivateKey.try_from(v["sk_g2"]))
        pk1 = bn.PublicKeyG1.from_private_key(bn.PrivateKey.try_from(v["sk_g1"]))
        pk1 = bn.PublicKeyG1.from_uncompressed(pk1.to_uncompressed())      # :114-128
        if v["status"] == 0:
            bn.check_public_keys(pk2, pk1)
        else:
            with pytest.raises(bn.Error) as e:
                bn.check_public_keys(pk2, pk1)
            assert e.value.kind == bn.ErrorKind.VerificationFailed
    # src/types_test.rs
    for v in kats["public_key_from_private_key"]:
        pk = bn.PublicKey.from_private_key(bn.PrivateKey.try_from(v["private_key"]))
        assert pk.to_uncompressed().hex() == v["uncompressed"]
        assert bn.PublicKey.from_uncompressed(H(v["uncompressed"])) == pk
    g2 = bn.PublicKey(c.g2_generator())
    assert (g2 + g2).to_compressed().hex() == kats["g2_double_generator_compressed"]["hex"]
    g1 = bn.Signature(c.g1_generator())
    assert (g1 + g1).to_compressed().hex() == kats["g1_double_generator_compressed"]["hex"]
    hx = kats["private_key_roundtrip"]["hex"]
    assert bn.PrivateKey.try_from(hx).to_hex() == hx
    for bad in kats["private_key_invalid_length"]["hex"]:
        with pytest.raises(bn.Error) as e:
            bn.PrivateKey.try_from(H(bad))
        assert e.value.kind == bn.ErrorKind.InvalidLength
    # examples/bn254.rs
    ex = kats["example"]
    ks = [bn.PrivateKey.try_from(k) for k in ex["private_keys"]]
    m = ex["message"].encode()
    agg_sig = bn.ECDSA.sign(m, ks[0]) + bn.ECDSA.sign(m, ks[1])
    agg_pk = bn.PublicKey.from_private_key(ks[0]) + bn.PublicKey.from_private_key(ks[1])
    bn.ECDSA.verify(m, agg_sig, agg_pk)
    assert bn.ECDSA.batch_verify([m, m], [agg_sig, agg_sig], [agg_pk, pks[0]]) == [None, bn.Error(9)]


def test_compressed_codecs(eng, kats, derived):
    """from_compressed on the device (src/types.rs:91-93, :233-237) vs the big-integer model, incl. error codes,
    and compressed round trips through the host API (src/types_test.rs:48-54, :131-159)."""
    import bn254_amd as bn
    from oracle import bn254_model as m
    from tests.test_hostsim import _codec_cases
    g1, g2, bad_g1, bad_g2 = _codec_cases(kats, derived)
    datas = [d for hx in g1 for d in (H(hx), bytes([5 - H(hx)[0]]) + H(hx)[1:])] + [d for d, _ in bad_g1]
    out, st = eng.batch_g1_decompress(b"".join(datas), len(datas))
    for i, d in enumerate(datas):
        if i < 2 * len(g1):
            assert st[i] == 0 and out[64 * i:64 * i + 64] == m.g1_to_uncompressed(m.g1_from_compressed(d))
        else:
            assert st[i] == bad_g1[i - 2 * len(g1)][1] and out[64 * i:64 * i + 64] == bytes(64)
    datas = [d for hx in g2 for d in (H(hx), bytes([0x15 - H(hx)[0]]) + H(hx)[1:])] + [d for d, _ in bad_g2]
    out, st = eng.batch_g2_decompress(b"".join(datas), len(datas))
    for i, d in enumerate(datas):
        if i < 2 * len(g2):
            assert st[i] == 0 and out[128 * i:128 * i + 128] == m.g2_to_uncompressed(m.g2_from_compressed(d))
        else:
            assert st[i] == bad_g2[i - 2 * len(g2)][1]
    c2 = H(kats["g2_compressed_roundtrip"]["hex"])
    assert bn.PublicKey.from_compressed(c2).to_compressed() == c2
    sig = bn.Signature.from_compressed(H(kats["sign"][0]["signature_compressed"]))          # src/ecdsa_test.rs:26-28
    sk = bn.PrivateKey.try_from(kats["sign"][0]["private_key"])
    bn.ECDSA.verify(H(kats["sign"][0]["message_hex"]), sig, bn.PublicKey.from_private_key(sk))
    with pytest.raises(bn.Error) as e:
        bn.PublicKey.from_compressed(b"\x0c" + c2[1:])
    assert e.value.kind == bn.ErrorKind.InvalidEncoding


def test_pairing_check_formatters(eng, c, kats):
    """src/utils.rs:197-239 (untested upstream): the two tuples are the little-endian images of
    (H(m), pk) and (sig, -G2::one()); fed back through the pairing check they verify."""
    import bn254_amd as bn
    v = kats["sign"][0]
    msg = H(v["message_hex"])
    sk = bn.PrivateKey.try_from(v["private_key"])
    sig, pk = bn.ECDSA.sign(msg, sk), bn.PublicKey.from_private_key(sk)
    t1 = bn.format_pairing_check_values(msg, sig.to_compressed(), pk.to_compressed())
    t2 = bn.format_pairing_check_uncompressed_values(msg, sig.to_uncompressed(), pk.to_uncompressed())
    assert t1 == t2 and [len(x) for t in t1 for x in t] == [64, 128, 64, 128]
    be = lambda b: b"".join(b[i:i + 32][::-1] for i in range(0, len(b), 32))     # noqa: E731
    assert be(t1[0][0]) == c.hash_to_g1(msg)[1] and be(t1[0][1]) == pk.raw and be(t1[1][0]) == sig.raw
    g2 = c.g2_generator()
    assert be(t1[1][1]) == g2[:64] + b"".join((Q - int.from_bytes(g2[i:i + 32], "big")).to_bytes(32, "big") for i in (64, 96))
    assert eng.batch_pairing_check(be(t1[0][0]) + be(t1[1][0]), be(t1[0][1]) + be(t1[1][1]), 1, 2) == b"\x00"
    with pytest.raises(bn.Error):
        bn.format_pairing_check_uncompressed_values(msg, sig.raw[:10], pk.raw)


def test_bn256_vectors(eng, kats):
    adds = kats["g1_add"]
    out, st = eng.batch_g1_add(b"".join(H(v["x1"] + v["y1"]) for v in adds), b"".join(H(v["x2"] + v["y2"]) for v in adds), len(adds))
    assert st == bytes(len(adds)) and out.hex() == "".join(v["result"] for v in adds)
    muls = kats["g1_mul"]
    out, st = eng.batch_g1_mul(b"".join(H(v["x"] + v["y"]) for v in muls), b"".join(H(v["scalar"]) for v in muls), len(muls))
    assert st == bytes(len(muls)) and out.hex() == "".join(v["result"] for v in muls)


def test_variable_base_g1_multiplication_and_sign_through_the_endomorphism_vs_oracle(eng, c, kats):
    """round 6: bn254_batch_g1_mul with explicit points and bn254_batch_sign (sk * H(m), /root/reference/src/ecdsa.rs:26-35) run the joint
    128-step ladder over phi(x, y) = (beta x, y) (csrc/bn254_curve.h: g1_mul_glv_full).  Against the oracle: scalars around 0, r, 2^128 and
    2^256 raw and reduced (a raw scalar acts mod r: G1 has cofactor 1), scalars whose decomposition has the rare POSITIVE k2, window patterns
    that carry through every digit, random ones; the generator, other points, the identity (all-zero bytes) and an invalid point (status,
    no output); a ragged batch.  Signatures: the reference's known answer and random keys incl. keys above r."""
    from tests.test_hostsim import GLV_R as R, GLV_LAMBDA, glv_positive_k2_scalars
    rnd = random.Random(606)
    pos = glv_positive_k2_scalars(5)
    ks = [0, 1, 2, 15, 16, 17, R - 1, R, R + 1, 2 * R + 5, GLV_LAMBDA, GLV_LAMBDA + 1, R - GLV_LAMBDA, (R - 1) // 2, 2 ** 127 - 1, 2 ** 128, 2 ** 128 + 1, 2 ** 253,
          2 ** 256 - 1, 2 ** 256 - 16, int("8" * 64, 16), int("9" * 63, 16), int("7" * 64, 16)] + pos
    ks += [rnd.randrange(2 ** 256) for _ in range(120)] + [rnd.randrange(R) for _ in range(60)]
    g1 = c.g1_generator()
    bases = [g1, c.g1_mul(g1, (12345).to_bytes(32, "big")), c.g1_mul(g1, (R - 7).to_bytes(32, "big")), bytes(64)]
    pts = b"".join(bases[i % len(bases)] for i in range(len(ks)))
    n = len(ks)
    assert n % 64 != 0
    scal = b"".join(k.to_bytes(32, "big") for k in ks)
    for reduce in (False, True):
        got, st = eng.batch_g1_mul(pts, scal, n, reduce_scalar=reduce)
        assert st == bytes(n)
        for i, k in enumerate(ks):
            want = c.g1_mul(bases[i % len(bases)], ((k % R) if reduce else k).to_bytes(32, "big"))
            assert got[64 * i:64 * i + 64] == want, (hex(k), i % len(bases), reduce)
    bad = (1).to_bytes(32, "big") + (1).to_bytes(32, "big")                       # (1, 1) is not on the curve
    out, st = eng.batch_g1_mul(g1 + bad + g1, (5).to_bytes(32, "big") * 3, 3)
    assert st[0] == 0 and st[2] == 0 and st[1] != 0 and out[:64] == out[128:] == c.g1_mul(g1, (5).to_bytes(32, "big"))
    # signatures: known answer, keys above r (Fr::from_slice reduces), the positive-k2 keys
    v = kats["sign"][0]
    sig, st = eng.batch_sign([H(v["message_hex"])], H(v["private_key"]))
    assert st == b"\0" and c.g1_compress(sig).hex() == v["signature_compressed"]
    keys = [rnd.randrange(1, 2 ** 256) for _ in range(40)] + pos + [R - 1, R + 1, 1]
    msgs = [b"glv sign %d" % i for i in range(len(keys))]
    sigs, st = eng.batch_sign(msgs, b"".join(k.to_bytes(32, "big") for k in keys))
    assert st == bytes(len(keys))
    for i, k in enumerate(keys):
        assert sigs[64 * i:64 * i + 64] == c.sign(msgs[i], k.to_bytes(32, "big")), hex(k)


# ---- batches vs the oracle -----------------------------------------------------------------
def test_batch_verify_vs_oracle_4k(eng, c):
    from tests.datagen import make_verify_batch
    n = 4096 + 37                      # ragged tail wave
    msgs, sigs, pks, expected = make_verify_batch(eng, n)
    got = eng.batch_verify(msgs, sigs, pks, flags=0)
    assert got == expected
    want, _ = c.batch_verify(msgs[:512], sigs[:512 * 64], pks[:512 * 128], flags=0, nthreads=8)
    assert got[:512] == want
    # signatures produced on the GPU equal the oracle's
    for i in range(0, 64, 7):
        from tests.datagen import sk_bytes
        assert sigs[64 * i:64 * i + 64] == c.sign(msgs[i], sk_bytes(i % 256)) or expected[i] == 9


def test_fused_and_split_miller_agree(eng, derived):
    """the fused 2-pair Miller loop (default) and the one-pairing-per-lane path give the same statuses"""
    from bn254_amd.engine import OPT_SPLIT_MILLER
    cs = derived["verify_cases"]
    args = ([H(v["message_hex"]) for v in cs], b"".join(H(v["sig"]) for v in cs), b"".join(H(v["pk"]) for v in cs))
    want = [v["status"] for v in cs]
    assert list(eng.batch_verify(*args, flags=1)) == want
    eng.set_option(OPT_SPLIT_MILLER, 1)
    try:
        assert list(eng.batch_verify(*args, flags=1)) == want
    finally:
        eng.set_option(OPT_SPLIT_MILLER, 0)


def test_pair_lanes_and_single_lane_agree(eng, derived):
    """verify on lane pairs (default, bn254_pair.hip) and on one lane per verify give the same statuses"""
    from bn254_amd.engine import OPT_PAIR_LANES
    from tests.datagen import make_verify_batch
    cs = derived["verify_cases"]
    args = ([H(v["message_hex"]) for v in cs], b"".join(H(v["sig"]) for v in cs), b"".join(H(v["pk"]) for v in cs))
    want = [v["status"] for v in cs]
    big = make_verify_batch(eng, 2048 + 33)
    for mode in (1, 0):
        eng.set_option(OPT_PAIR_LANES, mode)
        try:
            assert list(eng.batch_verify(*args, flags=1)) == want, mode
            assert eng.batch_verify(big[0], big[1], big[2]) == big[3], mode
        finally:
            eng.set_option(OPT_PAIR_LANES, 1)


from tests.conftest import ws_default  # noqa: E402
NONET_DEFAULT = ws_default("NONET_MAX_BATCH_DEFAULT")
LM_DEFAULT = ws_default("LM_MAX_BATCH_DEFAULT")
TRIO_DEFAULT = ws_default("TRIO_MAX_BATCH_DEFAULT")


def test_octet_and_pair_layouts_agree_with_oracle(eng, c, derived, kats):
    """small batches run in the OCTET layout (eight lanes per verify, bn254_trio.hip; default up to 16384 items; the Miller loop
    with the four lane pairs of a verify as four waves with their own roles, or as lane groups of one wave), larger ones on
    lane pairs: all against the golden cases, the oracle on ragged sizes with faults of every class, and
    check_public_keys; the threshold itself (8192 octet, 8193 pairs) gives the same bytes on either side"""
    from bn254_amd.engine import OPT_LM_MAX_BATCH, OPT_NONET_MAX_BATCH, OPT_NONET_WIDE, OPT_TRIO_MAX_BATCH, OPT_TRIO_WAVE_ROLES
    from tests.datagen import make_verify_batch
    cs = derived["verify_cases"]
    args = ([H(v["message_hex"]) for v in cs], b"".join(H(v["sig"]) for v in cs), b"".join(H(v["pk"]) for v in cs))
    want = [v["status"] for v in cs]
    batches = []
    for n in (1, 2, 7, 9, 65, 1000 + 27):
        msgs, sigs, pks, expected = make_verify_batch(eng, n, corrupt_every=3 if n > 2 else 0)
        sigs = bytearray(sigs)
        if n >= 9:
            sigs[64 * 4:64 * 5] = bytes(64)                   # identity signature
            sigs[64 * 8 + 40] ^= 2                            # off-curve y
        if n >= 65:
            sigs[64 * 20:64 * 20 + 32] = b"\xff" * 32          # x >= q
            pks = bytearray(pks); pks[128 * 11:128 * 12] = bytes(128); pks[128 * 13 + 3] ^= 1                        # identity / off-twist keys
            pks[128 * 17:128 * 18] = H(derived["g2_not_in_subgroup"]); pks = bytes(pks)                              # on the twist, outside the order-r subgroup (flag bit 0: status 4)
        sigs = bytes(sigs)
        oracle, _ = c.batch_verify(msgs, sigs, pks, flags=1, nthreads=8)
        batches.append((msgs, sigs, pks, oracle))
    assert {0, 9} <= set(batches[-1][3]) and len(set(batches[-1][3])) >= 4
    cpk = kats["check_public_keys"]
    g2s = b"".join(c.public_key_g2(H(v["sk_g2"])) for v in cpk) * 3
    g1s = b"".join(c.public_key_g1(H(v["sk_g1"])) for v in cpk) * 3
    cpk_want = bytes(v["status"] for v in cpk) * 3
    edge = make_verify_batch(eng, 8193, corrupt_every=11)
    try:
        # the octet path's Miller loop as eight (default) / four wave roles and as lane groups of one wave; lane pairs; the default threshold
        # (the lane machine, which takes the smallest batches by default, off: these kernels at every size)
        eng.set_option(OPT_LM_MAX_BATCH, 0)
        for lim, roles in ((1 << 20, 2), (1 << 20, 1), (1 << 20, 0), (0, 2), (8192, 2)):   # 8192: the edge batch of 8193 on lane pairs, its first 8192 in octets
            eng.set_option(OPT_TRIO_MAX_BATCH, lim)
            eng.set_option(OPT_TRIO_WAVE_ROLES, roles)
            assert list(eng.batch_verify(*args, flags=1)) == want, lim
            for msgs, sigs, pks, oracle in batches:
                assert eng.batch_verify(msgs, sigs, pks, flags=1) == oracle, (lim, len(msgs))
            assert eng.batch_check_public_keys(g2s, g1s, len(cpk) * 3) == cpk_want, lim
            assert eng.batch_verify(edge[0], edge[1], edge[2]) == edge[3], lim
            assert eng.batch_verify(edge[0][:8192], edge[1][:8192 * 64], edge[2][:8192 * 128]) == edge[3][:8192], lim
        # the final exponentiation on NINE lane pairs per verify (bn254_nonet.hip; 3 verifies per wave, 12 per workgroup): forced on for
        # every small-batch size (1, 2, 7, 9, 65, 1027: not multiples of 3 or 12; 8192: several passes), and off
        eng.set_option(OPT_TRIO_MAX_BATCH, TRIO_DEFAULT)
        eng.set_option(OPT_TRIO_WAVE_ROLES, 2)
        nonet_default = NONET_DEFAULT
        for lim in (1 << 20, 0, nonet_default):
            eng.set_option(OPT_NONET_MAX_BATCH, lim)
            assert list(eng.batch_verify(*args, flags=1)) == want, ("nonet", lim)
            for msgs, sigs, pks, oracle in batches:
                assert eng.batch_verify(msgs, sigs, pks, flags=1) == oracle, ("nonet", lim, len(msgs))
            assert eng.batch_check_public_keys(g2s, g1s, len(cpk) * 3) == cpk_want, ("nonet", lim)
            assert eng.batch_verify(edge[0][:8192], edge[1][:8192 * 64], edge[2][:8192 * 128]) == edge[3][:8192], ("nonet", lim)
            assert eng.batch_verify(edge[0][:3073], edge[1][:3073 * 64], edge[2][:3073 * 128]) == edge[3][:3073], ("nonet", lim)
        # the Miller loop as the LANE MACHINE (bn254_lmiller.hip; nine lane pairs in each of four waves per verify, 3 verifies per workgroup):
        # forced on for every small-batch size (1, 2, 7, 9, 65, 1027: not multiples of 3; 8192: eleven passes; identity operands in the
        # batches and the golden cases: pair A / pair B skipped), with either final exponentiation behind it, then the default threshold on
        # both sides (1536 | 1537)
        for lim, nonet, wide in ((1 << 20, NONET_DEFAULT, 1), (1 << 20, NONET_DEFAULT, 0), (1 << 20, 0, 1), (LM_DEFAULT, NONET_DEFAULT, 1)):
            eng.set_option(OPT_LM_MAX_BATCH, lim)
            eng.set_option(OPT_NONET_MAX_BATCH, nonet)
            eng.set_option(OPT_NONET_WIDE, wide)                  # final exponentiation on eighteen lane pairs up to 1 024 verifies (default) / nine at every size
            assert list(eng.batch_verify(*args, flags=1)) == want, ("lane machine", lim)
            for msgs, sigs, pks, oracle in batches:
                assert eng.batch_verify(msgs, sigs, pks, flags=1) == oracle, ("lane machine", lim, len(msgs))
            assert eng.batch_check_public_keys(g2s, g1s, len(cpk) * 3) == cpk_want, ("lane machine", lim)
            for cut in (8192, LM_DEFAULT, LM_DEFAULT + 1, 1024, 1025):
                assert eng.batch_verify(edge[0][:cut], edge[1][:cut * 64], edge[2][:cut * 128]) == edge[3][:cut], ("lane machine", lim, cut)
    finally:
        eng.set_option(OPT_TRIO_MAX_BATCH, TRIO_DEFAULT)               # the defaults
        eng.set_option(OPT_TRIO_WAVE_ROLES, 2)
        eng.set_option(OPT_NONET_MAX_BATCH, NONET_DEFAULT)
        eng.set_option(OPT_LM_MAX_BATCH, LM_DEFAULT)
        eng.set_option(OPT_NONET_WIDE, 1)
    # the default threshold itself: 16384 verifies in two passes of the small-batch kernels, 16385 on lane pairs
    big = make_verify_batch(eng, 16385, corrupt_every=13)
    assert eng.batch_verify(big[0], big[1], big[2]) == big[3]
    assert eng.batch_verify(big[0][:16384], big[1][:16384 * 64], big[2][:16384 * 128]) == big[3][:16384]


def test_every_boundary_of_the_routing_table_vs_expected(eng, c):
    """The batch size -> layout routing is ONE table (bn254_amd/csrc/bn254_ws.h: bn_route; bn254_debug_route_table hands out the rows of this
    context): every boundary of it, on both sides, is generated FROM the table — for the defaults and for three other settings of the
    thresholds — and must give the expected status bytes (one batch, checked once against the oracle, cut at every size needed); the keyed
    verify and check_public_keys take the same table and are cut at the same sizes."""
    from bn254_amd.engine import OPT_LM_MAX_BATCH, OPT_NONET_MAX_BATCH, OPT_NONET_WIDE, OPT_TRIO_MAX_BATCH
    from tests.datagen import make_verify_batch
    big = make_verify_batch(eng, 16386, corrupt_every=13)
    oracle, _ = c.batch_verify(big[0][:3100], big[1][:3100 * 64], big[2][:3100 * 128], flags=0, nthreads=8)
    assert oracle == big[3][:3100]                          # the expected pattern IS the oracle's (checked where the oracle is quick)
    default = eng.route_table()
    assert default == [(1024, 0, 0), (LM_DEFAULT, 0, 1), (NONET_DEFAULT, 1, 1), (TRIO_DEFAULT, 1, 2), (2 ** 64 - 1, 2, 3)], default
    seen_routes = set()
    settings = ((LM_DEFAULT, NONET_DEFAULT, 1, TRIO_DEFAULT), (0, NONET_DEFAULT, 1, TRIO_DEFAULT), (2048, 1500, 0, 4096), (700, 5000, 1, 5000))
    try:
        for lm, nonet, wide, trio in settings:
            eng.set_option(OPT_LM_MAX_BATCH, lm); eng.set_option(OPT_NONET_MAX_BATCH, nonet)
            eng.set_option(OPT_NONET_WIDE, wide); eng.set_option(OPT_TRIO_MAX_BATCH, trio)
            table = eng.route_table()
            assert table[-1] == (2 ** 64 - 1, 2, 3) and [r[0] for r in table] == sorted(r[0] for r in table)
            for (max_n, miller, fe), nxt in zip(table[:-1], table[1:]):
                assert (miller, fe) != (nxt[1], nxt[2])        # a row per distinct route
                seen_routes |= {(miller, fe), (nxt[1], nxt[2])}
                for n in (max_n - 1, max_n, max_n + 1):        # both sides of the boundary, generated from the table
                    assert eng.batch_verify(big[0][:n], big[1][:n * 64], big[2][:n * 128]) == big[3][:n], ((lm, nonet, wide, trio), n)
    finally:
        eng.set_option(OPT_LM_MAX_BATCH, LM_DEFAULT); eng.set_option(OPT_NONET_MAX_BATCH, NONET_DEFAULT)
        eng.set_option(OPT_NONET_WIDE, 1); eng.set_option(OPT_TRIO_MAX_BATCH, TRIO_DEFAULT)
    assert seen_routes >= {(0, 0), (0, 1), (1, 1), (1, 2), (2, 3), (0, 2)}, seen_routes
    assert eng.route_table() == default


def test_oversized_batches_are_sliced_inside_the_library(eng, c, derived):
    """A batch whose workspace does not fit is cut into slices inside the library instead of failing with an out-of-memory error: forced
    by BN254_OPT_MAX_CHUNK (a cap that is no multiple of anything) and by the automatic rule priced against a pretended 3 MB of free device
    memory (BN254_OPT_ASSUME_FREE_MB); host and device entry points of verify, verify from compressed encodings and keyed verify; ragged
    messages with faults of every class — the statuses are those of the oracle, i.e. of the one-piece call."""
    import ctypes
    import torch
    from bn254_amd.engine import OPT_ASSUME_FREE_MB, OPT_MAX_CHUNK
    from tests.test_mgpu import _faulty_batch
    n = 4099
    msgs, sigs, pks = _faulty_batch(eng, derived, n, 4242)
    want, _ = c.batch_verify(msgs, sigs, pks, flags=1, nthreads=8)
    assert len(set(want)) >= 4
    dev = torch.device("cuda", 0)
    blob = b"".join(msgs)
    offs = [0]
    for m in msgs:
        offs.append(offs[-1] + len(m))
    d_msgs = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(dev)
    d_off = torch.tensor(offs, dtype=torch.int64, device=dev)
    d_sigs = torch.frombuffer(bytearray(sigs), dtype=torch.uint8).to(dev)
    d_pks = torch.frombuffer(bytearray(pks), dtype=torch.uint8).to(dev)
    fresh = __import__("bn254_amd").Engine(0)                 # its own context: the workspace starts EMPTY, so the automatic rule has something to decide
    try:
        for opt, val in ((OPT_MAX_CHUNK, 1000), (OPT_MAX_CHUNK, 4098), (OPT_ASSUME_FREE_MB, 3), (OPT_MAX_CHUNK, 0)):
            fresh.set_option(OPT_MAX_CHUNK, 0); fresh.set_option(OPT_ASSUME_FREE_MB, 0)
            fresh.set_option(opt, val)
            assert fresh.batch_verify(msgs, sigs, pks, flags=1) == want, (opt, val)
            d_st = torch.full((n,), 0xEE, dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()
            fresh.batch_verify_device(d_msgs.data_ptr(), d_off.data_ptr(), d_sigs.data_ptr(), d_pks.data_ptr(), n, d_st.data_ptr(), flags=1)
            fresh.synchronize()
            assert bytes(d_st.cpu().numpy()) == want, (opt, val)
        # 3 MB priced against 811 B per item: slices of 2 816 items (the largest multiple of 256 that fits 80 % of it) — the workspace never grew beyond
        fresh2 = __import__("bn254_amd").Engine(0)
        fresh2.set_option(OPT_ASSUME_FREE_MB, 3)
        assert fresh2.batch_verify(msgs, sigs, pks, flags=1) == want
        # compressed encodings and registered keys through the same rule
        good = [i for i in range(n) if want[i] in (0, 9) and sigs[64 * i:64 * i + 64] != bytes(64) and pks[128 * i:128 * i + 128] != bytes(128)][:1500]
        gm = [msgs[i] for i in good]
        gs = b"".join(sigs[64 * i:64 * i + 64] for i in good)
        gp = b"".join(pks[128 * i:128 * i + 128] for i in good)
        gw = bytes(want[i] for i in good)
        from bn254_amd.api import PublicKey, Signature
        s33 = b"".join(Signature(gs[64 * i:64 * i + 64]).to_compressed() for i in range(len(good)))       # byte logic only (utils.rs:84-104, :130-158)
        p65 = b"".join(PublicKey(gp[128 * i:128 * i + 128]).to_compressed() for i in range(len(good)))
        fresh.set_option(OPT_MAX_CHUNK, 0); fresh.set_option(OPT_ASSUME_FREE_MB, 0)
        one_piece = fresh.batch_verify_compressed(gm, s33, p65)
        assert one_piece == gw
        fresh.set_option(OPT_MAX_CHUNK, 333)
        assert fresh.batch_verify_compressed(gm, s33, p65) == gw
        keys = sorted(set(gp[128 * i:128 * i + 128] for i in range(len(good))))
        st = fresh.register_keys(b"".join(keys))
        assert st == bytes(len(keys))
        idx = [keys.index(gp[128 * i:128 * i + 128]) for i in range(len(good))]
        assert fresh.batch_verify_keyed(gm, gs, idx) == gw
        fresh.set_option(OPT_MAX_CHUNK, 0)
        assert fresh.batch_verify_keyed(gm, gs, idx) == gw
    finally:
        fresh.close()


def test_malformed_inputs_fuzz_vs_oracle(eng, c, derived):
    """3000 verifies whose signature / public key bytes are valid, mutated (bit flips, coordinate >= q,
    swapped coordinates, zeros) or random: every status byte must equal the oracle's, with and without the
    G2 subgroup flag."""
    from tests.datagen import make_verify_batch
    rnd = random.Random(2024)
    n = 3000
    msgs, sigs, pks, _ = make_verify_batch(eng, n, corrupt_every=0)
    sigs, pks = bytearray(sigs), bytearray(pks)
    off_sub = H(derived["g2_not_in_subgroup"])
    for i in range(n):
        kind = rnd.randrange(12)
        s, p = memoryview(sigs)[64 * i:64 * i + 64], memoryview(pks)[128 * i:128 * i + 128]
        if kind == 0:
            s[rnd.randrange(64)] ^= 1 << rnd.randrange(8)
        elif kind == 1:
            p[rnd.randrange(128)] ^= 1 << rnd.randrange(8)
        elif kind == 2:
            s[:32] = (Q + rnd.randrange(1000)).to_bytes(32, "big")
        elif kind == 3:
            j = 32 * rnd.randrange(4)
            p[j:j + 32] = (Q + rnd.randrange(1 << 200)).to_bytes(32, "big")
        elif kind == 4:
            s[:] = bytes(64)
        elif kind == 5:
            p[:] = bytes(128)
        elif kind == 6:
            s[:] = bytes(rnd.randrange(256) for _ in range(64))
        elif kind == 7:
            p[:] = bytes(rnd.randrange(256) for _ in range(128))
        elif kind == 8:
            s[:32], s[32:] = bytes(s[32:]), bytes(s[:32])
        elif kind == 9:
            p[:] = off_sub
        # kinds 10, 11: left valid
    sigs, pks = bytes(sigs), bytes(pks)
    for flags in (0, 1, 3):
        got = eng.batch_verify(msgs, sigs, pks, flags=flags)
        want, _ = c.batch_verify(msgs, sigs, pks, flags=flags, nthreads=8)
        bad = [i for i in range(n) if got[i] != want[i]]
        assert not bad, (flags, bad[:5], [(got[i], want[i]) for i in bad[:5]])
        if flags == 0:
            assert set(got) >= {0, 4, 6, 9}    # the corpus really exercises several status codes


def test_pairing_check_k_pairs(eng, c):
    ps, qs = _rand_points(c, 6, b"kp")
    # e(aP, Q) * e(-aP, Q) == 1  and a 3-pair product that is not one
    neg = lambda p: p[:32] + ((Q - int.from_bytes(p[32:], "big")) % Q).to_bytes(32, "big")   # noqa: E731
    g1 = ps[0] + neg(ps[0]) + ps[1] + ps[2] + ps[3] + ps[4]
    g2 = qs[0] + qs[0] + qs[1] + qs[2] + qs[3] + qs[4]
    st = eng.batch_pairing_check(g1[:128] + g1[128:256], g2[:256] + g2[256:512], 2, 2)
    assert st == bytes([0, 9])
    assert st[0] == c.pairing_check(g1[:128], g2[:256], 2) and st[1] == c.pairing_check(g1[128:256], g2[256:512], 2)
    gt, _ = eng.batch_pairing(g1[128:320], g2[256:640], 1, 3)
    assert gt == c.pairing(g1[128:320], g2[256:640], k=3)


def test_pairing_small_batches_take_the_small_batch_kernels(eng, c, derived):
    """bn254_batch_pairing / _check for batches that cannot fill the chip (n k <= 1 536 pairs, n <= 1 024 items): the lane machine with the
    fixed pair skipped + the final exponentiation (exact program) on eighteen lane pairs — canonical Gt bytes and statuses equal the
    lane-pair path's (BN254_OPT_LM_MAX_BATCH = 0) and the oracle's, for k = 1, 2, 3 pairs per item, sizes that are no multiples of the
    kernels' 3 / 4 items per workgroup, identity operands (the pair contributes one), an undecodable point (its status, Gt untouched
    semantics as on lane pairs), and on both sides of the routing limits"""
    from bn254_amd.engine import OPT_LM_MAX_BATCH
    ps, qs = _rand_points(c, 23, b"sbp")
    neg = lambda p: p[:32] + ((Q - int.from_bytes(p[32:], "big")) % Q).to_bytes(32, "big")   # noqa: E731
    for n, k in ((1, 1), (7, 1), (5, 2), (3, 3), (23, 1), (11, 2)):
        g1 = bytearray(b"".join(ps[(i * 3 + j) % 23] for i in range(n) for j in range(k)))
        g2 = bytearray(b"".join(qs[(i * 5 + 2 * j) % 23] for i in range(n) for j in range(k)))
        if n >= 5:
            g1[64 * (2 * k):64 * (2 * k) + 64] = bytes(64)                     # identity G1 operand in item 2
            g2[128 * (3 * k):128 * (3 * k) + 128] = bytes(128)                 # identity G2 operand in item 3
            g1[64 * (4 * k) + 63] ^= 1                                         # off-curve point in item 4 -> its status
        if k == 2 and n >= 2:                                                   # item 1: e(P, Q) e(-P, Q) = 1
            g1[64 * 2:64 * 4] = ps[0] + neg(ps[0]); g2[128 * 2:128 * 4] = qs[0] + qs[0]
        g1, g2 = bytes(g1), bytes(g2)
        eng.set_option(OPT_LM_MAX_BATCH, 0)
        try:
            gt_pair, st_pair = eng.batch_pairing(g1, g2, n, k)
            ck_pair = eng.batch_pairing_check(g1, g2, n, k)
        finally:
            eng.set_option(OPT_LM_MAX_BATCH, LM_DEFAULT)
        gt, st = eng.batch_pairing(g1, g2, n, k)
        assert st == st_pair and eng.batch_pairing_check(g1, g2, n, k) == ck_pair == st, (n, k, st, st_pair)
        for i in range(n):
            if st[i] in (0, 9):
                assert gt[384 * i:384 * i + 384] == gt_pair[384 * i:384 * i + 384], (n, k, i)
                assert gt[384 * i:384 * i + 384] == c.pairing(g1[64 * k * i:64 * k * (i + 1)], g2[128 * k * i:128 * k * (i + 1)], k=k), (n, k, i)
        if n >= 5:
            assert st[4] == 4 and gt[384 * 2:384 * 3] == gt_pair[384 * 2:384 * 3]
        if k == 2 and n >= 2:
            assert st[1] == 0 and gt[384:768].hex() == derived["gt_one"]
    # the routing limits: 1 024 items of one pair (small-batch kernels) and 1 025 (lane pairs) give the same bytes on the common prefix
    n = 1025
    g1 = b"".join(ps[i % 23] for i in range(n)); g2 = b"".join(qs[(7 * i) % 23] for i in range(n))
    gt_a, st_a = eng.batch_pairing(g1[:64 * 1024], g2[:128 * 1024], 1024)
    gt_b, st_b = eng.batch_pairing(g1, g2, n)
    assert gt_a == gt_b[:384 * 1024] and st_a == st_b[:1024]


def test_key_derivation_from_the_comb_table_vs_oracle_and_ladder(eng, c, kats):
    """sk * G2::one() (PublicKey::from_private_key, /root/reference/src/types.rs:85-87) through the fixed-base comb of round 6 (65 table additions on a
    lane pair, window entries found by constant-time scans, blinded accumulator) against the oracle's scalar multiplication AND the general
    256-step ladder it replaces (BN254_OPT_G2_FIXED_BASE = 0): scalars that put every digit pattern on the table's edges — 0, 1, 7, 8, 9, 15,
    16, 2^k and 2^k - 1, all-8 / all-9 nibbles (every window carries), r - 1, r, r + 1, 2^256 - 1 (raw: used as they are; reduced: Fr::from_slice),
    random ones, a batch that is no multiple of the workgroup, and the four sk * G2 known answers of the reference (src/types_test.rs:71-129)."""
    from bn254_amd.engine import OPT_G2_FIXED_BASE
    R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
    rnd = random.Random(99)
    vals = [0, 1, 7, 8, 9, 15, 16, 17, 2 ** 64, 2 ** 64 - 1, 2 ** 128 + 1, 2 ** 252, 2 ** 253 - 1, int("8" * 64, 16), int("9" * 64, 16) % 2 ** 256, int("7" * 64, 16),
            R - 1, R, R + 1, 2 * R, 2 ** 256 - 1, 2 ** 256 - 2 ** 4]
    # ADVERSARIAL scalars: the blinding scalar is a public constant of the library (comb_build, csrc/bn254_group.hip), so scalars exist that
    # walk the blinded accumulator INTO the exceptional route of the in-place addition the blinding is there to avoid: accumulator == +entry
    # (a doubling) or == -entry (the identity) at window 62 / 63, accumulator == -B at the final subtraction of the blinding point, and for
    # G1 the two lanes' partial sums being each other's negative.  The derivation is restated here from the table's definition.
    b0 = 0x2b67ae85a54ff53a3c6ef3721f83d9ab5be0cd199c4d21a70f3a7e556b2f1c9d

    def comb_hits(k):
        acc, hits, carry = b0, [], 0
        for j in range(65):
            v = ((k >> (4 * j)) & 15 if j < 64 else 0) + carry
            carry = 1 if v > 8 else 0
            d = v - 16 * carry
            e = d * 16 ** j % R
            if d and acc == e:
                hits.append("doubling")
            if d and (acc + e) % R == 0:
                hits.append("identity")
            acc = (acc + e) % R
        return hits

    adversarial = []
    for j in (62, 63):
        for nib in range(16):
            for cin in (0, 1):
                for sgn in (1, -1):
                    v = nib + cin
                    d = v - 16 * (v > 8)
                    low = (sgn * d * 16 ** j + cin * 16 ** j - b0) % R
                    k = low + nib * 16 ** j
                    if d and low < 16 ** j and k < 2 ** 256 and comb_hits(k) and k not in adversarial:
                        adversarial.append(k)
    assert len(adversarial) >= 8 and {h for k in adversarial for h in comb_hits(k)} == {"doubling", "identity"}
    vals += adversarial + [(-2 * b0) % R, (-4 * b0) % R, (-b0) % R, b0, (2 * b0) % R]
    vals += [rnd.randrange(2 ** 256) for _ in range(175)] + [rnd.randrange(R) for _ in range(70)]
    n = len(vals)
    if n % 128 == 0:
        vals.append(3)
        n += 1
    scal = b"".join(v.to_bytes(32, "big") for v in vals)
    g2 = c.g2_generator()
    for reduce in (False, True):
        want = b"".join(c.g2_mul(g2, ((v % R) if reduce else v).to_bytes(32, "big")) for v in vals)
        eng.set_option(OPT_G2_FIXED_BASE, 1)
        got, st = eng.batch_g2_mul(None, scal, n, reduce_scalar=reduce)
        eng.set_option(OPT_G2_FIXED_BASE, 0)
        ladder, st_l = eng.batch_g2_mul(None, scal, n, reduce_scalar=reduce)
        eng.set_option(OPT_G2_FIXED_BASE, 1)
        assert st == st_l == bytes(n)
        bad = [i for i in range(n) if got[128 * i:128 * i + 128] != want[128 * i:128 * i + 128]]
        assert not bad, (reduce, [hex(vals[i]) for i in bad[:4]])
        assert ladder == want
        assert got[:128] == bytes(128)                       # 0 * G = the identity = all-zero bytes
        # the same scalars on G1::one() (PublicKeyG1::from_private_key, src/types.rs:155-157): comb (points = None), the general ladder on
        # the explicit generator, the comb switched off, and the oracle
        g1 = c.g1_generator()
        want1 = b"".join(c.g1_mul(g1, ((v % R) if reduce else v).to_bytes(32, "big")) for v in vals)
        got1, st1 = eng.batch_g1_mul(None, scal, n, reduce_scalar=reduce)
        lad1, st1l = eng.batch_g1_mul(g1 * n, scal, n, reduce_scalar=reduce)
        eng.set_option(OPT_G2_FIXED_BASE, 0)
        off1, st1o = eng.batch_g1_mul(None, scal, n, reduce_scalar=reduce)
        eng.set_option(OPT_G2_FIXED_BASE, 1)
        assert st1 == st1l == st1o == bytes(n)
        bad = [i for i in range(n) if got1[64 * i:64 * i + 64] != want1[64 * i:64 * i + 64]]
        assert not bad, ("G1", reduce, [hex(vals[i]) for i in bad[:4]])
        assert lad1 == want1 and off1 == want1
    # one key, and the reference's own known answers, through the mirror of the reference API
    one, st = eng.batch_g2_mul(None, (5).to_bytes(32, "big"), 1)
    assert one == c.g2_mul(g2, (5).to_bytes(32, "big")) and st == b"\0"
    for v in kats["public_key_from_private_key"]:
        out, st = eng.batch_g2_mul(None, H(v["private_key"]), 1, reduce_scalar=True)
        assert out == H(v["uncompressed"]) and st == b"\0"


def test_group_ops_vs_oracle(eng, c):
    ps, qs = _rand_points(c, 40, b"grp")
    n = 40
    out, st = eng.batch_g2_add(b"".join(qs), b"".join(qs[::-1]), n)
    assert st == bytes(n)
    for i in range(n):
        assert out[128 * i:128 * i + 128] == c.g2_add(qs[i], qs[n - 1 - i])
    ks = [hashlib.sha256(b"k%d" % i).digest() for i in range(n)]
    out, st = eng.batch_g2_mul(b"".join(qs), b"".join(ks), n)
    for i in range(0, n, 5):
        assert out[128 * i:128 * i + 128] == c.g2_mul(qs[i], ks[i])
    # segmented sums (aggregation)
    seg = [0, 1, 1, 8, 40]
    out, st = eng.batch_g1_sum(b"".join(ps), seg)
    acc = bytes(64)
    for p in ps[8:40]:
        acc = c.g1_add(acc, p)
    assert out[192:256] == acc and out[64:128] == bytes(64) and out[:64] == ps[0]
    out2, _ = eng.batch_g2_sum(b"".join(qs), seg)
    acc = bytes(128)
    for q in qs[1:8]:
        acc = c.g2_add(acc, q)
    assert out2[256:384] == acc
    # the rare cases of the running sum: P + P, P + (-P), identity terms — in lanes next to ordinary sums
    neg = lambda p: p[:32] + ((Q - int.from_bytes(p[32:], "big")) % Q).to_bytes(32, "big")    # noqa: E731
    pts = [ps[0], ps[0], ps[1], neg(ps[1]), bytes(64), ps[2], ps[3], ps[3], ps[3]]
    out, st = eng.batch_g1_sum(b"".join(pts), [0, 2, 4, 6, 9])
    assert st == bytes(4)
    assert out[:64] == c.g1_add(ps[0], ps[0]) and out[64:128] == bytes(64) and out[128:192] == ps[2]
    assert out[192:256] == c.g1_add(c.g1_add(ps[3], ps[3]), ps[3])
    qneg = lambda q: q[:64] + b"".join(((Q - int.from_bytes(q[64 + 32 * k:96 + 32 * k], "big")) % Q).to_bytes(32, "big") for k in range(2))  # noqa: E731
    qpts = [qs[0], qs[0], qs[1], qneg(qs[1]), qs[2]]
    out2, st = eng.batch_g2_sum(b"".join(qpts), [0, 2, 4, 5])
    assert st == bytes(3) and out2[:128] == c.g2_add(qs[0], qs[0]) and out2[128:256] == bytes(128) and out2[256:384] == qs[2]


def test_aggregate_verify_vs_oracle(eng, c):
    """config-3 shape in miniature: M messages, S signers, per-tuple signer subsets over shared pools."""
    from tests.datagen import sk_bytes
    M, S = 3, 9
    msgs = [b"agg-msg-%d" % m for m in range(M)]
    sks = [sk_bytes(100 + s) for s in range(S)]
    pk_pool, st = eng.batch_g2_mul(None, b"".join(sks), S, reduce_scalar=True)
    assert st == bytes(S)
    sig_pool, st = eng.batch_sign([msgs[m] for m in range(M) for _ in range(S)], b"".join(sks * M))
    assert st == bytes(M * S)
    pk = lambda s: pk_pool[128 * s:128 * s + 128]                          # noqa: E731
    sg = lambda m, s: sig_pool[64 * (m * S + s):64 * (m * S + s) + 64]     # noqa: E731
    tuples = [(0, [0]), (1, list(range(S))), (2, [1, 3, 5, 7]), (0, []), (1, [4, 4]), (2, [8, 0, 2]), (0, [2, 99]), (1, [6])]
    tuples += [(m % M, [s for s in range(S) if (m * 37 + s * 11) % 3]) for m in range(70)]
    got = eng.batch_aggregate_verify(msgs, pk_pool, sig_pool, [t[0] for t in tuples], [t[1] for t in tuples])
    # 78 tuples: the pairing part runs on the small-batch kernels (lane machine, eighteen lane pairs); the same bytes from the eight wave roles
    # (lane machine off) and from the lane-pair kernels (small-batch kernels off)
    from bn254_amd.engine import OPT_LM_MAX_BATCH, OPT_TRIO_MAX_BATCH
    for opt, off, dflt in ((OPT_LM_MAX_BATCH, 0, LM_DEFAULT), (OPT_TRIO_MAX_BATCH, 0, TRIO_DEFAULT)):
        eng.set_option(opt, off)
        try:
            assert eng.batch_aggregate_verify(msgs, pk_pool, sig_pool, [t[0] for t in tuples], [t[1] for t in tuples]) == got
        finally:
            eng.set_option(opt, dflt)
    for i, (m, lst) in enumerate(tuples):
        if any(s >= S for s in lst):
            assert got[i] == 2
            continue
        asig, apk = bytes(64), bytes(128)
        for s in lst:
            asig, apk = c.g1_add(asig, sg(m, s)), c.g2_add(apk, pk(s))
        assert got[i] == c.verify(msgs[m], asig, apk, 0) == 0, (i, m, lst)
    # a tuple whose signatures come from another message must fail: swap the pools' message rows
    bad_pool = sig_pool[64 * S:128 * S] + sig_pool[:64 * S] + sig_pool[128 * S:]
    got = eng.batch_aggregate_verify(msgs, pk_pool, bad_pool, [0, 2], [[1, 2], [1, 2]])
    assert got == bytes([9, 0])
    # an undecodable pool entry poisons exactly the tuples that use it
    broken = bytearray(pk_pool); broken[128 * 3 + 127] ^= 1
    got = eng.batch_aggregate_verify(msgs, bytes(broken), sig_pool, [0, 0], [[3, 4], [4, 5]])
    assert got == bytes([4, 0])


def test_aggregate_verify_subset_sum_table_vs_oracle(eng, c):
    """the subset-sum route of the aggregate kernel (k_pool_subsets_g2 + mask bytes in LDS, forced on by
    BN254_OPT_AGG_SUBSET_MIN_TUPLES = 1) gives the oracle's statuses: dense lists (longer than the number of groups), a pool
    size that is not a multiple of 8, lists in descending and shuffled order, a signer named twice / three times (direct
    route for that tuple), an out-of-range signer, an identity and an undecodable pool entry, empty lists, a wave with only
    short lists (direct route), and the same inputs with the table switched off."""
    import random
    from bn254_amd.engine import OPT_AGG_SUBSET_MIN_TUPLES
    from tests.datagen import sk_bytes
    rnd = random.Random(77)
    M, S = 3, 43                                                           # 6 groups, the last one with 3 keys
    msgs = [b"sub-msg-%d" % m for m in range(M)]
    sks = [sk_bytes(700 + s) for s in range(S)]
    pk_pool, st = eng.batch_g2_mul(None, b"".join(sks), S, reduce_scalar=True)
    sig_pool, st2 = eng.batch_sign([msgs[m] for m in range(M) for _ in range(S)], b"".join(sks * M))
    assert st == bytes(S) and st2 == bytes(M * S)
    pk_pool, sig_pool = bytearray(pk_pool), bytearray(sig_pool)
    pk_pool[128 * 11:128 * 12] = bytes(128)                                 # identity key 11 ...
    for m in range(M):
        sig_pool[64 * (m * S + 11):64 * (m * S + 12)] = bytes(64)           # ... with identity signatures: tuples using it still verify
    pk_pool[128 * 20 + 127] ^= 1                                            # key 20 does not decode -> status 4 for its tuples
    pk_pool, sig_pool = bytes(pk_pool), bytes(sig_pool)
    tuples = []
    for i in range(150):
        k = rnd.choice([0, 1, 5, 7, 12, 25, 40, S])
        lst = rnd.sample(range(S), k)
        if i % 7 == 0:
            lst = sorted(lst, reverse=True)
        if i % 13 == 5 and lst:
            lst = lst + [lst[0]]                                             # a signer twice
        if i % 29 == 9 and lst:
            lst = [lst[-1]] * 3 + lst                                        # ... and four times
        if i % 31 == 3:
            lst = lst + [S + 2]                                              # out of range
        tuples.append((rnd.randrange(M), lst))
    tuples += [(1, [2, 3])] * 70                                             # a whole wave of short lists at the end
    off, flat = [0], []
    for _, lst in tuples:
        flat += lst
        off.append(len(flat))
    want = c.batch_aggregate_verify(msgs, pk_pool, sig_pool, [t[0] for t in tuples], off, flat)
    assert {0, 2, 4} <= set(want)
    for knob in (1, 0):
        eng.set_option(OPT_AGG_SUBSET_MIN_TUPLES, knob)
        got = eng.batch_aggregate_verify(msgs, pk_pool, sig_pool, [t[0] for t in tuples], [t[1] for t in tuples])
        diff = [(i, got[i], want[i], tuples[i]) for i in range(len(tuples)) if got[i] != want[i]]
        assert not diff, (knob, diff[:5])
    # REGISTERED POOLS (bn254_ctx_register_pools): the same pools decoded, hashed and tabulated once, then only tuples per call — the
    # oracle's statuses for the whole list and for slices of it in another order (calls back to back on the same tables), with tables
    # chosen for a large batch (subset sums on), for a tiny one (none), and with the widened tables forced on; a raw-pool call in between
    # replaces the tables and the registered call says so instead of reading them
    import pytest as _pytest
    from bn254_amd.engine import NativeError, OPT_AGG_WIDE_MIN_TUPLES
    try:
        for expect, wide_min in ((1 << 20, ws_default("AGG_WIDE_MIN_TUPLES_DEFAULT")), (1, ws_default("AGG_WIDE_MIN_TUPLES_DEFAULT")), (1 << 20, 1)):
            eng.set_option(OPT_AGG_SUBSET_MIN_TUPLES, 1 if expect > 1 else ws_default("AGG_SUBSET_MIN_TUPLES_DEFAULT"))
            eng.set_option(OPT_AGG_WIDE_MIN_TUPLES, wide_min)
            eng.register_pools(msgs, pk_pool, sig_pool, expect_tuples=expect)
            got = eng.batch_aggregate_verify_registered([t[0] for t in tuples], [t[1] for t in tuples])
            diff = [(i, got[i], want[i], tuples[i]) for i in range(len(tuples)) if got[i] != want[i]]
            assert not diff, ("registered", expect, wide_min, diff[:5])
            order = list(range(len(tuples)))
            rnd.shuffle(order)
            for part in (order[:64], order[64:], order[:1], []):
                got = eng.batch_aggregate_verify_registered([tuples[i][0] for i in part], [tuples[i][1] for i in part])
                assert got == bytes(want[i] for i in part), ("registered slice", expect, len(part))
        assert eng.batch_aggregate_verify(msgs, pk_pool, sig_pool, [0], [[1, 2]]) == bytes([want_small := c.batch_aggregate_verify(msgs, pk_pool, sig_pool, [0], [0, 2], [1, 2])[0]])
        with _pytest.raises(NativeError):
            eng.batch_aggregate_verify_registered([0], [[1, 2]])          # raw pools have replaced the tables: refused, not followed
        eng.register_pools(msgs, pk_pool, sig_pool, expect_tuples=64)
        assert eng.batch_aggregate_verify_registered([0], [[1, 2]]) == bytes([want_small])
    finally:
        eng.set_option(OPT_AGG_SUBSET_MIN_TUPLES, ws_default("AGG_SUBSET_MIN_TUPLES_DEFAULT"))
        eng.set_option(OPT_AGG_WIDE_MIN_TUPLES, ws_default("AGG_WIDE_MIN_TUPLES_DEFAULT"))


def test_aggregate_verify_widened_tables_vs_oracle(eng, c):
    """BN254_OPT_AGG_WIDE_MIN_TUPLES: the subset-sum tables widened once more (keys: 16 signers per entry, k_pool_widen_g2; signatures per
    message: 8, k_pool_widen_g1 — one batched AFFINE addition per entry, eight denominators per inversion) and the aggregation loop over
    16-signer chunks give the oracle's statuses (aggregation = `Add`, /root/reference/src/types.rs:126-132, :264-270).  The pools are
    built to hit the builders' exceptional paths: a key that appears twice and a key next to its NEGATIVE inside one 8-signer group and
    across the two groups of a chunk (x_B = x_A: the doubling / the identity by the complete formula), identity and undecodable entries,
    a pool size that leaves the last chunk with one group only and that group partly empty; all four combinations of wide / narrow
    key and signature tables, each against the same oracle result."""
    import random
    from bn254_amd.engine import OPT_AGG_SUBSET_MIN_TUPLES, OPT_AGG_WIDE_MIN_TUPLES, OPT_AGG_SORT_BY_MSG
    from tests.datagen import sk_bytes
    Qm = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
    rnd = random.Random(4242)
    M, S = 2, 43                                                           # 6 groups of 8 -> 3 chunks of 16; the last group holds 3 keys
    msgs = [b"wide-msg-%d" % m for m in range(M)]
    sks = [sk_bytes(900 + s) for s in range(S)]
    sks[5] = sks[2]                                                        # signer 5 IS signer 2 (same group of 8)
    sks[17] = sks[9]                                                       # signer 17 IS signer 9 (the two groups of chunk 0 ... no: 9 is group 1, 17 group 2)
    sks[12] = sks[3]                                                       # signer 12 (group 1) IS signer 3 (group 0): the two halves of chunk 0
    pk_pool, st = eng.batch_g2_mul(None, b"".join(sks), S, reduce_scalar=True)
    sig_pool, st2 = eng.batch_sign([msgs[m] for m in range(M) for _ in range(S)], b"".join(sks * M))
    assert st == bytes(S) and st2 == bytes(M * S)
    pk_pool, sig_pool = bytearray(pk_pool), bytearray(sig_pool)

    def neg_g1(p):
        return p[:32] + ((Qm - int.from_bytes(p[32:], "big")) % Qm).to_bytes(32, "big")

    def neg_g2(p):
        return p[:64] + b"".join(((Qm - int.from_bytes(p[64 + 32 * k:96 + 32 * k], "big")) % Qm).to_bytes(32, "big") for k in range(2))
    # signer 30 = MINUS signer 27 (same group 3), signer 38 = MINUS signer 24 (groups 4 and 3: different chunks), signer 41 = MINUS signer 33
    # (groups 5 and 4: the two halves of chunk 2): subsets holding both cancel to the identity inside the tables
    for a, b in ((30, 27), (38, 24), (41, 33)):
        pk_pool[128 * a:128 * a + 128] = neg_g2(bytes(pk_pool[128 * b:128 * b + 128]))
        for m in range(M):
            sig_pool[64 * (m * S + a):64 * (m * S + a) + 64] = neg_g1(bytes(sig_pool[64 * (m * S + b):64 * (m * S + b) + 64]))
    pk_pool[128 * 11:128 * 12] = bytes(128)                                 # identity key 11 with identity signatures
    for m in range(M):
        sig_pool[64 * (m * S + 11):64 * (m * S + 12)] = bytes(64)
    pk_pool[128 * 20 + 127] ^= 1                                            # key 20 does not decode -> status 4 for its tuples
    pk_pool, sig_pool = bytes(pk_pool), bytes(sig_pool)
    n = 512 * M + 77                                                        # >= 512 tuples per message: the per-message 8-signer tables qualify
    tuples = []
    for i in range(n):
        kk = rnd.choice([0, 1, 7, 12, 25, 40, S, S])
        lst = rnd.sample(range(S), kk)
        if i % 7 == 0:
            lst = sorted(lst, reverse=True)
        if i % 11 == 1:
            lst = sorted(set(lst) | {2, 5, 3, 12, 27, 30})                   # the equal and the opposite pairs together
        if i % 13 == 5 and lst:
            lst = lst + [lst[0]]                                             # a signer twice: direct route for that tuple
        if i % 31 == 3:
            lst = lst + [S + 2]                                              # out of range
        if i % 37 == 8:
            lst = [24, 38] + [s for s in lst if s not in (24, 38)]           # cancels across chunks: in the running sum, not in a table
        tuples.append((rnd.randrange(M), lst))
    off, flat = [0], []
    for _, lst in tuples:
        flat += lst
        off.append(len(flat))
    want = c.batch_aggregate_verify(msgs, pk_pool, sig_pool, [t[0] for t in tuples], off, flat, nthreads=8)
    assert {0, 2, 4} <= set(want)
    e2 = __import__("bn254_amd").Engine(0)                                   # a context of its own: the option changes stay local
    e2.set_option(OPT_AGG_SUBSET_MIN_TUPLES, 1)
    for wide, sort in ((1, 1), (1, 0), (0, 1)):
        e2.set_option(OPT_AGG_WIDE_MIN_TUPLES, wide)
        e2.set_option(OPT_AGG_SORT_BY_MSG, sort)
        got = e2.batch_aggregate_verify(msgs, pk_pool, sig_pool, [t[0] for t in tuples], [t[1] for t in tuples])
        diff = [(i, got[i], want[i], tuples[i]) for i in range(n) if got[i] != want[i]]
        assert not diff, (wide, sort, diff[:5])
    # wide keys with NARROW signature tables (fewer than 512 tuples per message), and 64 tuples (no signature tables at all)
    e2.set_option(OPT_AGG_WIDE_MIN_TUPLES, 1)
    for cut in (300, 64):
        got = e2.batch_aggregate_verify(msgs, pk_pool, sig_pool, [t[0] for t in tuples[:cut]], [t[1] for t in tuples[:cut]])
        assert got == want[:cut], cut
    # pools of ONE group of 8 (a chunk without a second half), of exactly two groups, and of 17 signers (three groups: the last chunk is half empty)
    for S2 in (5, 16, 17):
        sks2 = [sk_bytes(950 + s) for s in range(S2)]
        pk2, _ = eng.batch_g2_mul(None, b"".join(sks2), S2, reduce_scalar=True)
        sg2, _ = eng.batch_sign([msgs[0]] * S2, b"".join(sks2))
        tl = [(0, rnd.sample(range(S2), rnd.randrange(0, S2 + 1))) for _ in range(560)]
        off2, flat2 = [0], []
        for _, lst in tl:
            flat2 += lst
            off2.append(len(flat2))
        want2 = c.batch_aggregate_verify(msgs[:1], pk2, sg2, [0] * len(tl), off2, flat2, nthreads=8)
        got2 = e2.batch_aggregate_verify(msgs[:1], pk2, sg2, [0] * len(tl), [t[1] for t in tl])
        assert got2 == want2, S2
    e2.close()


def test_aggregate_verify_bucketed_by_message_vs_oracle(eng, c):
    """BN254_OPT_AGG_SORT_BY_MSG: with the per-message signature tables in use the tuples are bucketed by message on the device
    (counting sort into an index map, XCD-contiguous slots) before the aggregation kernel.  Statuses must land at the tuples' OWN
    indices: a batch whose tuple_msg is random, the same batch sorted by message, and one with every tuple on one message — each with
    the sort on and off — against the oracle; message indices out of range (their own bucket), dense / short / duplicate / empty
    signer lists, a tuple count that is not a multiple of the workgroup size."""
    import random
    from bn254_amd.engine import OPT_AGG_SUBSET_MIN_TUPLES, OPT_AGG_SORT_BY_MSG
    from tests.datagen import sk_bytes
    rnd = random.Random(2024)
    M, S = 5, 40
    msgs = [b"bucket-msg-%d" % m for m in range(M)]
    sks = [sk_bytes(800 + s) for s in range(S)]
    pk_pool, st = eng.batch_g2_mul(None, b"".join(sks), S, reduce_scalar=True)
    sig_pool, st2 = eng.batch_sign([msgs[m] for m in range(M) for _ in range(S)], b"".join(sks * M))
    assert st == bytes(S) and st2 == bytes(M * S)
    sig_pool = bytearray(sig_pool)
    sig_pool[64 * (2 * S + 7):64 * (2 * S + 8)] = sig_pool[64 * (2 * S + 8):64 * (2 * S + 9)]      # signer 7's signature on message 2 is wrong -> 9
    sig_pool = bytes(sig_pool)
    n = 64 * M + 128 * 3 + 37                                                # >= 64 tuples per message (tables), not a multiple of 128
    tuples = []
    for i in range(n):
        k = rnd.choice([0, 1, 3, 9, 20, 33, S])
        lst = rnd.sample(range(S), k)
        if i % 17 == 4 and lst:
            lst = lst + [lst[0]]                                             # a signer twice: direct route for that tuple
        if i % 41 == 6:
            lst = lst + [S + 1]                                              # signer out of range
        m = rnd.randrange(M)
        if i % 53 == 11:
            m = M + rnd.randrange(3)                                         # message out of range -> 2, bucket M
        tuples.append((m, lst))
    orders = {"random": tuples, "sorted": sorted(tuples, key=lambda t: t[0]), "one message": [(3, t[1]) for t in tuples]}
    eng.set_option(OPT_AGG_SUBSET_MIN_TUPLES, 1)
    try:
        for name, tl in orders.items():
            off, flat = [0], []
            for _, lst in tl:
                flat += lst
                off.append(len(flat))
            want = c.batch_aggregate_verify(msgs, pk_pool, sig_pool, [t[0] for t in tl], off, flat)
            assert {0, 2, 9} <= set(want) or name == "one message", (name, set(want))
            for knob in (1, 0):
                eng.set_option(OPT_AGG_SORT_BY_MSG, knob)
                got = eng.batch_aggregate_verify(msgs, pk_pool, sig_pool, [t[0] for t in tl], [t[1] for t in tl])
                diff = [(i, got[i], want[i], tl[i]) for i in range(n) if got[i] != want[i]]
                assert not diff, (name, knob, diff[:5])
    finally:
        eng.set_option(OPT_AGG_SORT_BY_MSG, 1)
        eng.set_option(OPT_AGG_SUBSET_MIN_TUPLES, ws_default("AGG_SUBSET_MIN_TUPLES_DEFAULT"))


def test_full_size_batch_properties(eng):
    """config-2 size (65 536): expected-status pattern (valid except every 64th), and
    permutation-equivariance of the result — size-independent properties, no oracle needed."""
    from tests.datagen import make_verify_batch
    n = 65536
    msgs, sigs, pks, expected = make_verify_batch(eng, n)
    got = eng.batch_verify(msgs, sigs, pks, flags=0)
    assert got == expected
    assert hashlib.sha256(got).hexdigest() == hashlib.sha256(expected).hexdigest()
    # reversing the batch reverses the statuses
    rm = msgs[::-1]
    rs = b"".join(sigs[64 * i:64 * i + 64] for i in range(n - 1, -1, -1))
    rp = b"".join(pks[128 * i:128 * i + 128] for i in range(n - 1, -1, -1))
    assert eng.batch_verify(rm, rs, rp, flags=0) == expected[::-1]


# ---- randomised batch verification -------------------------------------
RAND_SEED = hashlib.sha256(b"bn254/rand-seed").digest()


def test_randomized_verify_vs_oracle(eng, c):
    """statuses AND per-group verdicts equal the oracle's restatement (same seed-derived scalars): valid,
    corrupted, undecodable, identity and wrong-key items, ragged last group, both scalar widths"""
    from tests.datagen import make_verify_batch
    n = 64 * 5 + 21
    msgs, sigs, pks, expected = make_verify_batch(eng, n, corrupt_every=0, pool=7)
    sigs, pks = bytearray(sigs), bytearray(pks)
    assert eng.batch_verify_randomized(msgs, bytes(sigs), bytes(pks), RAND_SEED) == (bytes(n), b"\x01" * 6)
    g = lambda i: bytes(sigs[64 * i:64 * i + 64])
    sigs[64 * 70:64 * 71] = g(71)                                   # group 1: wrong signature
    sigs[64 * 130:64 * 131] = b"\xff" * 64                          # group 2: undecodable (status 6), rest valid -> passes
    sigs[64 * 200:64 * 201] = bytes(64)                             # group 3: identity signature
    pks[128 * 260:128 * 261] = pks[128 * 261:128 * 262]             # group 4: wrong key
    d = c.g1_mul(c.g1_generator(), (777).to_bytes(32, "big"))       # group 5: cancelling pair
    dn = c.g1_mul(c.g1_generator(), (R - 777).to_bytes(32, "big"))
    sigs[64 * 325:64 * 326] = c.g1_add(g(325), d)
    sigs[64 * 330:64 * 331] = c.g1_add(g(330), dn)
    from bn254_amd.engine import OPT_RAND_ITEMS_PER_LANE
    for flags, per_lane in ((0, 1), (0x100, 1), (1, 1), (0, 2), (0x100, 2), (0x200, 1), (0x200, 2)):
        eng.set_option(OPT_RAND_ITEMS_PER_LANE, per_lane)
        try:
            got = eng.batch_verify_randomized(msgs, bytes(sigs), bytes(pks), RAND_SEED, flags=flags)
        finally:
            eng.set_option(OPT_RAND_ITEMS_PER_LANE, 0)
        want = c.batch_verify_randomized(msgs, bytes(sigs), bytes(pks), RAND_SEED, flags=flags)
        assert got == want, (flags, per_lane)
        assert got[1] == bytes([1, 0, 1, 0, 0, 0])
        assert got[0] == eng.batch_verify(msgs, bytes(sigs), bytes(pks), flags=flags & 3)
        assert [i for i in range(n) if got[0][i]] == [70, 130, 200, 260, 325, 330]


def test_randomized_verify_golden_cases(eng, c, derived):
    cs = derived["verify_cases"]
    args = ([H(v["message_hex"]) for v in cs], b"".join(H(v["sig"]) for v in cs), b"".join(H(v["pk"]) for v in cs))
    st, gr = eng.batch_verify_randomized(*args, RAND_SEED, flags=1)
    assert list(st) == [v["status"] for v in cs] and gr == b"\x00"
    assert (st, gr) == c.batch_verify_randomized(*args, RAND_SEED, flags=1)


def test_randomized_verify_large_batch(eng):
    """16 Ki + ragged tail: all-valid batch passes without touching the exact kernels; the config-2 pattern
    (every 64th corrupted) fails every group and falls back to exact statuses; independent of the seed"""
    from tests.datagen import make_verify_batch
    n = 16384 + 45
    msgs, sigs, pks, _ = make_verify_batch(eng, n, corrupt_every=0)
    st, gr = eng.batch_verify_randomized(msgs, sigs, pks, RAND_SEED)
    assert st == bytes(n) and gr == b"\x01" * ((n + 63) // 64)
    msgs, sigs, pks, expected = make_verify_batch(eng, n)
    for seed in (RAND_SEED, bytes(32)):
        st, gr = eng.batch_verify_randomized(msgs, sigs, pks, seed)
        assert st == expected and gr == b"\x00" * (n // 64) + b"\x01"
    # one bad item in a large valid batch: only its group is re-verified (both Miller kernels)
    from bn254_amd.engine import OPT_RAND_ITEMS_PER_LANE
    bad = bytearray(make_verify_batch(eng, n, corrupt_every=0)[1])
    bad[64 * 9000:64 * 9001] = bad[64 * 9001:64 * 9002]
    for per_lane in (1, 2):
        eng.set_option(OPT_RAND_ITEMS_PER_LANE, per_lane)
        try:
            st, gr = eng.batch_verify_randomized(msgs, bytes(bad), pks, RAND_SEED)
        finally:
            eng.set_option(OPT_RAND_ITEMS_PER_LANE, 0)
        assert [i for i in range(n) if st[i]] == [9000] and st[9000] == 9 and gr.count(b"\x00") == 1 and gr[9000 // 64] == 0


def test_randomized_verify_edge_sizes(eng, c):
    """n = 0, 1, 63, 64, 65, 129 and messages of 0..130 bytes (SHA-256 padding boundaries) through the randomised path"""
    from tests.datagen import sk_bytes
    assert eng.batch_verify_randomized([], b"", b"", RAND_SEED) == (b"", b"")
    sks = [sk_bytes(j) for j in range(5)]
    pk_pool, _ = eng.batch_g2_mul(None, b"".join(sks), 5, reduce_scalar=True)
    lens = [0, 1, 31, 32, 54, 55, 56, 63, 64, 65, 118, 119, 120, 127, 128, 130]
    for n in (1, 63, 64, 65, 129):
        msgs = [hashlib.sha256(b"edge%d" % i).digest() * 5 for i in range(n)]
        msgs = [m[:lens[i % len(lens)]] for i, m in enumerate(msgs)]
        sigs, st = eng.batch_sign(msgs, b"".join(sks[i % 5] for i in range(n)))
        assert st == bytes(n)
        pks = b"".join(pk_pool[128 * (i % 5):128 * (i % 5) + 128] for i in range(n))
        assert eng.batch_verify_randomized(msgs, sigs, pks, RAND_SEED) == (bytes(n), b"\x01" * ((n + 63) // 64))
        assert eng.batch_verify(msgs, sigs, pks) == bytes(n)
        bad = bytearray(sigs)
        bad[64 * (n - 1):64 * n] = c.g1_generator()
        got = eng.batch_verify_randomized(msgs, bytes(bad), pks, RAND_SEED)
        assert got == c.batch_verify_randomized(msgs, bytes(bad), pks, RAND_SEED, flags=0)
        assert got[0] == bytes(n - 1) + b"\x09" and got[1][-1] == 0


def test_argument_validation_and_empty_batches(eng):
    """error behaviour at the C ABI: empty batches succeed, NULL / misaligned device pointers are refused with
    BN254_E_BAD_ARGUMENT / BN254_E_MISALIGNED (never a fault), bad option values are refused"""
    import ctypes
    L, h = eng._lib, eng._h
    assert eng.batch_verify([], b"", b"") == b""
    assert eng.batch_hash_to_g1([]) == (b"", b"", b"")
    assert eng.batch_pairing_check(b"", b"", 0, 2) == b""
    assert eng.batch_g1_add(b"", b"", 0)[0] == b""
    st = ctypes.create_string_buffer(8)
    assert L.bn254_batch_verify(h, None, None, None, None, 4, 0, st) == -10001
    assert L.bn254_batch_verify(None, None, None, None, None, 0, 0, st) == -10001
    assert L.bn254_batch_verify_randomized(h, b"", (ctypes.c_uint64 * 1)(0), b"", b"", 0, 0, None, st, None) == -10001   # no seed
    p, off = 0x7F0000001000, 0x7F0000100000        # pointer VALUES only: every call below is refused before any access
    assert L.bn254_batch_verify_device(h, p, off, p + 1, p + 1024, 2, 0, p + 2048, None) == -10002        # sigs not 4-byte aligned
    assert L.bn254_batch_verify_device(h, p, off + 4, p, p + 1024, 2, 0, p + 2048, None) == -10002        # offsets not 8-byte aligned
    assert L.bn254_batch_verify_device(h, p, off, p, None, 2, 0, p + 2048, None) == -10001
    assert L.bn254_ctx_set_option(h, 3, 7) == -10001 and L.bn254_ctx_set_option(h, 2, 300) == -10001
    assert L.bn254_ctx_set_option(h, 99, 0) == -10001
    # host entry points refuse a decreasing offsets array (a wrapped length would walk outside the staged buffer)
    bad_off = (ctypes.c_uint64 * 4)(0, 40, 8, 48)
    assert L.bn254_batch_verify(h, bytes(64), bad_off, bytes(192), bytes(384), 3, 0, st) == -10001
    pts = ctypes.create_string_buffer(192)
    assert L.bn254_batch_hash_to_g1(h, bytes(64), bad_off, 3, pts, st, None) == -10001
    assert L.bn254_batch_g1_sum(h, bytes(64 * 48), bad_off, 3, pts, st) == -10001
    # the context is still usable afterwards
    from tests.datagen import make_verify_batch
    msgs, sigs, pks, expected = make_verify_batch(eng, 130)
    assert eng.batch_verify(msgs, sigs, pks) == expected


def test_aggregate_verify_index_validation_device_and_host(eng, c):
    """indices from caller memory: a message index >= n_msgs or a signer index >= n_signers is IndexOutOfBounds (status 2)
    and a decreasing tuple_off pair in the _device variant is status 2 as well — never an out-of-range pool read; both
    kernels (lane pairs / one lane per tuple) and the oracle's restatement agree"""
    import torch
    from bn254_amd.engine import OPT_PAIR_LANES
    from tests.datagen import sk_bytes
    dev = torch.device("cuda", 0)
    M, S = 2, 5
    msgs = [b"idx-msg-%d" % m for m in range(M)]
    sks = [sk_bytes(300 + s) for s in range(S)]
    pk_pool, _ = eng.batch_g2_mul(None, b"".join(sks), S, reduce_scalar=True)
    sig_pool, _ = eng.batch_sign([msgs[m] for m in range(M) for _ in range(S)], b"".join(sks * M))
    tuple_msg = [0, 1, 2, 0xFFFFFFFF, 1, 0, 1]
    lists = [[0, 1], [2, 3, 4], [0], [1], [S], [4, 0, 3], []]
    want = c.batch_aggregate_verify(msgs, pk_pool, sig_pool, tuple_msg, [0, 2, 5, 6, 7, 8, 11, 11], sum(lists, []))
    assert list(want) == [0, 0, 2, 2, 2, 0, 0]
    for pair in (1, 0):
        eng.set_option(OPT_PAIR_LANES, pair)
        assert eng.batch_aggregate_verify(msgs, pk_pool, sig_pool, tuple_msg, lists) == want
        # _device entry with a decreasing offset pair (tuple 1: [5, 2)) -> status 2 for that tuple, others unaffected
        t = lambda data, dt: torch.tensor(data, dtype=dt, device=dev)   # noqa: E731
        d_msgs = torch.frombuffer(bytearray(b"".join(msgs)), dtype=torch.uint8).to(dev)
        d_moff = t([0, len(msgs[0]), len(msgs[0]) + len(msgs[1])], torch.int64)
        d_pk = torch.frombuffer(bytearray(pk_pool), dtype=torch.uint8).to(dev)
        d_sig = torch.frombuffer(bytearray(sig_pool), dtype=torch.uint8).to(dev)
        # ... and a message index >= n_msgs (tuple 3) through the same entry point
        d_tm, d_to, d_si = t([0, 1, 1, 9], torch.int32), t([0, 5, 2, 4, 6], torch.int64), t([0, 1, 2, 3, 4, 0], torch.int32)
        d_st = torch.full((4,), 255, dtype=torch.uint8, device=dev)
        stream = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(stream):
            eng.batch_aggregate_verify_device(d_msgs.data_ptr(), d_moff.data_ptr(), M, d_pk.data_ptr(), S, d_sig.data_ptr(), d_tm.data_ptr(),
                                              d_to.data_ptr(), d_si.data_ptr(), 4, d_st.data_ptr(), stream=stream.cuda_stream)
        stream.synchronize()
        assert d_st.cpu().tolist() == [0, 2, 0, 2]
    eng.set_option(OPT_PAIR_LANES, 1)


def test_device_entry_points_bound_check_message_offsets(eng, c):
    """*_device entry points read their offsets array on the device: a reversed pair, or — once the caller has declared the
    buffer size (bn254_ctx_expect_msgs_len) — a span past the buffer, is never dereferenced; its item reports 5
    (InvalidLength), every other item what the oracle says.  Small batch (direct hash + wave roles) and lane-pair sizes;
    verify, hash_to_g1 and sign."""
    import torch
    from tests.datagen import make_verify_batch
    dev = torch.device("cuda", 0)
    for n in (70, 20000):
        msgs, sigs, pks, expected = make_verify_batch(eng, n, corrupt_every=9)          # 32-byte messages, status 9 at 8, 17, ...
        blob = b"".join(msgs)
        off = [32 * i for i in range(n + 1)]
        want = c.batch_verify(msgs, sigs, pks, flags=0, nthreads=8)[0]
        assert want == expected
        d_msgs = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(dev)
        d_sigs = torch.frombuffer(bytearray(sigs), dtype=torch.uint8).to(dev)
        d_pks = torch.frombuffer(bytearray(pks), dtype=torch.uint8).to(dev)
        stream = torch.cuda.Stream(device=dev)
        for declare in (True, False):
            bad = list(off)
            bad[4] = off[3] - 1            # item 3 reversed -> 5; item 4 = [off[3] - 1, off[5]) is another message: not compared
            exp = bytearray(want)
            exp[3] = 5
            if declare:
                bad[n - 1] = len(blob) + (1 << 40)        # item n-2 runs past the buffer, item n-1 = [huge, off[n]) is reversed
                exp[n - 2] = exp[n - 1] = 5
                eng.expect_msgs_len(len(blob))
            else:
                bad[n] = off[n - 1] - 1                   # undeclared size: only reversed pairs can be seen — item n-1
                exp[n - 1] = 5
            d_off = torch.tensor(bad, dtype=torch.int64, device=dev)
            d_st = torch.full((n,), 255, dtype=torch.uint8, device=dev)
            eng.batch_verify_device(d_msgs.data_ptr(), d_off.data_ptr(), d_sigs.data_ptr(), d_pks.data_ptr(), n, d_st.data_ptr(), flags=0,
                                    stream=stream.cuda_stream)
            stream.synchronize()
            got = d_st.cpu().numpy().tobytes()
            diff = [(i, got[i], exp[i]) for i in range(n) if i != 4 and got[i] != exp[i]]
            assert not diff, (n, declare, diff[:10])
            assert got[4] in (5, 9)
    # hash_to_g1 and sign see the same check (status 5, tries 0)
    n = 40
    msgs = [b"span-%d" % i for i in range(n)]
    blob = b"".join(msgs)
    off = [0]
    for m in msgs:
        off.append(off[-1] + len(m))
    off[n] = len(blob) + 1000
    d_msgs = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(dev)
    d_off = torch.tensor(off, dtype=torch.int64, device=dev)
    d_pts = torch.zeros(64 * n, dtype=torch.uint8, device=dev)
    d_st = torch.full((n,), 255, dtype=torch.uint8, device=dev)
    d_tr = torch.full((n,), 255, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    eng.expect_msgs_len(len(blob))
    eng.batch_hash_to_g1_device(d_msgs.data_ptr(), d_off.data_ptr(), n, d_pts.data_ptr(), d_st.data_ptr(), d_tr.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    st, tr, pts = d_st.cpu().tolist(), d_tr.cpu().tolist(), d_pts.cpu().numpy().tobytes()
    assert st == [0] * (n - 1) + [5] and tr[n - 1] == 0
    for i in range(n - 1):
        assert (0, pts[64 * i:64 * i + 64], tr[i]) == c.hash_to_g1(msgs[i])
    # the declaration is consumed by ONE call: the same arrays again without it -> the last span is taken at face value
    # only if it lies inside the allocation; do not run that (it would read past the buffer) — instead check that a
    # declaration larger than the span accepts it
    off[n] = len(blob)
    d_off = torch.tensor(off, dtype=torch.int64, device=dev)
    eng.expect_msgs_len(len(blob))
    eng.batch_hash_to_g1_device(d_msgs.data_ptr(), d_off.data_ptr(), n, d_pts.data_ptr(), d_st.data_ptr(), d_tr.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    assert d_st.cpu().tolist() == [0] * n


def test_keyed_verify_vs_oracle(eng, c, derived):
    """bn254_ctx_register_keys + bn254_batch_verify_keyed[_device]: per-key registration statuses equal what the oracle's decoder
    (subgroup check on) reports, and the keyed statuses equal bn254o_batch_verify on the EXPANDED keys — valid and failing
    tuples, a key outside G2, a key with a coordinate >= q, an off-curve key, the identity key (pair contributes 1), an index
    >= n_keys (IndexOutOfBounds, 2), malformed signatures (their status comes first), at a size inside one wave and one
    past the small-batch threshold."""
    import random
    import torch
    from tests.datagen import D, sk_bytes
    rnd = random.Random(31)
    K = 37
    sks = [sk_bytes(500 + j) for j in range(K)]
    pk_pool, st = eng.batch_g2_mul(None, b"".join(sks), K, reduce_scalar=True)
    assert st == bytes(K)
    keys = [bytearray(pk_pool[128 * j:128 * j + 128]) for j in range(K)]
    keys[5] = bytearray(H(derived["g2_not_in_subgroup"]))                   # on the twist, outside the order-r subgroup -> 4
    keys[6][0:32] = Q.to_bytes(32, "big")                                    # x.re = q -> 6
    keys[7][127] ^= 1                                                        # off the curve -> 4
    keys[8] = bytearray(128)                                                 # identity
    key_bytes = b"".join(bytes(k) for k in keys)
    kst = eng.register_keys(key_bytes)
    want_kst = bytes(c.batch_verify([b""], bytes(64), bytes(k), flags=1)[0][0] for k in keys)   # sig = identity: status = the key's decode status, or 0 / 9
    for j in range(K):
        assert kst[j] == (want_kst[j] if want_kst[j] in (3, 4, 6) else 0), (j, kst[j], want_kst[j])
    assert kst[5] == 4 and kst[6] == 6 and kst[7] == 4 and kst[8] == 0
    for n in (50, 4099, 20011):
        msgs = [D("keyed", i) for i in range(n)]
        kidx = [rnd.randrange(K) for _ in range(n)]
        sigs, st = eng.batch_sign(msgs, b"".join(sks[k] for k in kidx))
        assert st == bytes(n)
        sigs = bytearray(sigs)
        for i in range(3, n, 11):
            sigs[64 * i:64 * i + 64] = sigs[64 * (i - 1):64 * i]             # wrong signature -> 9
        for i in range(7, n, 97):
            sigs[64 * i + 63] ^= 1                                            # off the curve -> 4 (comes before the key's status)
        for i in range(9, n, 131):
            sigs[64 * i:64 * i + 64] = bytes(64)                              # identity signature
        oob = set(range(13, n, 173))
        idx_call = [K + 5 if i in oob else kidx[i] for i in range(n)]
        idx_call[1 % n] = 0xFFFFFFFF
        oob.add(1 % n)
        want = bytearray(c.batch_verify(msgs, bytes(sigs), b"".join(bytes(keys[k]) for k in kidx), flags=1, nthreads=8)[0])
        sig_only = c.batch_verify(msgs, bytes(sigs), bytes(128) * n, flags=1, nthreads=8)[0]    # identity key: status = the signature's decode status or 0 / 9
        for i in oob:
            want[i] = sig_only[i] if sig_only[i] in (3, 4, 6) else 2
        got = eng.batch_verify_keyed(msgs, bytes(sigs), idx_call)
        diff = [(i, got[i], want[i], kidx[i]) for i in range(n) if got[i] != want[i]]
        assert not diff, diff[:10]
        assert n < 1000 or {0, 2, 4, 6, 9} <= set(got)
        # the smallest batches (<= 1 536) take the KEYED lane machine (k_miller_verify_lmk: the key's line table, no twist-point wave); off: the
        # keys are expanded and the generic small-batch kernels run; forced on at the larger size too (several passes of 256 workgroups)
        from bn254_amd.engine import OPT_LM_MAX_BATCH
        for lim in (0, 1 << 20):
            eng.set_option(OPT_LM_MAX_BATCH, lim)
            try:
                if n <= TRIO_DEFAULT or lim == 0:
                    got_l = eng.batch_verify_keyed(msgs, bytes(sigs), idx_call)
                    diff = [(i, got_l[i], want[i], kidx[i]) for i in range(n) if got_l[i] != want[i]]
                    assert not diff, (lim, diff[:10])
            finally:
                eng.set_option(OPT_LM_MAX_BATCH, LM_DEFAULT)
        # the device entry point on the caller's stream
        dev = torch.device("cuda", 0)
        t8 = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)   # noqa: E731
        d_msgs, d_sigs = t8(b"".join(msgs)), t8(bytes(sigs))
        d_off = torch.arange(0, 32 * (n + 1), 32, dtype=torch.int64, device=dev)
        d_idx = torch.tensor([x if x < 2**31 else x - 2**32 for x in idx_call], dtype=torch.int32, device=dev)
        d_st = torch.full((n,), 255, dtype=torch.uint8, device=dev)
        stream = torch.cuda.Stream(device=dev)
        eng.batch_verify_keyed_device(d_msgs.data_ptr(), d_off.data_ptr(), d_sigs.data_ptr(), d_idx.data_ptr(), n, d_st.data_ptr(), stream=stream.cuda_stream)
        stream.synchronize()
        assert d_st.cpu().numpy().tobytes() == bytes(want)
        # the keyed randomised mode (forced on at this size): the same statuses — failing groups fall back to the exact kernels
        from bn254_amd.engine import OPT_RAND_MIN_BATCH
        eng.set_option(OPT_RAND_MIN_BATCH, 0)
        for fl in (0, 0x100, 0x200):
            got_r = eng.batch_verify_keyed_randomized(msgs, bytes(sigs), idx_call, RAND_SEED, flags=fl)
            diff = [(i, got_r[i], want[i], kidx[i]) for i in range(n) if got_r[i] != want[i]]
            assert not diff, (fl, diff[:10])
        d_st.fill_(255)
        eng.batch_verify_keyed_randomized_device(d_msgs.data_ptr(), d_off.data_ptr(), d_sigs.data_ptr(), d_idx.data_ptr(), n, RAND_SEED, d_st.data_ptr(),
                                                 stream=stream.cuda_stream)
        stream.synchronize()
        assert d_st.cpu().numpy().tobytes() == bytes(want)
        eng.set_option(OPT_RAND_MIN_BATCH, 131072)
    # an all-valid batch: every group passes, nothing is re-checked
    n = 5000
    msgs = [D("keyed-ok", i) for i in range(n)]
    kk = [i % 4 for i in range(n)]                                              # keys 0..3 are valid
    sg, st = eng.batch_sign(msgs, b"".join(sks[k] for k in kk))
    eng.set_option(OPT_RAND_MIN_BATCH, 0)
    assert eng.batch_verify_keyed_randomized(msgs, sg, kk, RAND_SEED) == bytes(n)
    eng.set_option(OPT_RAND_MIN_BATCH, 131072)
    # more keys than a wave has lanes (the grouping kernel scans the key counts 64 at a time), some of them unused
    K2 = 150
    sks2 = [sk_bytes(900 + j) for j in range(K2)]
    pool2, st = eng.batch_g2_mul(None, b"".join(sks2), K2, reduce_scalar=True)
    assert eng.register_keys(pool2) == bytes(K2)
    n = 3000
    msgs2 = [D("keyed-many", i) for i in range(n)]
    kk2 = [(i * 7) % 140 for i in range(n)]                                    # keys 140 .. 149 never named
    sg2, st = eng.batch_sign(msgs2, b"".join(sks2[k] for k in kk2))
    sg2 = bytearray(sg2)
    for i in range(5, n, 37):
        sg2[64 * i:64 * i + 64] = sg2[64 * (i - 1):64 * i]
    want2 = c.batch_verify(msgs2, bytes(sg2), b"".join(pool2[128 * k:128 * k + 128] for k in kk2), flags=1, nthreads=8)[0]
    eng.set_option(OPT_RAND_MIN_BATCH, 0)
    assert eng.batch_verify_keyed_randomized(msgs2, bytes(sg2), kk2, RAND_SEED) == want2 and want2.count(9) == len(range(5, n, 37))
    eng.set_option(OPT_RAND_MIN_BATCH, 131072)
    assert eng.batch_verify_keyed(msgs2, bytes(sg2), kk2) == want2
    eng.register_keys(key_bytes)
    # the Python mirror of the reference API
    import bn254_amd as bn
    good = [bn.PublicKey(bytes(keys[j])) for j in (0, 1, 2)]
    got_reg = bn.ECDSA.register_keys(good + [bn.PublicKey(bytes(keys[j])) for j in (5, 6, 7, 8)])
    assert got_reg == [None, None, None, bn.Error(4), bn.Error(6), bn.Error(4), None], got_reg      # the mirror's error mapping, key by key
    assert got_reg[3].kind == bn.ErrorKind.InvalidGroupPoint and got_reg[4].kind == bn.ErrorKind.NotMemberError
    res = bn.ECDSA.batch_verify_keyed([b"api-keyed"] * 3, [bn.ECDSA.sign(b"api-keyed", bn.PrivateKey.try_from(sks[0].hex()))] * 3, [0, 1, 7])
    assert res[0] is None and res[1].kind == bn.ErrorKind.VerificationFailed and res[2].kind == bn.ErrorKind.IndexOutOfBounds
    # an empty key set: every index is out of range
    assert eng.register_keys(b"") == b""
    assert eng.batch_verify_keyed(msgs[:3], bytes(sigs[:192]), [0, 1, 2]) == bytes([2, 2, 2])
    # ... also on a context that never registered a key (no table in HBM at all); a malformed signature keeps its own status
    fresh = bn.Engine(0)
    bad_sig = bytearray(sigs[:64]); bad_sig[63] ^= 1
    assert fresh.batch_verify_keyed(msgs[:2], bytes(bad_sig) + bytes(sigs[64:128]), [0, 5]) == bytes([4, 2])


def test_cpp_host_mirror_example(eng):
    """the C++ mirror of the reference API (bn254_amd/host/bn254.hpp) runs the reference's example scenario
    (/root/reference/examples/bn254.rs:3-34) end to end on the GPU"""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    host = os.path.join(root, "bn254_amd", "host")
    exe = os.path.join(host, "example")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", os.path.join(host, "example.cpp"), "-L" + os.path.join(root, "bn254_amd"),
                           "-lbn254hip", "-Wl,-rpath," + os.path.join(root, "bn254_amd"), "-o", exe])
    p = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "Successful aggregate signature verification" in p.stdout, (p.stdout, p.stderr)


def test_api_batch_verify_randomized(eng):
    """the Python mirror's ECDSA.batch_verify / batch_verify_randomized give the same per-item results"""
    import bn254_amd
    from bn254_amd import ECDSA, PrivateKey, PublicKey, Signature, Error, ErrorKind
    sks = [PrivateKey(1000 + j) for j in range(5)]
    pks = [PublicKey.from_private_key(k) for k in sks]
    msgs = [b"api-rand-%d" % i for i in range(70)]
    sigs = [ECDSA.sign(m, sks[i % 5]) for i, m in enumerate(msgs)]
    keys = [pks[i % 5] for i in range(70)]
    assert ECDSA.batch_verify_randomized(msgs, sigs, keys, seed=RAND_SEED) == [None] * 70
    sigs[7] = sigs[8]
    keys[66] = keys[67]
    want = ECDSA.batch_verify(msgs, sigs, keys)
    assert [i for i, r in enumerate(want) if r is not None] == [7, 66] and want[7] == Error(ErrorKind.VerificationFailed)
    assert ECDSA.batch_verify_randomized(msgs, sigs, keys, seed=RAND_SEED) == want
    assert ECDSA.batch_verify_randomized(msgs, sigs, keys) == want          # os.urandom seed
    assert ECDSA.batch_verify_randomized(msgs, sigs, keys, rand64=True) == want


def test_batch_verify_from_compressed_encodings(eng, kats):
    """bn254_batch_verify_compressed == decompress (from_compressed semantics) + verify, item by item"""
    from bn254_amd import PublicKey, Signature
    from tests.datagen import make_verify_batch
    n = 300
    msgs, sigs, pks, expected = make_verify_batch(eng, n, corrupt_every=17, pool=9)
    sc = bytearray(b"".join(Signature(sigs[64 * i:64 * i + 64]).to_compressed() for i in range(n)))
    pc = bytearray(b"".join(PublicKey(pks[128 * i:128 * i + 128]).to_compressed() for i in range(n)))
    assert eng.batch_verify_compressed(msgs, bytes(sc), bytes(pc)) == expected
    # malformed encodings: bad sign bytes, x >= q, x without a root, public key outside the subgroup / undecodable
    sc[33 * 3] = 0x04
    pc[65 * 5] = 0x0C
    sc[33 * 7 + 1:33 * 8] = (Q + 5).to_bytes(32, "big")
    sc[33 * 9 + 1:33 * 10] = (4).to_bytes(32, "big")          # 4^3 + 3 = 67 is not a square mod q
    pc[65 * 11 + 40] ^= 0x55
    got = eng.batch_verify_compressed(msgs, bytes(sc), bytes(pc))
    us, s1 = eng.batch_g1_decompress(bytes(sc), n)
    up, s2 = eng.batch_g2_decompress(bytes(pc), n)
    base = eng.batch_verify(msgs, us, up)
    want = bytes(s1[i] or s2[i] or base[i] for i in range(n))
    assert got == want and got[3] == 3 and got[5] == 3 and got[7] == 6 and got[9] == 6 and got[11] in (6, 9)
    # the reference's own compressed KAT decodes and verifies
    v = kats["sign"][0]
    pk = PublicKey.from_private_key(__import__("bn254_amd").PrivateKey.try_from(v["private_key"]))
    assert eng.batch_verify_compressed([H(v["message_hex"])], H(v["signature_compressed"]), pk.to_compressed()) == b"\x00"


def test_randomized_small_batches_route_to_exact_kernels(c):
    """default policy: below BN254_OPT_RAND_MIN_BATCH the randomised entry point runs the exact kernels — same statuses,
    group_ok = no item of the group failed the pairing check (what the combined check reports as well)"""
    import bn254_amd
    from tests.datagen import make_verify_batch
    e = bn254_amd.Engine(0)                       # default options
    n = 200
    msgs, sigs, pks, expected = make_verify_batch(e, n, corrupt_every=70, pool=5)
    sigs = bytearray(sigs)
    sigs[64 * 130:64 * 131] = b"\xff" * 64       # undecodable: status 6, does not fail its group
    st, gr = e.batch_verify_randomized(msgs, bytes(sigs), pks, RAND_SEED)
    want = c.batch_verify_randomized(msgs, bytes(sigs), pks, RAND_SEED, flags=0)
    assert (st, gr) == want and gr == bytes([1, 0, 0, 1]) and st[130] == 6


def test_measurement_entry_points(eng):
    """the round-4 measurement ABI: argument validation (bad programs, sizes, modes are BN254_E_BAD_ARGUMENT, never a launch), the clock
    probe reports nothing until it is switched on and plausible clocks afterwards, and none of it disturbs verify results"""
    import ctypes
    from bn254_amd.engine import OPT_CLOCK_PROBE
    from tests.datagen import make_verify_batch
    BAD = -10001
    lib, h = eng._lib, eng._h
    n = 20000                                                           # above the small-batch threshold: the lane-pair kernels
    msgs, sigs, pks, expected = make_verify_batch(eng, n, corrupt_every=9)
    assert eng.batch_verify(msgs, sigs, pks) == expected
    ms = ctypes.c_float()
    mhz = (ctypes.c_double * 3)()
    assert lib.bn254_ctx_last_clocks(h, mhz) == BAD                     # probe off: no buffer
    assert lib.bn254_probe_leaf_floor(h, 0, 0, ctypes.byref(ms)) == BAD
    assert lib.bn254_probe_leaf_floor(h, n, 8, ctypes.byref(ms)) == BAD
    assert lib.bn254_probe_leaf_floor(h, 1 << 40, 0, ctypes.byref(ms)) == BAD          # beyond the workspace
    for prog in (bytes([8, 0]), bytes([0, 0]), bytes([4, 10]), bytes([1, 200]), bytes([6, 0]), bytes([6, 4])):   # unknown opcode, END inside, slot out of range, bad Frobenius power
        assert lib.bn254_probe_fe_program(h, n, prog, 1, ctypes.byref(ms)) == BAD, prog
    assert lib.bn254_probe_fe_program(h, n, bytes([3, 0]), 0, ctypes.byref(ms)) == BAD
    t_sq = eng.probe_fe_program(n, [(2, 0)] + [(3, 0)] * 20)
    t_mul = eng.probe_fe_program(n, [(2, 0)] + [(4, 0)] * 20)
    assert 0.0 < t_sq < 200.0 and 0.0 < t_mul < 200.0                   # plausible durations only: orderings and tight windows flake on a shared box
    for mode in range(8):                                               # dependent chains, inlined / called leaf, four / one / two independent chains
        assert 0.0 < eng.probe_leaf_floor(n, mode) < 200.0, mode
    eng.set_option(OPT_CLOCK_PROBE, 1)
    try:
        assert eng.batch_verify(msgs, sigs, pks) == expected
        clocks = eng.last_clocks()
        assert 500.0 < clocks["miller_loop"] < 4000.0 and 500.0 < clocks["final_exp"] < 4000.0, clocks
        again = eng.last_clocks()                                       # read AND cleared: nothing ran in between
        assert again["miller_loop"] == 0.0 and again["final_exp"] == 0.0, again
        assert eng.batch_verify(msgs, sigs, pks) == expected and eng.batch_verify(msgs, sigs, pks) == expected
        two = eng.last_clocks()                                         # accumulated over two launches: still a clock, not a sum of clocks
        assert 500.0 < two["miller_loop"] < 4000.0, two
    finally:
        eng.set_option(OPT_CLOCK_PROBE, 0)
    assert lib.bn254_ctx_last_clocks(h, mhz) == BAD
    assert eng.batch_verify(msgs, sigs, pks) == expected
