"""The PAIR layout of the Fq2 tower (bn254_amd/csrc/bn254_fp2_pair.h; kernels in bn254_pair.hip): the per-role code
compiled for the host with both lane roles run in sequence — parity against the oracle / golden vectors, and the
limb / value bound proof of this layout (tracker build aborts on a violation).  CPU only."""
import ctypes
import os
import subprocess
import sys

import pytest

from oracle import c_oracle as c

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H = bytes.fromhex


@pytest.fixture(scope="module")
def pair_lib():
    from tests import hostsim_binding
    hostsim_binding.build_all()
    return ctypes.CDLL(os.path.join(ROOT, "tests", "hostsim", "libhostsim_pair.so"))


def test_pair_layout_verify_and_gt_match_oracle(pair_lib, derived, kats):
    n = 0
    for v in derived["verify_cases"]:
        if v["status"] not in (0, 9):
            continue                                   # decode errors never reach the pairing kernels
        st, h, _ = c.hash_to_g1(H(v["message_hex"]))
        assert pair_lib.hp_verify_decoded(h, H(v["sig"]), H(v["pk"])) == v["status"], v["name"]
        n += 1
    assert n >= 10
    for v in derived["pairing_gt"]:
        out = ctypes.create_string_buffer(384)
        pair_lib.hp_pairing(H(v["g1"]), H(v["g2"]), out)
        assert out.raw.hex() == v["gt"]
    # random pairs against the oracle's canonical Gt
    import hashlib
    g1, g2 = c.g1_generator(), c.g2_generator()
    for i in range(4):
        a = hashlib.sha256(b"pair-a%d" % i).digest()
        b = hashlib.sha256(b"pair-b%d" % i).digest()
        p, q = c.g1_mul(g1, a), c.g2_mul(g2, b)
        out = ctypes.create_string_buffer(384)
        pair_lib.hp_pairing(p, q, out)
        assert out.raw == c.pairing(p, q)


def test_keyed_verify_line_tables(pair_lib, derived):
    """keyed verify on the host: the key's 87 lines in the c2 = 1 form (g2_line_table) give, through the table-driven loop
    (miller_loop_keyed), the very Gt value of the generic loop on every verify case; the table built by the one-lane source
    (k_register_keys) and by the pair-layout source are the same words; registration statuses for keys outside G2 / off the
    curve / out of range; and the products per lane of the keyed loop (2 508 dual + 348 single against 3 194 + 778)."""
    from tests import hostsim_binding as hs
    one = hs.lib()
    one.hs_register_key.argtypes = [ctypes.c_char_p, ctypes.c_uint32, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int)]
    W = 87 * 2 * 2 * 9
    n = 0
    for v in derived["verify_cases"]:
        if v["status"] not in (0, 9):
            continue
        st, h, _ = c.hash_to_g1(H(v["message_hex"]))
        tab_pair = (ctypes.c_int32 * W)()
        assert pair_lib.hp_verify_keyed_decoded(h, H(v["sig"]), H(v["pk"]), tab_pair) == v["status"], v["name"]
        tab_one, inf = (ctypes.c_int32 * W)(), ctypes.c_int(0)
        assert one.hs_register_key(H(v["pk"]), 0, tab_one, ctypes.byref(inf)) == 0
        assert inf.value == (1 if H(v["pk"]) == bytes(128) else 0)
        if not inf.value:
            assert list(tab_one) == list(tab_pair), v["name"]
        n += 1
    assert n >= 10
    tab, inf = (ctypes.c_int32 * W)(), ctypes.c_int(0)
    g2 = bytearray(c.g2_generator())
    assert one.hs_register_key(H(derived["g2_not_in_subgroup"]), 0, tab, ctypes.byref(inf)) == 4
    bad = bytearray(g2); bad[127] ^= 1
    assert one.hs_register_key(bytes(bad), 0, tab, ctypes.byref(inf)) == 4
    Qm = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
    bad = bytearray(g2); bad[32:64] = Qm.to_bytes(32, "big")
    assert one.hs_register_key(bytes(bad), 0, tab, ctypes.byref(inf)) == 6
    assert one.hs_register_key(bytes(128), 2, tab, ctypes.byref(inf)) == 4          # BN254_FLAG_REJECT_IDENTITY
    out = (ctypes.c_ulonglong * 2)()
    v = derived["verify_cases"][0]
    pair_lib.hp_lane_counts_keyed(c.hash_to_g1(H(v["message_hex"]))[1], H(v["sig"]), H(v["pk"]), out)
    assert list(out) == [64 * 12 + 87 * 20, 87 * 4]


def test_pair_layout_multi_pair_and_g2_sums(pair_lib, derived):
    import hashlib
    g1, g2 = c.g1_generator(), c.g2_generator()
    ps = [c.g1_mul(g1, hashlib.sha256(b"pp-a%d" % i).digest()) for i in range(4)]
    qs = [c.g2_mul(g2, hashlib.sha256(b"pp-b%d" % i).digest()) for i in range(4)]
    out = ctypes.create_string_buffer(384)
    pair_lib.hp_pairing_product4(b"".join(ps), b"".join(qs), out)
    assert out.raw == c.pairing(b"".join(ps), b"".join(qs), k=4)
    # running G2 sums incl. P + P, P + (-P), identity terms; subgroup membership
    Q = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
    neg = lambda q: q[:64] + b"".join(((Q - int.from_bytes(q[64 + 32 * k:96 + 32 * k], "big")) % Q).to_bytes(32, "big") for k in range(2))  # noqa: E731
    for pts in ([qs[0], qs[1], qs[2]], [qs[0], qs[0]], [qs[1], neg(qs[1])], [bytes(128), qs[3], bytes(128)], [qs[0], qs[0], qs[0], neg(qs[0])]):
        want = bytes(128)
        for q in pts:
            want = c.g2_add(want, q)
        o = ctypes.create_string_buffer(128)
        assert pair_lib.hp_g2_sum_and_subgroup(b"".join(pts), len(pts), o) == 1
        assert o.raw == want
    o = ctypes.create_string_buffer(128)
    assert pair_lib.hp_g2_sum_and_subgroup(H(derived["g2_not_in_subgroup"]), 1, o) == 0


def test_pair_layout_g2_decompress(pair_lib, kats, derived):
    """G2::from_compressed in the pair layout against the classic-layout host build (itself pinned on the reference's
    compressed-key vector) on random keys of both signs, and on malformed encodings"""
    import hashlib
    from tests import hostsim_binding as hs
    Q = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47

    def compress(p):                                   # utils.rs:130-158
        w = [int.from_bytes(p[i:i + 32], "big") for i in range(0, 128, 32)]
        y = w[3] * Q + w[2]
        yn = ((-w[3]) % Q) * Q + ((-w[2]) % Q)
        return bytes([0x0B if y > yn else 0x0A]) + (w[1] * Q + w[0]).to_bytes(64, "big")
    cases = [H(kats["g2_compressed_roundtrip"]["hex"])]
    g2 = c.g2_generator()
    signs = set()
    for i in range(8):
        p = c.g2_mul(g2, hashlib.sha256(b"dec%d" % i).digest())
        cases.append(compress(p))
        signs.add(cases[-1][0])
        out = ctypes.create_string_buffer(128)
        assert pair_lib.hp_g2_decompress(cases[-1], out) == 0 and out.raw == p
    assert signs == {0x0A, 0x0B}
    bad = [b"\x0c" + cases[1][1:], cases[1][:1] + b"\xff" * 64, cases[2][:40] + bytes([cases[2][40] ^ 0x10]) + cases[2][41:], compress(H(derived["g2_not_in_subgroup"]))]
    for enc in cases + bad:
        out = ctypes.create_string_buffer(128)
        st = pair_lib.hp_g2_decompress(enc, out)
        wst, want = hs.g2_decompress(enc)
        assert (st, out.raw if st == 0 else None) == (wst, want if wst == 0 else None), enc.hex()[:20]


def test_nonet_schedule_matches_fe_machine(pair_lib, derived):
    """the final exponentiation on NINE lane pairs per verify (bn254_nonet.hip, default for every batch <= 3 072, i.e. for every single
    ECDSA::verify, /root/reference/src/ecdsa.rs:57-59): the kernel's own phase functions (bn254_nonet.h) on a host box, the nine pairs of
    every exchange step run one after the other, give the accumulator machine's result coefficient for coefficient on every verify case
    and on random signed tuples (valid, foreign key, identity operands)"""
    import hashlib
    pair_lib.hp_nonet_check.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int)]
    n, words = 0, []
    for v in derived["verify_cases"]:
        if v["status"] not in (0, 9):
            continue
        st, h, _ = c.hash_to_g1(H(v["message_hex"]))
        w = ctypes.c_int(-1)
        assert pair_lib.hp_nonet_check(h, H(v["sig"]), H(v["pk"]), ctypes.byref(w)) == v["status"], v["name"]
        words.append(w.value)
        n += 1
    assert n >= 10
    g1, g2 = c.g1_generator(), c.g2_generator()
    for i in range(4):
        sk = hashlib.sha256(b"nonet-sk%d" % i).digest()
        _, h, _ = c.hash_to_g1(b"nonet-msg-%d" % i)
        sig, pk = c.g1_mul(h, sk), c.g2_mul(g2, sk)
        assert pair_lib.hp_nonet_check(h, sig, pk, None) == 0
        assert pair_lib.hp_nonet_check(h, sig, c.g2_mul(g2, hashlib.sha256(sk).digest()), None) == 9
        assert pair_lib.hp_nonet_check(h, bytes(64), pk, None) == 9
        assert pair_lib.hp_nonet_check(h, c.g1_mul(g1, sk), bytes(128), None) == 9
    assert set(words) <= {0, 1}


def test_lane_machine_schedule_matches_generic_loop(pair_lib, derived):
    """the Miller loop as the LANE MACHINE (bn254_lmiller.hip, default for batches <= 1 536, i.e. for every single ECDSA::verify,
    /root/reference/src/ecdsa.rs:49-64): the kernel's own stage functions and level tables (bn254_lmachine.h) on a host box, tick by tick
    — twist point with w = 3b' z two steps ahead, line product one step ahead, accumulator by the nonet product — give, under the final
    exponentiation, the generic loop's Gt value coefficient for coefficient (hp_lm_verify returns 246 otherwise) on every verify case, on
    random signed tuples, and with identity operands (pair A, pair B, both skipped)"""
    import hashlib
    pair_lib.hp_lm_verify.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p]
    n = 0
    for v in derived["verify_cases"]:
        if v["status"] not in (0, 9):
            continue
        st, h, _ = c.hash_to_g1(H(v["message_hex"]))
        assert pair_lib.hp_lm_verify(h, H(v["sig"]), H(v["pk"])) == v["status"], v["name"]
        n += 1
    assert n >= 10
    g1, g2 = c.g1_generator(), c.g2_generator()
    for i in range(3):
        sk = hashlib.sha256(b"lm-sk%d" % i).digest()
        _, h, _ = c.hash_to_g1(b"lm-msg-%d" % i)
        sig, pk = c.g1_mul(h, sk), c.g2_mul(g2, sk)
        assert pair_lib.hp_lm_verify(h, sig, pk) == 0
        assert pair_lib.hp_lm_verify(h, sig, c.g2_mul(g2, hashlib.sha256(sk).digest())) == 9
        assert pair_lib.hp_lm_verify(h, bytes(64), pk) == 9
        assert pair_lib.hp_lm_verify(h, c.g1_mul(g1, sk), bytes(128)) == 9
        assert pair_lib.hp_lm_verify(h, bytes(64), bytes(128)) == 0


def test_lane_machine_keyed_schedule_matches_keyed_loop(pair_lib, derived):
    """the KEYED form of the lane machine (k_miller_verify_lmk; registered public keys, include/bn254_hip.h: bn254_batch_verify_keyed for
    batches <= 1 536): both table lines scaled two steps ahead, their product one step ahead, an addition step in ONE tick — the host
    emulation gives, under the final exponentiation, the Gt value of the keyed pair loop (miller_loop_keyed) on every verify case and with
    skipped pairs (identity signature, identity key, both)"""
    pair_lib.hp_lm_verify_keyed.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p]
    n = 0
    for v in derived["verify_cases"]:
        if v["status"] not in (0, 9):
            continue
        st, h, _ = c.hash_to_g1(H(v["message_hex"]))
        assert pair_lib.hp_lm_verify_keyed(h, H(v["sig"]), H(v["pk"])) == v["status"], v["name"]
        n += 1
    assert n >= 10


def test_small_batch_pairing_schedule_gives_the_pairing(pair_lib, derived):
    """bn254_batch_pairing* for batches that cannot fill the chip (include/bn254_hip.h; the reference's pairing(), /root/reference/src/lib.rs
    re-export of bn::pairing): the lane machine's schedule with the fixed pair skipped, then the EXACT final-exponentiation program in the
    nonet schedule — canonical Gt bytes equal the fixtures' (independent big-integer model) and the lane-pair path's, identity operands
    give one"""
    for v in derived["pairing_gt"]:
        o = ctypes.create_string_buffer(384)
        pair_lib.hp_pairing_small_batch(H(v["g1"]), H(v["g2"]), o)
        assert o.raw.hex() == v["gt"]
    g1, g2 = c.g1_generator(), c.g2_generator()
    a, b = ctypes.create_string_buffer(384), ctypes.create_string_buffer(384)
    for p1, q2 in ((c.g1_mul(g1, (77).to_bytes(32, "big")), c.g2_mul(g2, (91).to_bytes(32, "big"))), (bytes(64), g2), (g1, bytes(128))):
        pair_lib.hp_pairing(p1, q2, a)
        pair_lib.hp_pairing_small_batch(p1, q2, b)
        assert a.raw == b.raw


def test_lane_machine_subgroup_ladder_agrees_with_lane_pairs(pair_lib, derived):
    """the G2 subgroup test of a decode (AffineG2::new, /root/reference/src/utils.rs:113) with its ladder [u]P in the lane machine's level
    tables (k_g2_subgroup_lm, batches <= 1 536) gives the lane-pair form's verdict — and the definition's ([r]P == O, big-integer model) —
    on points of G2, on random twist points outside it, on their cofactor-cleared and mixed forms, on a point of the twist's small
    order 10 069 (2q - r = 10 069 x a 241-bit number), and on the identity"""
    import random
    from oracle import bn254_model as m
    rnd = random.Random(77)

    def enc(p):
        return b"".join(v.to_bytes(32, "big") for v in (p[0][0], p[0][1], p[1][0], p[1][1]))
    pts = [(m.g2_mul(m.G2_GEN, rnd.randrange(1, m.R)), True) for _ in range(2)]
    n = 0
    while n < 3:
        x = (rnd.randrange(m.Q), rnd.randrange(m.Q))
        y = m.f2_sqrt(m.f2_add(m.f2_mul(m.f2_mul(x, x), x), m.B2))
        if y is None:
            continue
        n += 1
        p = (x, y)
        cleared = m.g2_mul(p, 2 * m.Q - m.R)
        pts += [(p, m.g2_in_subgroup(p)), (cleared, True), (m.g2_add(cleared, m.g2_mul(p, m.R)), False)]
        small = m.g2_mul(p, (2 * m.Q - m.R) // 10069 * m.R)
        if small is not None:
            assert m.g2_mul(small, 10069) is None
            pts.append((small, False))
    assert any(not w for _, w in pts) and len(pts) >= 12
    for p, want in pts:
        assert pair_lib.hp_g2_subgroup_both(enc(p)) == (3 if want else 0), want
    assert pair_lib.hp_g2_subgroup_both(H(derived["g2_not_in_subgroup"])) == 0
    assert pair_lib.hp_g2_subgroup_both(bytes(128)) == 3


def test_lane_machine_tables_are_well_formed():
    """the level tables of the lane machine, read from the header: within a level no slot is written twice and no product reads a slot that
    a product of the same level writes (the stages publish between fences, so a level's reads see the previous level's values); every
    wave writes only its own slots and the hand-over block of the step's parity"""
    import re
    text = open(os.path.join(ROOT, "bn254_amd", "csrc", "bn254_lmachine.h")).read()
    tables = re.findall(r"LM_TABLE (LM_\w+)\[(\d+)\]\[9\] = \{(.*?)\};", text, re.S)
    assert {t[0] for t in tables} == {"LM_T_INIT", "LM_T_DBL", "LM_T_ADD", "LM_L_DBL", "LM_L_ADD", "LM_L_PROD", "LM_K_EVAL", "LM_K_PROD"}
    for name, nlev, body in tables:
        levels = re.split(r"\},\s*\{", body.strip().strip("{}")) if int(nlev) > 1 else [body]
        assert len(levels) == int(nlev), name
        for lvl in levels:
            muls = re.findall(r"lm_mul\((LS_\w+(?: \+ \d)?), (LS_\w+), (LS_\w+)\)", lvl) + [(m[4], m[0], m[2]) for m in re.findall(r"LmP\{(LS_\w+), (LS_\w+), (LS_\w+), (LS_\w+), (LS_\w+)\}", lvl)]
            lins = re.findall(r"lm_lin\((LS_\w+(?: \+ \d)?),", lvl)
            outs = [m[0] for m in muls]
            assert len(set(outs)) == len(outs) and len(set(lins)) == len(lins), (name, outs, lins)
            assert not set(outs) & set(lins), (name, "a product and a linear output share a slot")
            for out, a, b in muls:
                assert a not in outs and b not in outs, (name, out, "reads a product output of its own level")
            own = "LS_T" if name.startswith("LM_T") else "LS_L"                     # (the keyed waves LA / LB use wave L's temporaries)
            for o in outs + lins:
                assert o.startswith(own) or o.startswith("LS_HO") or o.startswith("LS_LP") or (name == "LM_T_INIT" and o in ("LS_Q1X", "LS_Q1Y", "LS_Q2X", "LS_NPKY")), (name, o)


DRIVER = r'''
import ctypes, json, sys
root = sys.argv[1]
L = ctypes.CDLL(root + "/tests/hostsim/libhostsim_pair_bounds.so")
d = json.load(open(root + "/tests/golden/derived_vectors.json"))
H = bytes.fromhex
v = d["pairing_gt"][1]; o = ctypes.create_string_buffer(384); L.hp_pairing(H(v["g1"]), H(v["g2"]), o); assert o.raw.hex() == v["gt"]
g1 = (1).to_bytes(32, "big") + (2).to_bytes(32, "big")
for v in d["verify_cases"]:
    if v["status"] in (0, 9):
        L.hp_verify_decoded(g1, H(v["sig"]), H(v["pk"]))      # any G1 point exercises the same operation sequence
        assert L.hp_verify_keyed_decoded(g1, H(v["sig"]), H(v["pk"]), None) <= 9      # keyed verify: line table + table-driven loop
        assert L.hp_nonet_check(g1, H(v["sig"]), H(v["pk"]), None) <= 9               # the nonet schedule of the final exponentiation (bn254_nonet.h)
g2 = H(d["g2_generator"]); o = ctypes.create_string_buffer(384)
L.hp_pairing_product4(g1 * 4, g2 * 4, o)
o = ctypes.create_string_buffer(128)
k = json.load(open(root + "/tests/golden/reference_kats.json"))
assert L.hp_g2_decompress(H(k["g2_compressed_roundtrip"]["hex"]), o) == 0
L.hp_g2_sum_and_subgroup(g2 + g2 + bytes(128) + g2 + H(d["g2_not_in_subgroup"]), 5, o)
print("ok")
'''


def test_pair_layout_bounds_hold(pair_lib):
    p = subprocess.run([sys.executable, "-c", DRIVER, ROOT], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and p.stdout.strip() == "ok", (p.stdout[-500:], p.stderr[-2000:])


def test_fp6_lazy_reduction_variant_parity_and_bounds(pair_lib, derived):
    """the A/B build with the Fq6-level lazy reduction (bn254_field.h: fp6_mul_lazy — schoolbook Fq6 products, one reduction per output
    coefficient over six limb products, in fp12_sqr and fp12_mul_line2; -DBN_PAIR_FP6_LAZY for bn254_pair.hip, measured neutral on the
    GPU: profiles/r06_z_ab_fp6_lazy.log): every verify case gives the oracle's status through the generic AND the keyed loop, the keyed
    tables are the default build's words, and the same flows pass the interval tracker build (columns of SIX limb products, int32 limbs,
    value bounds; it aborts on a violation)."""
    lazy = ctypes.CDLL(os.path.join(ROOT, "tests", "hostsim", "libhostsim_pair_lazy.so"))
    W = 87 * 2 * 2 * 9
    n = 0
    for v in derived["verify_cases"]:
        if v["status"] not in (0, 9):
            continue
        st, h, _ = c.hash_to_g1(H(v["message_hex"]))
        assert lazy.hp_verify_decoded(h, H(v["sig"]), H(v["pk"])) == v["status"], v["name"]
        tab_lazy, tab_dflt = (ctypes.c_int32 * W)(), (ctypes.c_int32 * W)()
        assert lazy.hp_verify_keyed_decoded(h, H(v["sig"]), H(v["pk"]), tab_lazy) == v["status"], v["name"]
        assert pair_lib.hp_verify_keyed_decoded(h, H(v["sig"]), H(v["pk"]), tab_dflt) == v["status"]
        assert list(tab_lazy) == list(tab_dflt)
        n += 1
    assert n >= 10
    drv = r'''
import ctypes, json, sys
sys.path.insert(0, sys.argv[1])
from oracle import c_oracle as c
root = sys.argv[1]
L = ctypes.CDLL(root + "/tests/hostsim/libhostsim_pair_lazy_bounds.so")
d = json.load(open(root + "/tests/golden/derived_vectors.json"))
H = bytes.fromhex
W = 87 * 2 * 2 * 9
n = 0
for v in d["verify_cases"]:
    if v["status"] in (0, 9):
        st, h, _ = c.hash_to_g1(H(v["message_hex"]))
        assert L.hp_verify_decoded(h, H(v["sig"]), H(v["pk"])) == v["status"]
        assert L.hp_verify_keyed_decoded(h, H(v["sig"]), H(v["pk"]), (ctypes.c_int32 * W)()) == v["status"]
        n += 1
assert n >= 10
print("ok")
'''
    p = subprocess.run([sys.executable, "-c", drv, ROOT], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and p.stdout.strip() == "ok", (p.stdout[-500:], p.stderr[-2000:])


def test_nonet_schedule_bounds_hold(pair_lib):
    """the nonet schedule under the interval tracker, on its own (the DRIVER above runs it too): no 64-bit column, int32 limb or value
    bound can be exceeded in the nine-pair arrangement — identity operands included (data-independent control flow: one pass per
    flow is a proof).  The tracker build ABORTS on a violation."""
    drv = r'''
import ctypes, json, sys
root = sys.argv[1]
L = ctypes.CDLL(root + "/tests/hostsim/libhostsim_pair_bounds.so")
d = json.load(open(root + "/tests/golden/derived_vectors.json"))
H = bytes.fromhex
g1 = (1).to_bytes(32, "big") + (2).to_bytes(32, "big")
n = 0
for v in d["verify_cases"]:
    if v["status"] in (0, 9):
        assert L.hp_nonet_check(g1, H(v["sig"]), H(v["pk"]), None) <= 9
        n += 1
assert L.hp_nonet_check(g1, bytes(64), H(d["g2_generator"]), None) <= 9
assert L.hp_nonet_check(g1, g1, bytes(128), None) <= 9
assert n >= 10
print("ok")
'''
    p = subprocess.run([sys.executable, "-c", drv, ROOT], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and p.stdout.strip() == "ok", (p.stdout[-500:], p.stderr[-2000:])


def test_lane_machine_schedule_bounds_hold(pair_lib):
    """the lane machine's schedule of the Miller loop under the interval tracker: every product column, int32 limb and value bound of every
    level of waves T and L (fresh formulas: w = 3b' z, the expanded addition), of the general Fq12 product on Miller values (f^2 and f * L
    through nn_mul_*, whose site modes were searched on the final exponentiation's flows) and of both final-exponentiation schedules on
    the value it hands over — identity operands included (pair A, pair B, both skipped: data-dependent selects are followed with the
    union of both sides).  The tracker build ABORTS on a violation."""
    drv = r'''
import ctypes, json, sys
root = sys.argv[1]
L = ctypes.CDLL(root + "/tests/hostsim/libhostsim_pair_bounds.so")
d = json.load(open(root + "/tests/golden/derived_vectors.json"))
H = bytes.fromhex
g1 = (1).to_bytes(32, "big") + (2).to_bytes(32, "big")
vs = [v for v in d["verify_cases"] if v["status"] in (0, 9)][:2]       # the bounds depend on the sequence of operations only: one pass per flow is the proof
for v in vs:
    assert L.hp_lm_verify(g1, H(v["sig"]), H(v["pk"])) <= 9
assert L.hp_lm_verify(g1, bytes(64), H(d["g2_generator"])) <= 9
assert L.hp_lm_verify(g1, g1, bytes(128)) <= 9
assert L.hp_lm_verify(g1, bytes(64), bytes(128)) <= 9
for v in vs:                                                              # the keyed form: table lines of both pairs, one tick per addition step
    assert L.hp_lm_verify_keyed(g1, H(v["sig"]), H(v["pk"])) <= 9
assert L.hp_lm_verify_keyed(g1, bytes(64), H(d["g2_generator"])) <= 9
assert L.hp_lm_verify_keyed(g1, g1, bytes(128)) <= 9
assert L.hp_lm_verify_keyed(g1, bytes(64), bytes(128)) <= 9
o = ctypes.create_string_buffer(384)                                       # one pairing: fixed pair skipped, EXACT program in the nonet schedule
v = d["pairing_gt"][1]; L.hp_pairing_small_batch(H(v["g1"]), H(v["g2"]), o); assert o.raw.hex() == v["gt"]
assert L.hp_g2_subgroup_both(H(d["g2_generator"])) == 3 and L.hp_g2_subgroup_both(H(d["g2_not_in_subgroup"])) == 0 and L.hp_g2_subgroup_both(bytes(128)) == 3   # the subgroup ladder in wave T's tables
print("ok")
'''
    p = subprocess.run([sys.executable, "-c", drv, ROOT], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and p.stdout.strip() == "ok", (p.stdout[-500:], p.stderr[-2000:])


def _adversarial():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "adversarial_fe_vectors.json")) as f:
        return json.load(f)


def test_adversarial_limb_vectors_through_host_emulations(pair_lib):
    """final exponentiation inputs at the edge of the tracker's contract for a Miller value — non-canonical representatives, extreme
    balanced digits, extreme top limbs (tests/golden/gen_adversarial_fe.py; expected results from the big-integer model): the pair
    layout's accumulator machine gives the model's canonical Gt bytes (exact program) and verdict (check program), and the nonet
    schedule agrees with it coefficient for coefficient (hp_final_exp_limbs returns 248 otherwise)"""
    adv = _adversarial()
    pair_lib.hp_final_exp_limbs.argtypes = [ctypes.POINTER(ctypes.c_int32), ctypes.c_int, ctypes.c_char_p]
    fams = set()
    for v in adv["vectors"]:
        limbs = (ctypes.c_int32 * 108)(*v["limbs"])
        gt = ctypes.create_string_buffer(384)
        assert pair_lib.hp_final_exp_limbs(limbs, 1, gt) == v["status"]
        assert gt.raw.hex() == v["gt"]
        assert pair_lib.hp_final_exp_limbs(limbs, 0, None) == v["status"]
        fams.add((v["family"], v["status"]))
    assert fams == {("full", 9), ("one", 0)}


def test_adversarial_contract_is_inside_what_the_tracker_proves(pair_lib):
    """the fixture's contract (limb, top-limb and value maxima) lies inside the bounds the tracker derives for EVERY coefficient of a
    Miller value — the domain on which the final-exponentiation flows are proven — and the vectors fill it to within 0.1 %; the
    tracker build runs all of them (it aborts on a violation)."""
    drv = r'''
import ctypes, json, sys
root = sys.argv[1]
L = ctypes.CDLL(root + "/tests/hostsim/libhostsim_pair_bounds.so")
d = json.load(open(root + "/tests/golden/derived_vectors.json"))
adv = json.load(open(root + "/tests/golden/adversarial_fe_vectors.json"))
H = bytes.fromhex
g1 = (1).to_bytes(32, "big") + (2).to_bytes(32, "big")
c = adv["contract"]
out = (ctypes.c_double * 60)()
for v in d["verify_cases"]:
    if v["status"] in (0, 9):
        L.hp_miller_output_bounds(g1, H(v["sig"]), H(v["pk"]), out)
        for e in range(12):
            lo, hi, top, vlo, vhi = out[5 * e:5 * e + 5]
            assert -lo >= c["limb_abs_max"] and hi >= c["limb_abs_max"] and top >= c["top_abs_max"], (e, lo, hi, top)
            assert -vlo >= c["value_over_q_abs_max"] and vhi >= c["value_over_q_abs_max"], (e, vlo, vhi)
            assert top < 1.03 * c["top_abs_max"] and vhi < 1.03 * c["value_over_q_abs_max"]      # the contract is not lax either
L.hp_final_exp_limbs.argtypes = [ctypes.POINTER(ctypes.c_int32), ctypes.c_int, ctypes.c_char_p]
tops = 0
for v in adv["vectors"]:
    limbs = (ctypes.c_int32 * 108)(*v["limbs"])
    assert L.hp_final_exp_limbs(limbs, 1, None) == v["status"] and L.hp_final_exp_limbs(limbs, 0, None) == v["status"]
    tops = max(tops, max(abs(v["limbs"][9 * e + 8]) for e in range(12)))
    assert max(abs(x) for e in range(12) for x in v["limbs"][9 * e:9 * e + 8]) == c["limb_abs_max"] or v["family"] == "one"
assert tops == c["top_abs_max"]
print("ok")
'''
    p = subprocess.run([sys.executable, "-c", drv, ROOT], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and p.stdout.strip() == "ok", (p.stdout[-500:], p.stderr[-2000:])


# ---- the octet layout for small batches (bn254_trio.hip): its FORMULAS on the host -------------------------------------
# -DBN_TRIO_FORMULAS compiles the same pair emulation with the Fq12 product / squaring / two-line multiplication as the
# generic Karatsuba product the octet kernels spread over three lane pairs, and hp_verify_decoded additionally runs the
# round-structured Miller loop of the octet kernel (miller_verify_rounds: doubling / addition steps as rounds of four
# Fq2 products) and returns 254 when its accumulator differs from the generic loop's.
@pytest.fixture(scope="module")
def trio_lib():
    from tests import hostsim_binding
    hostsim_binding.build_all()
    return ctypes.CDLL(os.path.join(ROOT, "tests", "hostsim", "libhostsim_trio.so"))


def test_octet_formulas_verify_and_gt_match_oracle(trio_lib, derived):
    n = 0
    for v in derived["verify_cases"]:
        if v["status"] not in (0, 9):
            continue
        st, h, _ = c.hash_to_g1(H(v["message_hex"]))
        assert trio_lib.hp_verify_decoded(h, H(v["sig"]), H(v["pk"])) == v["status"], v["name"]
        n += 1
    assert n >= 10
    for v in derived["pairing_gt"]:
        out = ctypes.create_string_buffer(384)
        trio_lib.hp_pairing(H(v["g1"]), H(v["g2"]), out)
        assert out.raw.hex() == v["gt"]
    # random signed tuples, valid and with a foreign key: both Miller loops agree and give the oracle's status
    import hashlib
    g1, g2 = c.g1_generator(), c.g2_generator()
    for i in range(6):
        sk = hashlib.sha256(b"octet-sk%d" % i).digest()
        msg = b"octet-msg-%d" % i
        _, h, _ = c.hash_to_g1(msg)
        sig, pk = c.g1_mul(h, sk), c.g2_mul(g2, sk)
        assert trio_lib.hp_verify_decoded(h, sig, pk) == 0
        assert trio_lib.hp_verify_decoded(h, sig, c.g2_mul(g2, hashlib.sha256(sk).digest())) == 9
        assert trio_lib.hp_verify_decoded(h, bytes(64), pk) == 9          # identity signature
        assert trio_lib.hp_verify_decoded(h, c.g1_mul(g1, sk), bytes(128)) == 9   # identity key


def test_octet_formulas_bounds_hold(trio_lib):
    p = subprocess.run([sys.executable, "-c", DRIVER.replace("libhostsim_pair_bounds.so", "libhostsim_trio_bounds.so"), ROOT],
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and p.stdout.strip() == "ok", (p.stdout[-500:], p.stderr[-2000:])
