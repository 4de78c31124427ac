"""Limb/value bounds of the unsaturated Fq arithmetic, proven by interval bookkeeping.

The device headers compiled with -DBN_TRACK_BOUNDS carry, next to every field element, an interval
for its limbs, its top limb and its value; every operation propagates worst-case bounds and aborts
if a product's 64-bit column accumulator could overflow, a limb could leave int32, or a value could
outgrow the Montgomery range.  The bounds depend only on the operation sequence (control flow of the
pairing/hash/group code is data-independent), so one pass of each flow is a proof for that flow.
CPU only."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = r'''
import ctypes, json, sys
root = sys.argv[1]; which = sys.argv[2]
hs = ctypes.CDLL(root + "/tests/hostsim/libhostsim_bounds.so")
d = json.load(open(root + "/tests/golden/derived_vectors.json")); k = json.load(open(root + "/tests/golden/reference_kats.json"))
H = bytes.fromhex
buf = ctypes.create_string_buffer
if which == "fp":
    o = buf(32)
    for op in range(6):
        hs.hs_fp_op(op, (5).to_bytes(32, "big"), (7).to_bytes(32, "big"), o)
elif which == "hash":
    for v in d["hash_to_g1"][:4]:
        o = buf(64); t = ctypes.c_int(0); m = H(v["message_hex"])
        hs.hs_hash_to_g1(m, len(m), o, ctypes.byref(t)); assert o.raw.hex() == v["uncompressed"]
elif which == "pairing":
    v = d["pairing_gt"][1]; o = buf(384); hs.hs_pairing(H(v["g1"]), H(v["g2"]), 1, 0, o, 0); assert o.raw.hex() == v["gt"]
    a, b = d["pairing_gt"][1], d["pairing_gt"][2]
    hs.hs_pairing(H(a["g1"]) + H(b["g1"]), H(a["g2"]) + H(b["g2"]), 2, 0, o, 0)
elif which == "verify":
    for v in d["verify_cases"]:
        m = H(v["message_hex"]); st = hs.hs_verify(m, len(m), H(v["sig"]), H(v["pk"]), 0)
        assert st == v["status"] or "subgroup" in v["name"], v["name"]
    v = k["check_public_keys"][1]
elif which == "group":
    v = k["g1_add"][0]; o = buf(64); hs.hs_g1_add(H(v["x1"] + v["y1"]), H(v["x2"] + v["y2"]), o); assert o.raw.hex() == v["result"]
    for v in k["g1_mul"][:3]:
        hs.hs_g1_mul(H(v["x"] + v["y"]), H(v["scalar"]), 0, o); assert o.raw.hex() == v["result"]
    v = k["public_key_from_private_key"][0]; o = buf(128); hs.hs_g2_mul(None, H(v["private_key"]), 1, o); assert o.raw.hex() == v["uncompressed"]
    o2 = buf(128); hs.hs_g2_add(o.raw, o.raw, o2)
    v = k["sign"][0]; o = buf(64); m = H(v["message_hex"]); hs.hs_sign(m, len(m), H(v["private_key"]), o)
elif which == "msum":
    g = H(d["g2_generator"]); o = buf(128); hs.hs_g2_msum(g + g + bytes(128) + g, 4, o)
    g1 = (1).to_bytes(32, "big") + (2).to_bytes(32, "big"); o = buf(64); hs.hs_g1_msum(g1 * 5, 5, o)
elif which == "codec":
    o = buf(64); assert hs.hs_g1_decompress(H(k["sign"][0]["signature_compressed"]), o) == 0
    o = buf(128); assert hs.hs_g2_decompress(H(k["g2_compressed_roundtrip"]["hex"]), o) == 0
    assert hs.hs_g2_decompress(b"\x0c" + H(k["g2_compressed_roundtrip"]["hex"])[1:], o) == 3      # valid x, bad sign byte
elif which == "randomized":
    # one ragged group with invalid members (combined check fails -> exact kernels), then its valid members only
    for cases in (d["verify_cases"], [v for v in d["verify_cases"] if v["status"] == 0]):
        n = len(cases); msgs = [H(v["message_hex"]) for v in cases]
        off = (ctypes.c_uint64 * (n + 1))(); pos = 0
        for i, m in enumerate(msgs):
            off[i] = pos; pos += len(m)
        off[n] = pos
        st = buf(n); gr = buf(1)
        for fl in (0, 0x100, 0x200, 0x80000000):
            hs.hs_verify_randomized(b"".join(msgs), off, b"".join(H(v["sig"]) for v in cases), b"".join(H(v["pk"]) for v in cases), n, fl,
                                    bytes(range(32)), st, gr)
        assert list(st.raw) == [v["status"] if "subgroup" not in v["name"] else st.raw[i] for i, v in enumerate(cases)], list(st.raw)
        assert gr.raw == (b"\x01" if all(v["status"] == 0 for v in cases) else b"\x00")
elif which.startswith("unsafe"):
    hs.hs_unsafe_sequence(int(which[-1]))
elif which == "subgroup":
    v = d["verify_cases"][0]; m = H(v["message_hex"]); assert hs.hs_verify(m, len(m), H(v["sig"]), H(v["pk"]), 1) == 0
print("ok")
'''


@pytest.fixture(scope="module")
def bounds_lib():
    from tests import hostsim_binding
    hostsim_binding.build_all()


@pytest.mark.parametrize("flow", ["fp", "hash", "pairing", "verify", "group", "subgroup", "codec", "msum", "randomized"])
def test_bounds_hold(bounds_lib, flow):
    p = subprocess.run([sys.executable, "-c", DRIVER, ROOT, flow], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and p.stdout.strip() == "ok", (p.stdout[-500:], p.stderr[-2000:])


@pytest.mark.parametrize("which", [0, 1, 2])
def test_tracker_catches_unsafe_sequences(bounds_lib, which):
    """the checker is not vacuous: deliberately unsafe operation sequences abort with a BOUND VIOLATION"""
    p = subprocess.run([sys.executable, "-c", DRIVER, ROOT, "unsafe%d" % which], capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "BOUND VIOLATION" in p.stderr
