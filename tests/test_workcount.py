"""bench.py's algorithmic-work constants (Montgomery products per kernel stage) are the exact counts
of the device arithmetic source (instrumented host compilation).  CPU only."""
import ctypes
import importlib.util
import os

from tests import hostsim_binding as hs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_stage_counts_match_bench_constants(derived):
    b = _bench()
    L = hs.lib()
    L.hs_verify_stage_counts.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_ulonglong)]
    for v in derived["verify_cases"][:6]:
        out = (ctypes.c_ulonglong * 4)()
        msg = bytes.fromhex(v["message_hex"])
        L.hs_verify_stage_counts(msg, len(msg), bytes.fromhex(v["sig"]), bytes.fromhex(v["pk"]), out)
        assert out[0] == b.FP_MUL_DECODE
        assert out[2] == b.FP_MUL_MILLER
        assert out[3] == b.FP_MUL_FINAL_EXP
        tries = hs.hash_to_g1(msg)[2]
        assert out[1] == b.FP_MUL_HASH_FILTER * tries + b.FP_MUL_HASH_FINISH, (out[1], tries)


def test_lane_product_counts_for_bench(derived):
    """per-lane product counts of the pair kernels (dual-accumulated / single) — what bench.py prices the multiplier
    instructions with; committed as profiles/lane_product_counts.json and re-derived here"""
    import json
    import subprocess
    from oracle import c_oracle as c
    from tests import hostsim_binding
    hostsim_binding.build_all()
    L = ctypes.CDLL(os.path.join(ROOT, "tests", "hostsim", "libhostsim_pair.so"))
    L.hp_lane_counts.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_ulonglong)]
    seen = set()
    for v in [v for v in derived["verify_cases"] if v["status"] in (0, 9)][:4]:
        st, h, _ = c.hash_to_g1(bytes.fromhex(v["message_hex"]))
        out = (ctypes.c_ulonglong * 6)()
        L.hp_lane_counts(h, bytes.fromhex(v["sig"]), bytes.fromhex(v["pk"]), out)
        seen.add(tuple(out))
    assert len(seen) == 1                                   # data-independent control flow
    o = seen.pop()
    counts = {"k_miller_verify_pair": {"dual": o[0], "single": o[1]}, "k_final_exp_pair": {"dual": o[2], "single": o[3]},
              "k_miller_var_pair": {"dual": o[4], "single": o[5]}}
    path = os.path.join(ROOT, "profiles", "lane_product_counts.json")
    if os.environ.get("BN254_WRITE_COUNTS") == "1" or not os.path.exists(path):
        with open(path, "w") as f:
            json.dump(counts, f, indent=1, sort_keys=True)
    assert json.load(open(path)) == counts
    assert _bench().lane_product_counts() == counts


def test_other_workload_product_counts_for_bench(derived):
    """the product counts bench.py prices the other workloads' rooflines with: keyed Miller loop (per verify) and the group
    operations of the aggregation kernel (per tuple), from the instrumented host compilation of the device source"""
    from oracle import c_oracle as c
    from tests import hostsim_binding
    hostsim_binding.build_all()
    L = ctypes.CDLL(os.path.join(ROOT, "tests", "hostsim", "libhostsim_pair.so"))
    b = _bench()
    v = [v for v in derived["verify_cases"] if v["status"] == 0][0]
    out = (ctypes.c_ulonglong * 2)()
    L.hp_lane_counts_keyed(c.hash_to_g1(bytes.fromhex(v["message_hex"]))[1], bytes.fromhex(v["sig"]), bytes.fromhex(v["pk"]), out)
    assert 3 * out[0] + 2 * out[1] == b.FP_MUL_MILLER_KEYED          # a dual product = 3 Fq products (Karatsuba), a square / scaling = 2
    o5 = (ctypes.c_ulonglong * 5)()
    L.hp_group_op_counts(o5)
    assert (o5[0], o5[1]) == (b.FP_MUL_G1_MADD, b.FP_MUL_G2_MADD) and o5[2] + o5[3] + o5[4] == b.FP_MUL_AGG_TAIL


def test_host_example_compiles():
    """the C++ host mirror (bn254_amd/host/bn254.hpp) compiles and links against the C ABI"""
    import subprocess
    from bn254_amd import _native
    _native.build()
    host = os.path.join(ROOT, "bn254_amd", "host")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", os.path.join(host, "example.cpp"), "-L" + os.path.join(ROOT, "bn254_amd"),
                           "-lbn254hip", "-Wl,-rpath," + os.path.join(ROOT, "bn254_amd"), "-o", os.path.join(host, "example")])
