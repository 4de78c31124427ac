#!/usr/bin/env python3
"""Search for the cheapest safe configuration of the carry / weak-reduction sites (bn254_field.h: NS / NR).

TEST INFRASTRUCTURE (CPU only).  The tower and pairing code marks every place where a lazy value may have to be
carried (fp2_norm) or carried and weakly reduced (fp2_reduce_weak) before it meets a product, with a numbered site and
a safe default.  The bound-tracking host builds of the lane layouts (tests/hostsim/libhostsim_bounds.so,
libhostsim_pair_bounds.so, and libhostsim_trio_bounds.so with the formulas of the octet layout) read the mode of each site from a table at run time and, in "soft" mode, record a bound
violation instead of aborting.  The control flow of every formula is data-independent, so one pass of the probe flows
under the tracker is a proof for that configuration.

Greedy, most-executed sites first: try mode 0 (nothing); if the probe fails and the site's mode is 2, try 1 (carry
only); keep the cheapest passing mode.  The result is written to bn254_amd/csrc/bn254_norm_sites.h; afterwards
tests/test_bounds.py and tests/test_pair_layout.py (hard mode, all flows) must pass with it.

    python tests/norm_site_search.py            # search from the source defaults
    python tests/norm_site_search.py --check    # only verify the committed table
"""
import argparse
import ctypes
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
HS = os.path.join(ROOT, "tests", "hostsim")
OUT = os.path.join(ROOT, "bn254_amd", "csrc", "bn254_norm_sites.h")
N_SITES = 1024
COST = {0: 0, 1: 36, 2: 60}          # instructions per lane of a site in each mode (pair layout, 9 limbs)


class Lib:
    def __init__(self, name):
        self.L = ctypes.CDLL(os.path.join(HS, name))
        self.mode = (ctypes.c_byte * N_SITES).in_dll(self.L, "bn_site_mode")
        self.hits = (ctypes.c_uint * N_SITES).in_dll(self.L, "bn_site_hits")
        self.dflt = (ctypes.c_byte * N_SITES).in_dll(self.L, "bn_site_dflt")
        self.soft = ctypes.c_int.in_dll(self.L, "bn_bound_soft")
        self.failed = ctypes.c_int.in_dll(self.L, "bn_bound_failed")
        self.soft.value = 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    args = ap.parse_args()
    subprocess.check_call(["make", "-s", "-C", HS, "libhostsim_bounds.so", "libhostsim_pair_bounds.so", "libhostsim_trio_bounds.so"])
    from oracle import c_oracle as c
    d = json.load(open(os.path.join(ROOT, "tests", "golden", "derived_vectors.json")))
    H = bytes.fromhex
    classic, pair, trio = Lib("libhostsim_bounds.so"), Lib("libhostsim_pair_bounds.so"), Lib("libhostsim_trio_bounds.so")
    v = [x for x in d["verify_cases"] if x["status"] == 0][0]
    msg, sig, pk = H(v["message_hex"]), H(v["sig"]), H(v["pk"])
    _, h, _ = c.hash_to_g1(msg)
    g1, g2 = c.g1_generator(), c.g2_generator()
    ps = [c.g1_mul(g1, hashlib.sha256(b"s-a%d" % i).digest()) for i in range(4)]
    qs = [c.g2_mul(g2, hashlib.sha256(b"s-b%d" % i).digest()) for i in range(4)]
    cases = [x for x in d["verify_cases"] if x["status"] == 0][:3]
    n = len(cases)
    msgs = [H(x["message_hex"]) for x in cases]
    off = (ctypes.c_uint64 * (n + 1))()
    pos = 0
    for i, m in enumerate(msgs):
        off[i] = pos
        pos += len(m)
    off[n] = pos
    buf = ctypes.create_string_buffer
    pair.L.hp_lane_counts.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_ulonglong)]

    def probe():
        for lib in (classic, pair, trio):
            lib.failed.value = 0
        o = buf(384)
        assert trio.L.hp_verify_decoded(h, sig, pk) == 0 or trio.failed.value     # incl. the round-structured Miller loop
        if trio.failed.value:
            return False
        for s_, k_ in ((bytes(64), bytes(128)), (bytes(64), pk), (sig, bytes(128))):      # the skip paths of the rounds
            trio.L.hp_verify_decoded(h, s_, k_)
        if trio.failed.value:
            return False
        trio.L.hp_pairing(ps[0], qs[0], o)
        trio.L.hp_pairing_product4(b"".join(ps), b"".join(qs), o)
        if trio.failed.value:
            return False
        assert pair.L.hp_verify_decoded(h, sig, pk) == 0 or pair.failed.value
        pair.L.hp_pairing(ps[0], qs[0], o)
        pair.L.hp_nonet_check(h, sig, pk, None)           # the nonet schedule of the final exponentiation (bn254_nonet.h) shares sites 20..43, 170..179
        if pair.failed.value:
            return False
        pair.L.hp_pairing_product4(b"".join(ps), b"".join(qs), o)
        out6 = (ctypes.c_ulonglong * 6)()
        pair.L.hp_lane_counts(h, sig, pk, out6)
        if pair.failed.value:
            return False
        classic.L.hs_verify(msg, ctypes.c_uint64(len(msg)), sig, pk, 0)
        classic.L.hs_pairing(ps[0] + ps[1], qs[0] + qs[1], ctypes.c_uint64(2), 0, o, 0)
        classic.L.hs_pairing(ps[0], qs[0], ctypes.c_uint64(1), 0, o, 0)
        if classic.failed.value:
            return False
        st, gr = buf(n), buf(1)
        for fl in (0, 0x200):
            classic.L.hs_verify_randomized(b"".join(msgs), off, b"".join(H(x["sig"]) for x in cases), b"".join(H(x["pk"]) for x in cases),
                                           ctypes.c_uint64(n), fl, bytes(range(32)), st, gr)
        return not (classic.failed.value or pair.failed.value)

    def set_mode(i, m):
        classic.mode[i] = m
        pair.mode[i] = m
        trio.mode[i] = m

    assert probe(), "the committed / default configuration does not pass the tracker"
    if args.check:
        print("ok: committed table passes")
        return
    # weights: executions per probe in the pair layout (the shipped kernels), classic as a tie-break
    for lib in (classic, pair, trio):
        for i in range(N_SITES):
            lib.hits[i] = 0
    probe()
    weight = {i: pair.hits[i] * 4 + classic.hits[i] + trio.hits[i] * 2 for i in range(N_SITES) if pair.hits[i] or classic.hits[i] or trio.hits[i]}
    sites = sorted(weight, key=lambda i: -weight[i])
    print("%d sites in the probe flows" % len(sites))
    defaults = {}
    src = open(os.path.join(ROOT, "bn254_amd", "csrc", "bn254_field.h")).read() + open(os.path.join(ROOT, "bn254_amd", "csrc", "bn254_pairing.h")).read()
    result = {}
    saved = 0
    total_before = total_after = 0
    for i in sites:
        cur = pair.mode[i]
        dflt = max(pair.dflt[i], classic.dflt[i], trio.dflt[i])
        eff = cur if cur >= 0 else dflt
        best = None
        for m in (0, 1):
            if m >= eff:
                break
            set_mode(i, m)
            if probe():
                best = m
                break
        if best is None:
            set_mode(i, cur)
        else:
            result[i] = best
        now = best if best is not None else eff
        total_before += pair.hits[i] * COST[dflt]
        total_after += pair.hits[i] * COST[now]
        print("site %3d weight %6d default %d: %s" % (i, weight[i], dflt, "-> mode %d" % best if best is not None else "kept at %d" % eff), flush=True)
    assert probe()
    print("site instructions per probe in the pair layout: %d -> %d" % (total_before, total_after))
    # carry over overrides that were already committed for sites outside the probe flows
    lines = ["// GENERATED by tests/norm_site_search.py — the carry / weak-reduction sites of the tower and pairing code",
             "// (bn254_field.h: NS / NR) whose mode differs from the safe default written in the source.",
             "//   -1 = source default   0 = nothing   1 = carry (fp2_norm)   2 = carry + weak reduction (fp2_reduce_weak)",
             "// Every configuration recorded here has passed the bound tracker on all flows of tests/test_bounds.py and",
             "// tests/test_pair_layout.py (all lane layouts).",
             "#pragma once", "constexpr int bn_site_override(int id) {", "  switch (id) {"]
    final = {i: int(pair.mode[i]) for i in range(N_SITES) if pair.mode[i] >= 0 and pair.mode[i] != max(pair.dflt[i], classic.dflt[i], trio.dflt[i])}
    for m in (0, 1, 2):
        ids = sorted(i for i, mm in final.items() if mm == m)
        for k in range(0, len(ids), 16):
            lines.append("    " + " ".join("case %d:" % i for i in ids[k:k + 16]))
            lines.append("      return %d;" % m)
    lines += ["    default: return -1;", "  }", "}", ""]
    with open(OUT, "w") as f:
        f.write("\n".join(lines))
    print("wrote", OUT, "overrides:", len(final))


if __name__ == "__main__":
    main()
