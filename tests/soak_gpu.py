#!/usr/bin/env python3
"""Differential soak of the HIP path against the oracle (GPU box; not part of pytest).

    python tests/soak_gpu.py --seconds 300 --out gpurun_out/soak.json

Each round draws a batch of random size and message lengths, mutates a fraction of the signatures / keys
(bit flips, coordinates >= q, zeros, random bytes, off-subgroup keys, wrong-message signatures), and requires
bit-identical status bytes from every mode — exact on lane pairs, exact with one lane per verify, randomised with
128-bit / GLV / 64-bit scalars — and from the oracle (all host cores).  The oracle is the checker only."""
import argparse
import hashlib
import json
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
Q = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47


class SoakMismatch(AssertionError):
    pass


R_ORDER = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--out", default=None)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    try:
        res = soak(args)
    except SoakMismatch as e:
        print(e)
        sys.exit(1)
    print(json.dumps(res))
    if args.out:
        with open(args.out, "w") as f:
            json.dump(res, f, indent=1)


def soak(args):
    """run the differential for args.seconds with args.seed; returns the summary, raises SoakMismatch on a difference
    (tests/test_gpu_fullsize.py runs a slice of it in the driver-run suite)"""
    import bn254_amd
    from tests.conftest import ws_default
    from bn254_amd.engine import OPT_MAX_CHUNK, OPT_AGG_WIDE_MIN_TUPLES, OPT_AGG_SORT_BY_MSG, OPT_AGG_SUBSET_MIN_TUPLES, OPT_LM_MAX_BATCH, OPT_NONET_MAX_BATCH, OPT_NONET_WIDE, OPT_PAIR_LANES, OPT_RAND_MIN_BATCH, OPT_TRIO_MAX_BATCH, OPT_TRIO_WAVE_ROLES
    from oracle import c_oracle as c
    from tests.datagen import sk_bytes
    eng = bn254_amd.Engine(0)
    eng.set_option(OPT_RAND_MIN_BATCH, 0)     # always the randomised kernels
    derived = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "derived_vectors.json")))
    off_sub = bytes.fromhex(derived["g2_not_in_subgroup"])
    rnd = random.Random(args.seed)
    cores = len(os.sched_getaffinity(0))
    sks = [sk_bytes(j) for j in range(64)]
    pk_pool, _ = eng.batch_g2_mul(None, b"".join(sks), 64, reduce_scalar=True)
    t0, rounds, items, codes, extra = time.time(), 0, 0, {}, {}
    last_note = t0
    while time.time() - t0 < args.seconds:
        n = rnd.choice([1, 63, 64, 65, 257, 1000, 2048, 4097])
        msgs = [hashlib.sha256(b"soak%d/%d" % (rounds, i)).digest() * 5 for i in range(n)]
        msgs = [m[:rnd.choice([0, 1, 31, 32, 55, 56, 64, 100, 119, 120, 160])] for m in msgs]
        key = [rnd.randrange(64) for _ in range(n)]
        sigs, st = eng.batch_sign(msgs, b"".join(sks[k] for k in key))
        assert st == bytes(n)
        sigs = bytearray(sigs)
        pks = bytearray(b"".join(pk_pool[128 * k:128 * k + 128] for k in key))
        for i in range(n):
            kind = rnd.randrange(30)
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
                s[:] = rnd.randbytes(64)
            elif kind == 7:
                p[:] = rnd.randbytes(128)
            elif kind == 8:
                p[:] = off_sub
            elif kind == 9 and i > 0:
                s[:] = sigs[64 * (i - 1):64 * i]              # a valid signature of another message / key
        sigs, pks = bytes(sigs), bytes(pks)
        seed = rnd.randbytes(32)
        # keyed verify: the distinct (mutated) keys of this batch registered as a set, every item names its key by index;
        # a few indices point outside the set
        distinct = {}
        kidx = [distinct.setdefault(pks[128 * i:128 * i + 128], len(distinct)) for i in range(n)]
        key_set = b"".join(distinct.keys())
        oob = {i for i in range(n) if rnd.randrange(40) == 0}
        kidx_call = [len(distinct) + rnd.randrange(3) if i in oob else kidx[i] for i in range(n)]
        eng.register_keys(key_set)
        got_keyed = eng.batch_verify_keyed(msgs, sigs, kidx_call)
        want1, _ = c.batch_verify(msgs, sigs, pks, flags=1, nthreads=cores)
        sig_only, _ = c.batch_verify(msgs, sigs, bytes(128) * n, flags=1, nthreads=cores) if oob else (want1, None)
        for i in range(n):
            w = (sig_only[i] if sig_only[i] in (3, 4, 6) else 2) if i in oob else want1[i]
            if got_keyed[i] != w:
                raise SoakMismatch("MISMATCH keyed round %d n %d item %d got %d want %d" % (rounds, n, i, got_keyed[i], w))
        for fl in (0, 0x100, 0x200):                          # ... and the keyed randomised mode (same-key groups of 64, exact re-check of failing groups)
            got_kr = eng.batch_verify_keyed_randomized(msgs, sigs, kidx_call, seed, flags=fl)
            for i in range(n):
                w = (sig_only[i] if sig_only[i] in (3, 4, 6) else 2) if i in oob else want1[i]
                if got_kr[i] != w:
                    raise SoakMismatch("MISMATCH keyed randomised round %d n %d flags %x item %d got %d want %d" % (rounds, n, fl, i, got_kr[i], w))
        extra["keyed_tuples"] = extra.get("keyed_tuples", 0) + n
        for flags in (0, 1):
            want, _ = c.batch_verify(msgs, sigs, pks, flags=flags, nthreads=cores)
            # the small-batch layouts (defaults: the lane machine up to 1536 items, eight wave roles up to 16384), then the lane pairs for the same batch
            got = {"default": eng.batch_verify(msgs, sigs, pks, flags=flags)}
            eng.set_option(OPT_MAX_CHUNK, max(1, n // 3 + 1))  # round 6: the same batch in three slices inside the library (oversized-batch route)
            got["sliced"] = eng.batch_verify(msgs, sigs, pks, flags=flags)
            eng.set_option(OPT_MAX_CHUNK, 0)
            eng.set_option(OPT_LM_MAX_BATCH, 1 << 20)        # the Miller loop as the lane machine whatever the size (several passes above 768)
            got["lane_machine"] = eng.batch_verify(msgs, sigs, pks, flags=flags)
            eng.set_option(OPT_NONET_WIDE, 0)                # ... with the final exponentiation on nine lane pairs also up to 1024 items (default there: eighteen)
            got["lane_machine_nonet9"] = eng.batch_verify(msgs, sigs, pks, flags=flags)
            eng.set_option(OPT_NONET_WIDE, 1)
            eng.set_option(OPT_LM_MAX_BATCH, 0)              # ... and never: the wave-role / octet kernels at every size below
            got["roles8"] = eng.batch_verify(msgs, sigs, pks, flags=flags)
            for name, roles in (("roles4", 1), ("octet", 0)):
                eng.set_option(OPT_TRIO_WAVE_ROLES, roles)
                got[name] = eng.batch_verify(msgs, sigs, pks, flags=flags)
            eng.set_option(OPT_TRIO_WAVE_ROLES, 2)
            eng.set_option(OPT_NONET_MAX_BATCH, 0)           # the final exponentiation of the small-batch path in the octet layout (default: nine lane pairs up to 3072)
            got["roles8_octet_fe"] = eng.batch_verify(msgs, sigs, pks, flags=flags)
            eng.set_option(OPT_NONET_MAX_BATCH, 1 << 20)     # ... and on nine lane pairs whatever the size (several passes above 3072)
            got["roles8_nonet_fe"] = eng.batch_verify(msgs, sigs, pks, flags=flags)
            eng.set_option(OPT_NONET_MAX_BATCH, ws_default("NONET_MAX_BATCH_DEFAULT"))
            eng.set_option(OPT_TRIO_MAX_BATCH, 0)
            got["pair"] = eng.batch_verify(msgs, sigs, pks, flags=flags)
            eng.set_option(OPT_TRIO_MAX_BATCH, ws_default("TRIO_MAX_BATCH_DEFAULT"))
            eng.set_option(OPT_PAIR_LANES, 0)
            got["single"] = eng.batch_verify(msgs, sigs, pks, flags=flags)
            eng.set_option(OPT_PAIR_LANES, 1)
            eng.set_option(OPT_LM_MAX_BATCH, ws_default("LM_MAX_BATCH_DEFAULT"))
            for name, fl in (("rand128", 0), ("rand_glv", 0x200), ("rand64", 0x100)):
                got[name] = eng.batch_verify_randomized(msgs, sigs, pks, seed, flags=flags | fl)[0]
            for name, g in got.items():
                bad = [i for i in range(n) if g[i] != want[i]]
                if bad:
                    raise SoakMismatch("MISMATCH %s round %d flags %d n %d %r %r" % (name, rounds, flags, n, bad[:5], [(g[i], want[i]) for i in bad[:5]]))
            for b in want:
                codes[b] = codes.get(b, 0) + 1
        if rounds % 8 == 0:
            # pairing API (canonical Gt bytes; both kernel families) and check_public_keys vs the oracle
            m = 24
            ks = [rnd.randrange(1, 1 << 250).to_bytes(32, "big") for _ in range(2 * m)]
            g1 = (1).to_bytes(32, "big") + (2).to_bytes(32, "big")
            ps, st1 = eng.batch_g1_mul(g1 * m, b"".join(ks[:m]), m)
            qs, st2 = eng.batch_g2_mul(None, b"".join(ks[m:]), m)
            assert st1 == bytes(m) and st2 == bytes(m)
            # the group operations behind them, themselves against the oracle (round 6: variable-base G1 multiplication and sign over the
            # endomorphism, key derivation from the comb tables): generator and random bases, raw 256-bit and reduced scalars, signatures
            g2gen = c.g2_generator()
            for j in range(m):
                if ps[64 * j:64 * j + 64] != c.g1_mul(g1, ks[j]) or qs[128 * j:128 * j + 128] != c.g2_mul(g2gen, ks[m + j]):
                    raise SoakMismatch("MISMATCH g1_mul / g2 key derivation round %d item %d" % (rounds, j))
            raw = [rnd.randrange(1 << 256).to_bytes(32, "big") for _ in range(m)]
            for reduce in (False, True):
                vb, stv = eng.batch_g1_mul(ps, b"".join(raw), m, reduce_scalar=reduce)
                for j in range(m):
                    kk = (int.from_bytes(raw[j], "big") % R_ORDER).to_bytes(32, "big") if reduce else raw[j]
                    if stv[j] != 0 or vb[64 * j:64 * j + 64] != c.g1_mul(ps[64 * j:64 * j + 64], kk):
                        raise SoakMismatch("MISMATCH variable-base g1_mul round %d item %d reduce %d" % (rounds, j, reduce))
            smsgs = [rnd.randbytes(rnd.randrange(0, 90)) for _ in range(m)]
            ssig, sst = eng.batch_sign(smsgs, b"".join(raw))
            for j in range(m):
                if sst[j] != 0 or ssig[64 * j:64 * j + 64] != c.sign(smsgs[j], raw[j]):
                    raise SoakMismatch("MISMATCH sign round %d item %d" % (rounds, j))
            extra["g1_mul_sign_keygen_items"] = extra.get("g1_mul_sign_keygen_items", 0) + 5 * m
            gt, stg = eng.batch_pairing(ps, qs, m, 1)          # a small batch: the small-batch kernels (lane machine, eighteen lane pairs) ...
            eng.set_option(OPT_LM_MAX_BATCH, 0)
            gt_lp, stg_lp = eng.batch_pairing(ps, qs, m, 1)    # ... and the lane-pair kernels
            eng.set_option(OPT_LM_MAX_BATCH, ws_default("LM_MAX_BATCH_DEFAULT"))
            if gt != gt_lp or stg != stg_lp:
                raise SoakMismatch("MISMATCH pairing small-batch kernels vs lane pairs round %d" % rounds)
            for j in range(m):
                want_gt = c.pairing(ps[64 * j:64 * j + 64], qs[128 * j:128 * j + 128])
                if gt[384 * j:384 * j + 384] != want_gt:
                    raise SoakMismatch("MISMATCH pairing Gt round %d item %d" % (rounds, j))
            pk1, _ = eng.batch_g1_mul(g1 * m, b"".join(ks[m:]), m)        # matching G1 keys for the G2 keys above
            bad_pk1 = bytearray(pk1)
            bad_pk1[64:128] = pk1[:64]
            got_cpk = eng.batch_check_public_keys(qs, bytes(bad_pk1), m)
            want_cpk = bytes(c.check_public_keys(qs[128 * j:128 * j + 128], bytes(bad_pk1[64 * j:64 * j + 64]), flags=0) for j in range(m))
            if got_cpk != want_cpk or got_cpk[1] != 9 or got_cpk[0] != 0:
                raise SoakMismatch("MISMATCH check_public_keys round %d %r %r" % (rounds, list(got_cpk), list(want_cpk)))
            # aggregate verification over pools (config-3 shape): random subsets, duplicates (P + P), one wrong-message row
            M, S = 3, 12
            amsgs = [b"soak-agg-%d-%d" % (rounds, j) for j in range(M)]
            apk_pool = pk_pool[:128 * S]
            asig_pool, st = eng.batch_sign([amsgs[j] for j in range(M) for _ in range(S)], b"".join(sks[:S]) * M)
            assert st == bytes(M * S)
            tuples = []
            for _ in range(40):
                k = rnd.randrange(0, 9)
                lst = [rnd.randrange(S) for _ in range(k)]
                if lst and rnd.randrange(4) == 0:
                    lst.append(lst[0])                       # duplicate signer
                tuples.append((rnd.randrange(M), lst))
            for _ in range(170):                             # dense lists, and enough tuples per message for the signature tables too
                lst = rnd.sample(range(S), rnd.randrange(3, S + 1))
                if rnd.randrange(5) == 0:
                    lst.append(lst[-1])
                tuples.append((rnd.randrange(M), lst))
            got_a = eng.batch_aggregate_verify(amsgs, apk_pool, asig_pool, [t[0] for t in tuples], [t[1] for t in tuples])
            eng.set_option(OPT_AGG_SUBSET_MIN_TUPLES, 1)     # ... forced on for this small batch: both routes must agree with the oracle
            got_b = eng.batch_aggregate_verify(amsgs, apk_pool, asig_pool, [t[0] for t in tuples], [t[1] for t in tuples])
            eng.set_option(OPT_AGG_SORT_BY_MSG, 0)           # ... and the same without the device-side bucketing by message
            got_c = eng.batch_aggregate_verify(amsgs, apk_pool, asig_pool, [t[0] for t in tuples], [t[1] for t in tuples])
            eng.set_option(OPT_AGG_SORT_BY_MSG, 1)
            eng.set_option(OPT_AGG_WIDE_MIN_TUPLES, 1)       # ... and with the WIDENED key table (16 signers per entry; the signatures stay on their 4-signer tables here)
            got_w = eng.batch_aggregate_verify(amsgs, apk_pool, asig_pool, [t[0] for t in tuples], [t[1] for t in tuples])
            # both tables widened: >= 512 tuples on ONE message; against the narrow route on the same tuples, which the oracle checks below for the first 210
            one_msg = [(0, t[1]) for t in tuples] + [(0, rnd.sample(range(S), rnd.randrange(0, S + 1))) for _ in range(330)]
            got_ww = eng.batch_aggregate_verify(amsgs, apk_pool, asig_pool, [t[0] for t in one_msg], [t[1] for t in one_msg])
            eng.set_option(OPT_AGG_WIDE_MIN_TUPLES, 0)
            got_wn = eng.batch_aggregate_verify(amsgs, apk_pool, asig_pool, [t[0] for t in one_msg], [t[1] for t in one_msg])
            # REGISTERED pools (round 6): the tables built once (subset sums forced on, the widened ones off), then the tuples alone, twice in two orders
            eng.set_option(OPT_AGG_WIDE_MIN_TUPLES, 0)
            eng.register_pools(amsgs, apk_pool, asig_pool, expect_tuples=len(tuples))
            got_r = eng.batch_aggregate_verify_registered([t[0] for t in tuples], [t[1] for t in tuples])
            rev = list(range(len(tuples)))[::-1]
            got_rr = eng.batch_aggregate_verify_registered([tuples[j][0] for j in rev], [tuples[j][1] for j in rev])
            if got_r != got_b or got_rr != bytes(got_b[j] for j in rev):
                raise SoakMismatch("MISMATCH aggregate registered pools round %d %r %r" % (rounds, list(got_b), list(got_r)))
            eng.set_option(OPT_AGG_WIDE_MIN_TUPLES, ws_default("AGG_WIDE_MIN_TUPLES_DEFAULT"))
            eng.set_option(OPT_AGG_SUBSET_MIN_TUPLES, ws_default("AGG_SUBSET_MIN_TUPLES_DEFAULT"))
            if got_w != got_b or got_ww != got_wn:
                raise SoakMismatch("MISMATCH aggregate widened tables round %d %r %r" % (rounds, list(got_b), list(got_w)))
            for j in rnd.sample(range(len(one_msg)), 24):    # the one-message batch against the oracle on a sample
                asig, apk = bytes(64), bytes(128)
                for sg in one_msg[j][1]:
                    asig = c.g1_add(asig, asig_pool[64 * sg:64 * sg + 64])
                    apk = c.g2_add(apk, apk_pool[128 * sg:128 * sg + 128])
                if got_ww[j] != c.verify(amsgs[0], asig, apk, 0):
                    raise SoakMismatch("MISMATCH aggregate widened tables vs oracle round %d tuple %d" % (rounds, j))
            if got_c != got_b:
                raise SoakMismatch("MISMATCH aggregate bucketed / caller order round %d %r %r" % (rounds, list(got_b), list(got_c)))
            if got_b != got_a:
                raise SoakMismatch("MISMATCH aggregate routes round %d %r %r" % (rounds, list(got_a), list(got_b)))
            for j, (mi, lst) in enumerate(tuples):
                asig, apk = bytes(64), bytes(128)
                for sg in lst:
                    asig = c.g1_add(asig, asig_pool[64 * (mi * S + sg):64 * (mi * S + sg) + 64])
                    apk = c.g2_add(apk, apk_pool[128 * sg:128 * sg + 128])
                want_a = c.verify(amsgs[mi], asig, apk, 0)
                if got_a[j] != want_a:
                    raise SoakMismatch("MISMATCH aggregate round %d tuple %d %r %d %d" % (rounds, j, lst, got_a[j], want_a))
            extra["aggregate_tuples"] = extra.get("aggregate_tuples", 0) + len(tuples)
            extra["pairings"] = extra.get("pairings", 0) + m
            extra["check_public_keys"] = extra.get("check_public_keys", 0) + m
        rounds += 1
        items += n
        if time.time() - last_note > 60:          # a progress line a minute (a silent GPU command is taken to be hung)
            last_note = time.time()
            print("soak: %d s, %d rounds, %d tuples, no mismatch" % (last_note - t0, rounds, items), flush=True)
    from bn254_amd import _native
    lib_sha = hashlib.sha256(open(_native.LIB_PATH, "rb").read()).hexdigest()[:16]
    res = {"lib_sha256_16": lib_sha, "rounds": rounds, "tuples": items, "comparisons": items * 2 * 10, "seconds": round(time.time() - t0, 1), "oracle_threads": cores,
           "status_histogram": {str(k): v for k, v in sorted(codes.items())}, "mismatches": 0, "seed": args.seed, "also_compared": extra,
           "modes": ["exact, the batch in three slices inside the library (BN254_OPT_MAX_CHUNK)", "aggregate verify on registered pools (every eighth round)", "keyed (registered keys, once per round with the subgroup check)", "keyed randomised 128-bit / 64-bit / GLV (once per round)", "exact, defaults (lane machine up to 1536, eight wave roles above; final exponentiation on nine lane pairs up to 3072)", "exact, Miller loop as the lane machine at every size", "exact, eight wave roles + final exponentiation on nine lane pairs", "... + octet final exponentiation", "... + nine lane pairs at every size", "exact, four wave roles", "exact, lane groups of one wave (octet)", "exact on lane pairs", "exact, one lane per verify", "randomised 128-bit", "randomised GLV", "randomised 64-bit"],
           "flags": [0, 1]}
    return res


if __name__ == "__main__":
    main()
