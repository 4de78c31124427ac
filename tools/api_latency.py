#!/usr/bin/env python3
"""latency of ONE item through every host-pointer entry point of the C ABI (best of 5 calls) beside the same operation on one core of the host
(oracle/bn254_oracle.c) — where a caller with a single item stands; DESIGN.md section 10.9"""
import json
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
import bn254_amd
from oracle import c_oracle as c
from tests.datagen import sk_bytes


def best(f, reps=5):
    b = None
    for _ in range(reps):
        t0 = time.perf_counter()
        r = f()
        dt = time.perf_counter() - t0
        b = dt if b is None or dt < b else b
    return round(1e3 * b, 3), r


eng = bn254_amd.Engine(0)
sk = sk_bytes(4242)
msg = b"api-latency"
g1, g2 = c.g1_generator(), c.g2_generator()
pk = c.public_key_g2(sk)
pk1 = c.public_key_g1(sk)
sig = c.sign(msg, sk)
rows = []


def row(name, gpu, cpu):
    g, _ = best(gpu)
    h, _ = best(cpu, 3)
    rows.append({"call": name, "gpu_ms": g, "one_host_core_ms": h})
    print(json.dumps(rows[-1]), flush=True)


row("verify", lambda: eng.batch_verify([msg], sig, pk, flags=0), lambda: c.verify(msg, sig, pk, 0))
row("verify + G2 subgroup check", lambda: eng.batch_verify([msg], sig, pk, flags=1), lambda: c.verify(msg, sig, pk, 1))
eng.register_keys(pk)
row("verify, registered key", lambda: eng.batch_verify_keyed([msg], sig, [0]), lambda: c.verify(msg, sig, pk, 0))
row("check_public_keys", lambda: eng.batch_check_public_keys(pk, pk1, 1), lambda: c.check_public_keys(pk, pk1, 0))
row("pairing", lambda: eng.batch_pairing(sig, pk, 1, 1), lambda: c.pairing(sig, pk, 1))
row("hash_to_g1", lambda: eng.batch_hash_to_g1([msg]), lambda: c.hash_to_g1(msg))
row("sign", lambda: eng.batch_sign([msg], sk), lambda: c.sign(msg, sk))
row("public key (G2 scalar multiplication)", lambda: eng.batch_g2_mul(None, sk, 1, reduce_scalar=True), lambda: c.public_key_g2(sk))
row("public key in G1 (fixed base)", lambda: eng.batch_g1_mul(None, sk, 1, reduce_scalar=True), lambda: c.g1_mul(g1, sk))
row("G1 scalar multiplication (variable base)", lambda: eng.batch_g1_mul(g1, sk, 1, reduce_scalar=True), lambda: c.g1_mul(g1, sk))
sig33 = c.g1_compress(sig)
from bn254_amd.api import PublicKey  # noqa: E402  (byte logic only: the sign byte of the compressed form)
pk65 = PublicKey(pk).to_compressed()
assert eng.batch_verify_compressed([msg], sig33, pk65) == b"\0"
row("verify from compressed encodings (one host core: decompress G1 + verify; no G2 decompression in the oracle's C API)",
    lambda: eng.batch_verify_compressed([msg], sig33, pk65), lambda: (c.g1_decompress(sig33), c.verify(msg, sig, pk, 1)))
row("G1 decompress", lambda: eng.batch_g1_decompress(sig33, 1), lambda: c.g1_decompress(sig33))
