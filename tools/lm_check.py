#!/usr/bin/env python3
"""small batches with the Miller loop as the LANE MACHINE (BN254_OPT_LM_MAX_BATCH; bn254_lmiller.hip) against the eight wave roles:
statuses vs the oracle (valid, corrupted and identity operands) and per-kernel times (HIP events inside the library), sizes from LM_SIZES"""
import json
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (one HIP runtime per process: torch first)
import bn254_amd
from bn254_amd.engine import OPT_LM_MAX_BATCH
from oracle import c_oracle as c
from tests.datagen import make_verify_batch

eng = bn254_amd.Engine(0)
eng.set_profiling(True)
sizes = [int(x) for x in os.environ.get("LM_SIZES", "1,2,3,4,7,64,256,768,1024,3072").split(",")]
for n in sizes:
    msgs, sigs, pks, expected = make_verify_batch(eng, n, corrupt_every=3 if n > 2 else 0)
    if n >= 7:                                   # identity operands: pair A / pair B skipped, both
        sigs = bytearray(sigs); pks = bytearray(pks)
        sigs[64 * 1:64 * 2] = bytes(64)
        pks[128 * 2:128 * 3] = bytes(128)
        sigs[64 * 4:64 * 5] = bytes(64); pks[128 * 4:128 * 5] = bytes(128)
        sigs, pks = bytes(sigs), bytes(pks)
    want, _ = c.batch_verify(msgs, sigs, pks, flags=0, nthreads=16)
    row = {"n": n}
    for name, lim in (("roles8", 0), ("lane_machine", 1 << 20)):
        eng.set_option(OPT_LM_MAX_BATCH, lim)
        best, kms = None, None
        for _ in range(5):
            t0 = time.perf_counter()
            got = eng.batch_verify(msgs, sigs, pks, flags=0)
            dt = time.perf_counter() - t0
            if best is None or dt < best:
                best, kms = dt, eng.last_kernel_ms()
        row[name] = {"ok": got == want, "call_ms": round(1e3 * best, 3), "final_exp_ms": round(kms["final_exp"], 3), "miller_ms": round(kms["miller_loop"], 3)}
        if got != want:
            row[name]["first_diff"] = [(i, g, w) for i, (g, w) in enumerate(zip(got, want)) if g != w][:5]
    print(json.dumps(row), flush=True)

# the KEYED form (registered public keys): k_miller_verify_lmk on the keys' line tables against the generic small-batch kernels on the expanded keys
from tests.datagen import D, sk_bytes
K = 8
sks = [sk_bytes(900 + j) for j in range(K)]
pk_pool, _ = eng.batch_g2_mul(None, b"".join(sks), K, reduce_scalar=True)
keys = bytearray(pk_pool)
keys[128 * 5:128 * 6] = bytes(128)                     # the identity key: pair A skipped
eng.register_keys(bytes(keys))
for n in [s for s in sizes if s <= 1024]:
    msgs = [D("lmk", i) for i in range(n)]
    kidx = [i % K for i in range(n)]
    sigs, _ = eng.batch_sign(msgs, b"".join(sks[k] for k in kidx))
    sigs = bytearray(sigs)
    for i in range(2, n, 3):
        sigs[64 * i:64 * i + 64] = sigs[64 * (i - 1):64 * i]
    if n >= 7:
        sigs[64 * 6:64 * 7] = bytes(64)                # identity signature: pair B skipped
    idx_call = list(kidx)
    if n >= 4:
        idx_call[3] = K + 1                            # out of range
    want = bytearray(c.batch_verify(msgs, bytes(sigs), b"".join(bytes(keys[128 * k:128 * k + 128]) for k in kidx), flags=1, nthreads=16)[0])
    if n >= 4:
        so = c.batch_verify(msgs[3:4], bytes(sigs[192:256]), bytes(128), flags=1)[0][0]
        want[3] = so if so in (3, 4, 6) else 2
    row = {"n": n, "keyed": True}
    for name, lim in (("expanded_keys_roles8", 0), ("lane_machine_keyed", 1 << 20)):
        eng.set_option(OPT_LM_MAX_BATCH, lim)
        best, kms = None, None
        for _ in range(5):
            t0 = time.perf_counter()
            got = eng.batch_verify_keyed(msgs, bytes(sigs), idx_call)
            dt = time.perf_counter() - t0
            if best is None or dt < best:
                best, kms = dt, eng.last_kernel_ms()
        row[name] = {"ok": bytes(got) == bytes(want), "call_ms": round(1e3 * best, 3), "final_exp_ms": round(kms["final_exp"], 3), "miller_ms": round(kms["miller_loop"], 3)}
        if bytes(got) != bytes(want):
            row[name]["first_diff"] = [(i, g, w) for i, (g, w) in enumerate(zip(got, want)) if g != w][:5]
    print(json.dumps(row), flush=True)
