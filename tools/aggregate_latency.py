#!/usr/bin/env python3
"""latency of SMALL aggregate verifies (bn254_batch_aggregate_verify: T tuples over a pool of S signers, every tuple all signers) with the
pairing part on the small-batch kernels (default) and on the lane-pair kernels (BN254_OPT_TRIO_MAX_BATCH = 0)"""
import json
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
import bn254_amd
from bn254_amd.engine import OPT_TRIO_MAX_BATCH
from tests.datagen import sk_bytes

eng = bn254_amd.Engine(0)
S = 64
sks = [sk_bytes(300 + s) for s in range(S)]
pk_pool, _ = eng.batch_g2_mul(None, b"".join(sks), S, reduce_scalar=True)
for T in (1, 16, 256):
    msgs = [b"agg-lat-%d" % m for m in range(T)]
    sig_pool, _ = eng.batch_sign([msgs[m] for m in range(T) for _ in range(S)], b"".join(sks * T))
    row = {"tuples": T, "signers_per_tuple": S}
    for name, lim in (("lane_pair_kernels", 0), ("small_batch_kernels", 16384)):
        eng.set_option(OPT_TRIO_MAX_BATCH, lim)
        best = None
        for _ in range(5):
            t0 = time.perf_counter()
            got = eng.batch_aggregate_verify(msgs, pk_pool, sig_pool, list(range(T)), [list(range(S))] * T)
            dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
        row[name] = {"ok": got == bytes(T), "call_ms": round(1e3 * best, 3)}
    print(json.dumps(row), flush=True)
