#!/usr/bin/env python3
"""small batches with the final exponentiation on nine lane pairs per verify (BN254_OPT_NONET_MAX_BATCH) against the octet layout:
statuses vs the oracle and per-kernel times (HIP events inside the library), sizes from NONET_SIZES"""
import json
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (one HIP runtime per process: torch first)
import bn254_amd
from bn254_amd.engine import OPT_NONET_MAX_BATCH
from oracle import c_oracle as c
from tests.datagen import make_verify_batch

eng = bn254_amd.Engine(0)
eng.set_profiling(True)
sizes = [int(x) for x in os.environ.get("NONET_SIZES", "1,2,3,12,13,64,1024,3072,4096,8192").split(",")]
for n in sizes:
    msgs, sigs, pks, expected = make_verify_batch(eng, n, corrupt_every=3 if n > 2 else 0)
    want, _ = c.batch_verify(msgs, sigs, pks, flags=0, nthreads=16)
    row = {"n": n}
    for name, lim in (("octet", 0), ("nonet", 1 << 20)):
        eng.set_option(OPT_NONET_MAX_BATCH, lim)
        best, kms = None, None
        for _ in range(5):
            t0 = time.perf_counter()
            got = eng.batch_verify(msgs, sigs, pks, flags=0)
            dt = time.perf_counter() - t0
            if best is None or dt < best:
                best, kms = dt, eng.last_kernel_ms()
        row[name] = {"ok": got == want, "call_ms": round(1e3 * best, 3), "final_exp_ms": round(kms["final_exp"], 3), "miller_ms": round(kms["miller_loop"], 3)}
    print(json.dumps(row), flush=True)
