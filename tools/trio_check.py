#!/usr/bin/env python3
"""octet layout (BN254_OPT_TRIO_MAX_BATCH) vs the pair layout and the oracle on small batches: statuses + latency"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402  (one HIP runtime per process: torch first)
import bn254_amd  # noqa: E402
from bn254_amd.engine import OPT_TRIO_MAX_BATCH, OPT_TRIO_WAVE_ROLES  # noqa: E402
from oracle import c_oracle as c  # noqa: E402
from tests.datagen import make_verify_batch  # noqa: E402

eng = bn254_amd.Engine(0)
out = []
SIZES = [int(x) for x in os.environ.get("TRIO_SIZES", "1,7,64,200,1024,4096,8192").split(",")]
for n in SIZES:
    msgs, sigs, pks, expected = make_verify_batch(eng, n, corrupt_every=5 if n > 4 else 0)
    sigs = bytearray(sigs)
    if n >= 64:
        sigs[64 * 3:64 * 4] = bytes(64)                     # identity signature
        sigs[64 * 9 + 5] ^= 4                               # off-curve
    sigs = bytes(sigs)
    want, _ = c.batch_verify(msgs, sigs, pks, flags=0, nthreads=16)
    res = {"n": n}
    for name, lim, roles in (("pair", 0, 0), ("octet", 1 << 20, 0), ("roles", 1 << 20, 1), ("roles8", 1 << 20, 2)):
        eng.set_option(OPT_TRIO_MAX_BATCH, lim)
        eng.set_option(OPT_TRIO_WAVE_ROLES, roles)
        got = eng.batch_verify(msgs, sigs, pks, flags=0)
        res[name + "_ok"] = got == want
        if got != want:
            bad = [i for i in range(n) if got[i] != want[i]]
            res[name + "_bad"] = [(i, got[i], want[i]) for i in bad[:6]]
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            eng.batch_verify(msgs, sigs, pks, flags=0)
            ts.append(time.perf_counter() - t0)
        res[name + "_ms"] = round(1e3 * min(ts), 3)
    out.append(res)
    print(json.dumps(res), flush=True)
