#!/usr/bin/env python3
"""latency of small batches through the compressed-encoding entry point (bn254_batch_verify_compressed)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
import bn254_amd
from bn254_amd.api import PublicKey, Signature
from tests.datagen import make_verify_batch
eng = bn254_amd.Engine(0)
for n in (1, 64, 1024):
    msgs, sigs, pks, expected = make_verify_batch(eng, n, corrupt_every=5 if n > 4 else 0)
    s33 = b"".join(Signature(sigs[64 * i:64 * i + 64]).to_compressed() for i in range(n))
    cache = {}
    def comp(b):
        if b not in cache:
            cache[b] = PublicKey(b).to_compressed()
        return cache[b]
    p65 = b"".join(comp(pks[128 * i:128 * i + 128]) for i in range(n))
    got = eng.batch_verify_compressed(msgs, s33, p65)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); eng.batch_verify_compressed(msgs, s33, p65); ts.append(time.perf_counter() - t0)
    ts2 = []
    for _ in range(5):
        t0 = time.perf_counter(); eng.batch_verify(msgs, sigs, pks, flags=1); ts2.append(time.perf_counter() - t0)
    print(json.dumps({"n": n, "ok": got == expected, "compressed_ms": round(1e3 * min(ts), 3), "uncompressed_with_subgroup_check_ms": round(1e3 * min(ts2), 3)}), flush=True)
