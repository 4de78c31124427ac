import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import bn254_amd
from oracle import c_oracle as c
eng = bn254_amd.Engine(0)
g1, g2 = c.g1_generator(), c.g2_generator()
for n in (1, 16, 256, 1024):
    ks = [(i + 5).to_bytes(32, "big") for i in range(n)]
    ps, _ = eng.batch_g1_mul(g1 * n, b"".join(ks), n)
    qs, _ = eng.batch_g2_mul(g2 * n, b"".join(ks[::-1]), n)
    best = None
    for _ in range(5):
        t0 = time.perf_counter(); gt, st = eng.batch_pairing(ps, qs, n, 1); dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    t0 = time.perf_counter(); c.pairing(ps[:64], qs[:128], 1); cpu = time.perf_counter() - t0
    print(json.dumps({"n": n, "batch_pairing_ms": round(1e3 * best, 3), "cpu_one_pairing_ms": round(1e3 * cpu, 3)}), flush=True)
