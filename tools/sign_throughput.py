"""Batch sign / key derivation throughput (N1 rows of SURVEY.md section 8f): n items through bn254_batch_sign / bn254_batch_g1_mul /
bn254_batch_g2_mul (host pointers), best of 3, beside the oracle on all host cores for a sample."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (HIP runtime order, tests/conftest.py)
import bn254_amd
from tests.datagen import sk_bytes
from oracle import c_oracle as c
eng = bn254_amd.Engine(0)
for n in (1, 1024, 65536):
    sks = b"".join(sk_bytes(j) for j in range(n))
    msgs = [b"msg-%d" % j for j in range(n)]
    g1 = ((1).to_bytes(32, "big") + (2).to_bytes(32, "big")) * n
    res = {"n": n}
    for name, fn in (("sign", lambda: eng.batch_sign(msgs, sks)), ("g1_mul", lambda: eng.batch_g1_mul(g1, sks, n, reduce_scalar=True)),
                     ("g2_keygen", lambda: eng.batch_g2_mul(None, sks, n, reduce_scalar=True))):
        fn()
        best = None
        for _ in range(3):
            t = time.perf_counter(); out = fn(); dt = time.perf_counter() - t
            best = dt if best is None or dt < best else best
        res[name + "_ms"] = round(1e3 * best, 3); res[name + "_per_s"] = round(n / best)
    if n == 1024:
        t = time.perf_counter(); [c.sign(m, sks[32 * j:32 * j + 32]) for j, m in enumerate(msgs[:256])]; res["oracle_sign_one_core_per_s"] = round(256 / (time.perf_counter() - t))
    print(json.dumps(res), flush=True)
