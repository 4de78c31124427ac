#!/usr/bin/env python3
"""what the calling convention costs per product: 3 219 dual products per lane with the leaf inlined into the loop (mode 2) against the
same products through the call (mode 3), 65 536 lane pairs, two waves per SIMD; clocks from the probe slot"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
import bn254_amd
from bn254_amd.engine import OPT_CLOCK_PROBE
from tests.datagen import make_verify_batch
eng = bn254_amd.Engine(0)
n = 65536
msgs, sigs, pks, exp = make_verify_batch(eng, n)
assert eng.batch_verify(msgs, sigs, pks) == exp
eng.set_option(OPT_CLOCK_PROBE, 1)
for rep in range(2):
    row = {}
    for mode, name in ((3, "called"), (2, "inlined"), (0, "miller_mix_called")):
        ms = eng.probe_leaf_floor(n, mode)
        row[name] = {"ms": round(ms, 3), "sclk_mhz": round(eng.last_clocks()["issue_probe"], 1)}
    print(json.dumps(row), flush=True)
