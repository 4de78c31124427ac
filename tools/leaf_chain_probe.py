"""How much of a product leaf's time is the wait for its predecessor?  bn254_probe_leaf_floor modes 0 (one dependent chain, loop in a real
function), 6 / 7 / 4 (the same product mix as 1 / 2 / 4 independent chains in one inlined loop), 1 / 5 (final-exponentiation mix, dependent /
4 chains), 3 / 2 (3 219 dual products called / inlined) at several batch sizes; clocks recorded per run.  Usage: python tools/leaf_chain_probe.py [n ...]"""
import json
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bn254_amd
from bn254_amd.engine import OPT_CLOCK_PROBE
from tests.datagen import make_verify_batch

eng = bn254_amd.Engine(0)
sizes = [int(a) for a in sys.argv[1:]] or [65536, 20480]
msgs, sigs, pks, _ = make_verify_batch(eng, max(sizes), corrupt_every=0)
eng.batch_verify(msgs, sigs, pks)                      # fills the workspace planes the probe reads
eng.set_option(OPT_CLOCK_PROBE, 1)
for n in sizes:
    row = {"n": n}
    for name, mode in (("miller_mix_dependent_loop_fn", 0), ("miller_mix_1_chain", 6), ("miller_mix_2_chains", 7), ("miller_mix_4_chains", 4),
                       ("fe_mix_dependent_loop_fn", 1), ("fe_mix_4_chains", 5), ("dual_only_called", 3), ("dual_only_inlined", 2)):
        eng.last_clocks()
        ms = min(eng.probe_leaf_floor(n, mode) for _ in range(2))
        row[name] = {"ms": round(ms, 3), "sclk_mhz": round(eng.last_clocks()["issue_probe"], 1)}
    print(json.dumps(row), flush=True)
