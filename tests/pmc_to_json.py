#!/usr/bin/env python3
"""Collapse the per-pass rocprofv3 counter CSVs written by tests/pmc_profile.sh into one JSON
(mean per launch and kernel).  usage: pmc_to_json.py <pmc_dir> <out.json> [batch]"""
import collections
import csv
import glob
import json
import sys

src, out = sys.argv[1], sys.argv[2]
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(src + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if k.startswith("k_"):
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {"command": "rocprofv3 --kernel-trace --pmc <group> -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline", "kernels": {}}
for k, cs in agg.items():
    res["kernels"][k] = {c: sum(v) / len(v) for c, v in cs.items()}
    res["kernels"][k]["launches_averaged"] = max(len(v) for v in cs.values())
    res["kernels"][k]["batch"] = batch
json.dump(res, open(out, "w"), indent=1, sort_keys=True)
print("wrote", out)
