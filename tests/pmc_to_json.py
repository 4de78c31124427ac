#!/usr/bin/env python3
"""Collapse the per-pass rocprofv3 counter CSVs written by tests/pmc_profile.sh into one JSON (mean per launch and kernel).
usage: pmc_to_json.py <pmc_dir> <out.json> [batch] [workload] [extra bench args]
With a workload name other than "verify" the kernels go under workloads[<name>] of an EXISTING out.json (merge), so that
profiles/pmc_latest.json carries one section per bench command; lib_sha256_16 names the binary that was measured."""
import collections
import csv
import glob
import hashlib
import json
import os
import sys

src, out = sys.argv[1], sys.argv[2]
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
workload = sys.argv[4] if len(sys.argv) > 4 else "verify"
extra = sys.argv[5] if len(sys.argv) > 5 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(src + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if k.startswith("k_"):
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
kernels = {}
for k, cs in agg.items():
    kernels[k] = {c: sum(v) / len(v) for c, v in cs.items()}
    kernels[k]["launches_averaged"] = max(len(v) for v in cs.values())
    kernels[k]["batch"] = batch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = os.environ.get("BN254_LIB", os.path.join(root, "bn254_amd", "libbn254hip.so"))
sha = hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16] if os.path.exists(lib) else None
command = "rocprofv3 --kernel-trace --pmc <group> -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline " + extra
if workload == "verify":
    res = {"command": command.strip(), "kernels": kernels, "lib_sha256_16": sha, "workloads": {}}
    if os.path.exists(out):
        try:
            old = json.load(open(out))
            if old.get("lib_sha256_16") == sha:
                res["workloads"] = old.get("workloads", {})
        except Exception:
            pass
else:
    res = json.load(open(out)) if os.path.exists(out) else {"kernels": {}, "workloads": {}, "lib_sha256_16": sha}
    assert res.get("lib_sha256_16") in (None, sha), "pmc_latest.json was measured on another binary: run the verify pass first"
    res["lib_sha256_16"] = sha
    res.setdefault("workloads", {})[workload] = {"command": command.strip(), "kernels": kernels}
json.dump(res, open(out, "w"), indent=1, sort_keys=True)
print("wrote", out, workload, len(kernels), "kernels")
