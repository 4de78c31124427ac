#!/usr/bin/env python3
"""Merge the PMC summaries of one profiling round into profiles/pmc_latest.json: the headline command's kernels (gpurun_out/pmc_<tag>.json,
written by `profile_round.sh <tag> b`) and the other workloads' sections (gpurun_out/<tag>/pmc.json, part c — a separate gpurun call, whose
box does not see the first file).  Both must name the same binary.   usage: pmc_merge.py <tag> [out.json]"""
import json
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(root, "profiles", "pmc_latest.json")
a = json.load(open(os.path.join(root, "gpurun_out", "pmc_%s.json" % tag)))
b = json.load(open(os.path.join(root, "gpurun_out", tag, "pmc.json")))
assert a["lib_sha256_16"] == b["lib_sha256_16"], (a["lib_sha256_16"], b["lib_sha256_16"])
a["workloads"] = b.get("workloads", {})
json.dump(a, open(out, "w"), indent=1, sort_keys=True)
print("wrote", out, a["lib_sha256_16"], len(a["kernels"]), "kernels +", sorted(a["workloads"]))
