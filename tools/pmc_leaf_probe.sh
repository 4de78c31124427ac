#!/bin/bash
# PMC pass over the product-leaf floor probes (tools/leaf_chain_probe.py) and one verify: how much of a wave's time is WAITING in a kernel that is nothing but leaves?
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_leaf_probe; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $O -- python3 $R/tools/leaf_chain_probe.py 65536 > $O/run.log 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob("$O/*/*counter_collection.csv")[0]
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"].split("(")[0]
    if k not in ("k_leaf_floor_pair","k_miller_verify_pair","k_final_exp_pair"): continue
    key=(k, r.get("Dispatch_Id"))
    acc[key][r["Counter_Name"]]+=float(r["Counter_Value"])
order=sorted(acc, key=lambda x:int(x[1]))
for key in order:
    c=acc[key]
    if c.get("SQ_WAVE_CYCLES"): print(key[0], key[1], "wait_any %.3f wait_inst %.3f valu/wave %.0f salu/wave %.0f lds/wave %.0f" % (c["SQ_WAIT_ANY"]/c["SQ_WAVE_CYCLES"], c["SQ_WAIT_INST_ANY"]/c["SQ_WAVE_CYCLES"], c["SQ_INSTS_VALU"]/c["SQ_WAVES"], c["SQ_INSTS_SALU"]/c["SQ_WAVES"], c["SQ_INSTS_LDS"]/c["SQ_WAVES"]))
PY
