#!/bin/bash
# kernel trace of small-batch verifies: small_trace.sh <n> <reps>  -> gpurun_out/small_trace_<n>/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/small_trace_$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/tools/trio_run.py $1 $2 > $O/run.log 2>&1
f=$(ls $O/*/*kernel_stats.csv | head -1); cut -d, -f1-4 $f | cut -c1-110 | head -14
python3 - <<PY
import csv,glob
f=glob.glob("$O/*/*kernel_trace.csv")[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last call: take the last 8 kernels
tail=rows[-9:]
t0=int(tail[0]["Start_Timestamp"])
for r in tail: print(r["Kernel_Name"][:40], (int(r["Start_Timestamp"])-t0)/1e3, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
PY
