#!/bin/bash
# rocprofv3 PMC passes for the bench (separate passes per counter group, kernel-trace only)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_FLAT SQ_ACTIVE_INST_ANY SQ_IFETCH SQ_INSTS_LDS GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/$tag -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline $2 > $OUT/$tag.log 2>&1
  tail -1 $OUT/$tag.log | cut -c1-200
done
python3 - <<PY
import csv,glob,collections,os
out="$OUT"
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out+"/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0]
        if not k.startswith("k_"): continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out+"/summary.txt","w") as fo:
    for k,cs in sorted(agg.items()):
        line=k+": "+", ".join("%s=%.4g(n=%d)"%(c,sum(v)/len(v),len(v)) for c,v in sorted(cs.items()))
        print(line); fo.write(line+"\n")
PY
