#!/bin/bash
# quick PMC pass: args <tag>
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmcq_$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_IFETCH" "SQ_INSTS_SALU SQ_INSTS_FLAT SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/$tag -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/$tag.log 2>&1
  tail -1 $OUT/$tag.log | cut -c1-120
done
python3 $R/tests/pmc_to_json.py $OUT $OUT.json
