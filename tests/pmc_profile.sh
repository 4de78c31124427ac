#!/bin/bash
# rocprofv3 PMC passes for bench.py: one pass per counter group (never combined with tracing domains
# other than --kernel-trace), then a per-kernel summary.
#   usage: pmc_profile.sh <tag> "<extra bench args>" [workload name] [batch] ["group1;group2;..."]
# The program follows `--` directly (python3 bench.py ...): no wrapper between the profiler and the process that touches the GPU.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_$1; mkdir -p $OUT
WL=${3:-verify}; BATCH=${4:-65536}
GROUPS_DEFAULT="FETCH_SIZE;WRITE_SIZE;SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY;SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_FLAT SQ_ACTIVE_INST_ANY SQ_IFETCH SQ_INSTS_LDS GRBM_GUI_ACTIVE;TCC_HIT_sum TCC_MISS_sum;TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"
GROUPS_LIST=${5:-$GROUPS_DEFAULT}
cd /tmp && export TMPDIR=/tmp
IFS=';' read -ra GRPS <<< "$GROUPS_LIST"
for grp in "${GRPS[@]}"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/$tag -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline $2 > $OUT/$tag.log 2>&1
  tail -1 $OUT/$tag.log | cut -c1-160
done
python3 $R/tests/pmc_to_json.py $OUT $OUT.json $BATCH $WL "$2"
