#!/bin/bash
# extra SQ counter passes (issue / wait breakdown).  usage: pmc_extra.sh <tag> "<extra bench args>"
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmcx_$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_WAVES SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_LEVEL_WAVES SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY" "SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_FLAT SQ_INSTS_FLAT_NO_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_BRANCH"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/$tag -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline $2 > $OUT/$tag.log 2>&1
done
python3 $R/tests/pmc_to_json.py $OUT $OUT.json
