#!/bin/bash
# PMC passes over the octet-layout kernels at a small batch: args <tag> [n]
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/triopmc_$1; N=${2:-64}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_IFETCH" "SQ_INSTS_SALU SQ_INSTS_FLAT SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM" "SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/$tag -- python3 $R/tools/trio_run.py $N 3 > $OUT/$tag.log 2>&1
  tail -1 $OUT/$tag.log | cut -c1-120
done
python3 $R/tests/pmc_to_json.py $OUT $OUT.json
