#!/bin/bash
# instruction-cache counters of the headline's kernels for the library BN254_LIB names: args <tag>
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmci_$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/$tag -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/$tag.log 2>&1
  tail -1 $OUT/$tag.log | cut -c1-120
done
python3 $R/tests/pmc_to_json.py $OUT $OUT.json
