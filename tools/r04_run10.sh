#!/bin/bash
set -o pipefail
out=gpurun_out/r04_h; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "hash or compressed or fp_ops or group_ops or bn256" > $out/pytest_powinl.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -2 $out/pytest_powinl.log
[ $rc -eq 0 ] || exit 1
bash tools/ab_workload.sh hash 2 callpow base 2>&1 | tee $out/ab_inline_pow_leaves.log
bash tools/ab_workload.sh verify-compressed 2 callpow base 2>&1 | tee -a $out/ab_inline_pow_leaves.log
bash tools/ab.sh 2 callpow base 2>&1 | tee -a $out/ab_inline_pow_leaves.log
