#!/bin/bash
set -o pipefail
out=gpurun_out/r04_h; mkdir -p $out
# correctness first: the parity tests on the asm-leaf variant
BN254_LIB=$GRAFT_REPO_ROOT/bn254_amd/ab/lib_asmleaf.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fp12 or miller or verify_cases or batch_verify_vs_oracle or pairing_gt or octet or fuzz or group_ops or compressed" > $out/pytest_asmleaf.log 2>&1; rc=$?; echo "pytest(asmleaf) rc=$rc"; tail -3 $out/pytest_asmleaf.log
[ $rc -eq 0 ] || exit 1
bash tools/ab.sh 3 base asmleaf 2>&1 | tee $out/ab_asm_mul_leaf.log
./bn254_amd/csrc/microbench/leaf_variants > $out/leaf_variants_microbench.jsonl 2>/dev/null
