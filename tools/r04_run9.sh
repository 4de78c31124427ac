#!/bin/bash
set -o pipefail
out=gpurun_out/r04_h; mkdir -p $out
BN254_LIB=$GRAFT_REPO_ROOT/bn254_amd/ab/lib_csqrinl.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fp12 or verify_cases or batch_verify_vs_oracle or pairing_gt or octet" > $out/pytest_csqrinl.log 2>&1; rc=$?; echo "pytest(csqrinl) rc=$rc"; tail -2 $out/pytest_csqrinl.log
[ $rc -eq 0 ] || exit 1
bash tools/ab.sh 3 base csqrinl 2>&1 | tee $out/ab_inline_csqr_leaves.log
timeout -k 5 300 python tools/leaf_call_cost.py 2>/dev/null | tee $out/leaf_call_cost.jsonl
