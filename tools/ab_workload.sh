#!/bin/bash
# same-box A/B of library variants on another bench workload: ab_workload.sh <workload> <reps> name1 name2 ...   (prints value, ms/step and kernel_ms when present)
w=$1; reps=$2; shift; shift
for r in $(seq $reps); do for v in "$@"; do
  BN254_LIB=$GRAFT_REPO_ROOT/bn254_amd/ab/lib_$v.so timeout -k 10 300 python bench.py --workload $w --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$v', '$w', round(d['value']/1e6,3), 'M/s', round(d['ms_per_step'],3), 'ms', {k: round(x,3) for k,x in d.get('kernel_ms',{}).items()})"
done; done
