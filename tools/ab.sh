#!/bin/bash
# same-box A/B of library variants: ab.sh <reps> name1 name2 ...
reps=$1; shift
for r in $(seq $reps); do for v in "$@"; do
  BN254_LIB=$GRAFT_REPO_ROOT/bn254_amd/ab/lib_$v.so timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);k=d['roofline']['kernel_ms'];c=d['roofline'].get('effective_sclk_mhz') or {};print('$v', round(d['value']/1e6,3), 'M/s  miller %.3f fe %.3f hash %.3f' % (k['miller_loop'],k['final_exp'],k['hash_to_g1']), ' clock MHz miller %s fe %s' % (c.get('miller_loop'), c.get('final_exp')))"
done; done
