#!/bin/bash
# same-box A/B of the clock probe inside the timed steps (round-5 form) against the stamp-free timed region (default since round 6):
#   ab_clock_probe.sh <reps> <out.jsonl>
reps=${1:-4}; out=${2:-$GRAFT_REPO_ROOT/gpurun_out/ab_clock_probe.jsonl}; rm -f $out
for r in $(seq $reps); do for mode in after timed off; do
  timeout -k 10 300 python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --clock-probe $mode 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);r=d['roofline'];k=r['kernel_ms']
print(json.dumps({'clock_probe':'$mode','rep':$r,'pairings_per_s':d['value'],'ms_per_step':d['ms_per_step'],'miller_ms':k['miller_loop'],'final_exp_ms':k['final_exp'],'hash_ms':k['hash_to_g1'],
 'sclk':{kk:vv for kk,vv in (r.get('effective_sclk_mhz') or {}).items() if kk not in ('method',)}}))" | tee -a $out
done; done
