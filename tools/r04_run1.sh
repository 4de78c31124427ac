#!/bin/bash
# round 4, first GPU pass: the full -m gpu suite on the new binary, the headline line, and the A/Bs of the Phase-A changes
set -o pipefail
out=gpurun_out/r04_a; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee $out/pytest_rc.txt
tail -3 $out/pytest_gpu.log
timeout -k 10 300 python bench.py > $out/bench.json 2> $out/bench.err || exit 1
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_a/bench.json').read().strip().splitlines()[-1])
r=d['roofline']; print('headline %.3f M/s' % (d['value']/1e6), r['kernel_ms'], 'sclk', r.get('effective_sclk_mhz'), 'frac', round(r['frac'],4), 'frac@sclk', r.get('frac_at_effective_sclk'))
PY
timeout -k 10 600 python bench.py --workload aggregate --steps 4 --warmup 1 > $out/bench_aggregate.json 2> $out/bench_aggregate.err || exit 1
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_a/bench_aggregate.json').read().strip().splitlines()[-1])
print('aggregate %.3f M/s' % (d['value']/1e6), d['kernel_ms'], 'unsorted:', d['without_bucketing_by_message'])
PY
timeout -k 10 600 python bench.py --workload hash --steps 4 --warmup 1 > $out/bench_hash.json 2> $out/bench_hash.err || exit 1
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_a/bench_hash.json').read().strip().splitlines()[-1])
print('hash %.1f M/s' % (d['value']/1e6), d['kernel_ms'], 'finish frac', d['roofline']['frac'], 'whole', d['roofline']['whole_sequence']['frac'])
PY
for t in 0 1 2 4 8 0 4; do
  BN254_PINNED_STAGING=$t timeout -k 10 300 python bench.py --workload verify-host --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('verify-host pinned_threads=$t %.3f M/s' % (d['value']/1e6), d['kernel_ms'])" | tee -a $out/ab_pinned_staging.log
done
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('device-resident %.3f M/s' % (d['value']/1e6))" | tee -a $out/ab_pinned_staging.log
