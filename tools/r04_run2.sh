#!/bin/bash
# round 4, second GPU pass: Jacobi with explicit borrow chains, hoisted table fetch in the aggregation kernel, the product-leaf floor
set -o pipefail
out=gpurun_out/r04_b; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "hash or aggregate or config3 or config5 or smoke or golden" > $out/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee $out/pytest_rc.txt
tail -3 $out/pytest_gpu.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err || exit 1
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_b/bench.json').read().strip().splitlines()[-1])
r=d['roofline']; print('headline %.3f M/s' % (d['value']/1e6), r['kernel_ms'], 'sclk', {k:v for k,v in r['effective_sclk_mhz'].items() if k!='method'}); print('leaf floor', r['product_leaf_floor'])
PY
timeout -k 10 600 python bench.py --workload aggregate --steps 4 --warmup 1 --no-cpu-baseline > $out/bench_aggregate.json 2> $out/bench_aggregate.err || exit 1
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_b/bench_aggregate.json').read().strip().splitlines()[-1])
print('aggregate %.3f M/s' % (d['value']/1e6), d['kernel_ms'], 'unsorted:', d['without_bucketing_by_message'])
PY
timeout -k 10 600 python bench.py --workload hash --steps 4 --warmup 1 --no-cpu-baseline > $out/bench_hash.json 2> $out/bench_hash.err || exit 1
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_b/bench_hash.json').read().strip().splitlines()[-1])
print('hash %.1f M/s' % (d['value']/1e6), d['kernel_ms'], 'finish frac', d['roofline']['frac'], 'whole', d['roofline']['whole_sequence']['frac'])
PY
