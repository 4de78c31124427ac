#!/bin/bash
set -o pipefail
out=gpurun_out/r04_e; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "hash or config5 or smoke or golden or verify_cases or batch_verify_vs_oracle" > $out/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $out/pytest_gpu.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python bench.py --workload hash --steps 4 --warmup 1 --no-cpu-baseline > $out/bench_hash.json 2> $out/bench_hash.err || exit 1
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_e/bench_hash.json').read().strip().splitlines()[-1])
print('hash %.1f M/s' % (d['value']/1e6), d['ms_per_step'], d['kernel_ms'])
PY
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('headline %.3f M/s' % (d['value']/1e6), d['roofline']['kernel_ms'])"
