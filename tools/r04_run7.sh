#!/bin/bash
set -o pipefail
out=gpurun_out/r04_g; mkdir -p $out
NONET_SIZES=1,2,3,12,13,64,1024,3072,3073 timeout -k 10 500 python tools/nonet_check.py > $out/nonet_check.jsonl 2>/dev/null; cat $out/nonet_check.jsonl
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $out/pytest_gpu.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python bench.py --steps 10 --warmup 2 2>/dev/null > $out/bench.json; python -c "
import json;d=json.loads(open('$out/bench.json').read().strip().splitlines()[-1]);print('headline %.3f M/s' % (d['value']/1e6), d['roofline']['kernel_ms'], d['small_batch_latency'])"
