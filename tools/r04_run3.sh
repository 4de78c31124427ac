#!/bin/bash
set -o pipefail
out=gpurun_out/r04_c; mkdir -p $out
timeout -k 10 300 python bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err || exit 1
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_c/bench.json').read().strip().splitlines()[-1])
r=d['roofline']; print('headline %.3f M/s' % (d['value']/1e6), r['kernel_ms']); print(json.dumps(r['product_leaf_floor'], indent=1))
PY
cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/stats_compressed -- python3 $GRAFT_REPO_ROOT/bench.py --workload verify-compressed --steps 5 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/bench_compressed.json 2> $GRAFT_REPO_ROOT/$out/stats_compressed.err
cd $GRAFT_REPO_ROOT; f=$(find $out/stats_compressed -name "*kernel_stats.csv" | head -1); head -12 $f | cut -c1-200
