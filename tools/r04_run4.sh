#!/bin/bash
set -o pipefail
out=gpurun_out/r04_d; mkdir -p $out
timeout -k 10 300 python bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_d/bench.json').read().strip().splitlines()[-1])
r=d['roofline']; print('headline %.3f M/s' % (d['value']/1e6), r['kernel_ms']); print(json.dumps(r['final_exp_split'], indent=1)); print({k:(v['floor_share_of_kernel'] if isinstance(v,dict) else None) for k,v in r['product_leaf_floor'].items()})
PY
