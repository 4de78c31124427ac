#!/bin/bash
# socket power and clocks while the headline bench runs (rocm-smi, sampled twice a second): is the part at its power cap under this instruction mix?
# usage (on the GPU box): tools/power_sample.sh <out.jsonl>
out=${1:-/dev/stdout}
rocm-smi --showmaxpower --showpower --showclocks 2>&1 | grep -v "^$" | head -30 > ${out}.idle.txt
python bench.py --steps 2500 --warmup 5 --no-cpu-baseline > ${out}.bench.json 2>/dev/null &
pid=$!
sleep 4
for i in $(seq 60); do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|Temperature \(Sensor (edge|junction)" | tr -s ' ' | tr '\n' ';'
  echo
  sleep 0.5
done > ${out}.samples.txt
wait $pid
tail -c 400 ${out}.bench.json | head -c 0
python - <<PY
import json
d=json.loads(open("${out}.bench.json").read().strip().splitlines()[-1])
print(json.dumps({"pairings_per_s": d["value"], "ms_per_step": d["ms_per_step"], "effective_sclk_mhz": {k: v for k, v in d["roofline"]["effective_sclk_mhz"].items() if k != "method"}}))
PY
