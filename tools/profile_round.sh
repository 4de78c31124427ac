#!/bin/bash
# final-binary measurements for profiles/r02_<tag>_*: bench line, kernel stats, PMC, other workloads, batch sweep
tag=$1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_$tag; mkdir -p $O
python $R/bench.py > $O/bench.json 2> $O/bench.err; echo bench done
python $R/bench.py --workload pairing --steps 3 --warmup 1 > $O/bench_pairing.json 2>> $O/bench.err; echo pairing done
for w in verify-host verify-compressed hash aggregate; do python $R/bench.py --workload $w --steps 3 --warmup 1 2>/dev/null | tail -1 >> $O/other_workloads.jsonl; echo $w done; done
python $R/bench.py --workload verify-randomized --steps 3 --warmup 1 --batch 1048576 2>/dev/null | tail -1 > $O/randomized_1m.json; echo rand1m done
for b in 1 64 1024 4096 8192 16384 32768 65536 131072 262144 1048576; do python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --batch $b 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({'batch': $b, 'pairings_per_s': d['value'], 'ms_per_step': d['ms_per_step'], 'kernel_ms': d['roofline']['kernel_ms']}))" >> $O/batch_sweep.jsonl; done; echo sweep done
bash $R/tests/pmc_profile.sh r02_$tag "" > $O/pmc.log 2>&1; echo pmc done
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/stats.log 2>&1; echo stats done
