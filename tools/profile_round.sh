#!/bin/bash
# measurements behind profiles/<round>_<tag>_*: usage ROUND=r05 profile_round.sh <tag> <part>   (parts keep every gpurun call under its limit)
#   part a: bench lines of every workload (roofline + cpu_baseline each), batch sweep
#   part b: PMC passes of the headline command (all counter groups) + kernel-trace stats
#   part c: PMC passes (traffic + SQ group) and kernel-trace stats of the other workloads
tag=$1; part=$2
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${ROUND:-r04}_$tag; mkdir -p $O
SHORT="FETCH_SIZE;WRITE_SIZE;SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY"
if [ "$part" = a ]; then
  python $R/bench.py > $O/bench.json 2> $O/bench.err; echo bench done
  python $R/bench.py --workload pairing --steps 3 --warmup 1 > $O/bench_pairing.json 2>> $O/bench.err; echo pairing done
  rm -f $O/other_workloads.jsonl
  for w in verify-host verify-keyed verify-compressed hash aggregate; do python $R/bench.py --workload $w --steps 3 --warmup 1 2>>$O/bench.err | tail -1 >> $O/other_workloads.jsonl; echo $w done; done
  python $R/bench.py --workload verify-randomized --steps 3 --warmup 1 --batch 1048576 2>/dev/null | tail -1 > $O/randomized_1m.json; echo rand1m done
  python $R/bench.py --workload verify-keyed-randomized --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/keyed_randomized_1m.json; echo keyedrand done
  python $R/bench.py --workload verify-mgpu --gpus 1 --steps 25 --warmup 3 2>/dev/null | tail -1 > $O/bench_verify_mgpu_g1.json; echo mgpu1 done
  python $R/bench.py --workload verify-mgpu --gpus 4 --mgpu-devices 0,0,0,0 --steps 10 --warmup 2 2>/dev/null | tail -1 > $O/bench_verify_mgpu_4ctx_one_gpu.json; echo mgpu4 done
  python $R/tools/lm_check.py 2>/dev/null | grep "^{" > $O/small_batch_lane_machine_vs_wave_roles.jsonl; echo lm_check done
  bash $R/tools/small_trace.sh 1 20 > $O/single_verify_kernel_timeline.txt 2>&1; echo timeline done
  rm -f $O/batch_sweep.jsonl
  for b in 1 64 1024 4096 8192 16384 32768 65536 131072 262144 1048576; do python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --batch $b 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; fl=r.get('product_leaf_floor') or {}
print(json.dumps({'batch': $b, 'pairings_per_s': d['value'], 'ms_per_step': d['ms_per_step'], 'kernel_ms': r['kernel_ms'], 'effective_sclk_mhz': {k: v for k, v in (r.get('effective_sclk_mhz') or {}).items() if k != 'method'},
                  'product_leaf_floor_share': {k: v.get('floor_share_of_kernel') for k, v in fl.items() if isinstance(v, dict)}, 'product_leaf_floor_variants_ms': {k: {kk: vv.get('ms') for kk, vv in (v.get('variants') or {}).items()} for k, v in fl.items() if isinstance(v, dict)}, 'product_leaf_floor_ms': {k: v.get('ms_scaled_to_the_kernels_product_counts') for k, v in fl.items() if isinstance(v, dict)}}))" >> $O/batch_sweep.jsonl; done; echo sweep done
elif [ "$part" = s ]; then
  # the batch sweep alone (lane-pair sizes carry the product-leaf floors measured at THAT batch size)
  rm -f $O/batch_sweep.jsonl
  for b in 1 64 1024 4096 8192 16384 20480 24576 32768 49152 65536 131072 262144 1048576; do python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --batch $b 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; fl=r.get('product_leaf_floor') or {}
print(json.dumps({'batch': $b, 'pairings_per_s': d['value'], 'ms_per_step': d['ms_per_step'], 'kernel_ms': r['kernel_ms'], 'effective_sclk_mhz': {k: v for k, v in (r.get('effective_sclk_mhz') or {}).items() if k != 'method'},
                  'product_leaf_floor_share': {k: v.get('floor_share_of_kernel') for k, v in fl.items() if isinstance(v, dict)}, 'product_leaf_floor_variants_ms': {k: {kk: vv.get('ms') for kk, vv in (v.get('variants') or {}).items()} for k, v in fl.items() if isinstance(v, dict)}, 'product_leaf_floor_ms': {k: v.get('ms_scaled_to_the_kernels_product_counts') for k, v in fl.items() if isinstance(v, dict)}}))" >> $O/batch_sweep.jsonl; done; echo sweep done
elif [ "$part" = b ]; then
  bash $R/tests/pmc_profile.sh ${ROUND:-r04}_$tag "" verify 65536 > $O/pmc_verify.log 2>&1; cp $R/gpurun_out/pmc_${ROUND:-r04}_$tag.json $O/pmc.json; echo pmc verify done
  cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_verify -- python3 $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/stats_verify.err; echo stats done
elif [ "$part" = c ]; then
  # (a separate gpurun call: the verify summary of part b is not on this box; tools/pmc_merge.py joins the two afterwards)
  for spec in "verify-keyed:65536" "pairing:524288" "hash:16777216" "aggregate:1048576"; do
    w=${spec%%:*}; b=${spec##*:}
    OUTJ=$R/gpurun_out/pmc_${ROUND:-r04}_${tag}_$w.json; [ -f $O/pmc.json ] && cp $O/pmc.json $OUTJ
    EXTRA=""; [ "$w" = aggregate ] && EXTRA=" --agg-main-only"     # only the default route: the bench's side runs (direct additions, narrow tables) launch the same kernel
    bash $R/tests/pmc_profile.sh ${ROUND:-r04}_${tag}_$w "--workload $w$EXTRA" $w $b "$SHORT" > $O/pmc_$w.log 2>&1; cp $OUTJ $O/pmc.json; echo pmc $w done
    cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$w -- python3 $R/bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_${w}_under_rocprof.json 2> $O/stats_$w.err; echo stats $w done
  done
fi
