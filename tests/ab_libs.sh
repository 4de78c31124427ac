#!/bin/bash
# A/B of library builds on one GPU box: tests/ab_libs.sh libA.so libB.so ...   (paths relative to bn254_amd/)
# Each library runs configs[1] (65 536 verifies) twice, interleaved, and 262 144 once; same box, same inputs.
fmt='import json,sys; d=json.loads(sys.stdin.read()); k=d["roofline"]["kernel_ms"]; print("%-28s n=%-7s %.2f Mpair/s  step %.2f ms  hash %.2f miller %.2f fexp %.2f exact=%s" % (sys.argv[1], sys.argv[2], d["value"]/1e6, d["ms_per_step"], k["hash_to_g1"], k["miller_loop"], k["final_exp"], d["config"]["bit_exact_vs_expected"]))'
for rep in 1 2; do for lib in "$@"; do
  BN254_LIB=$PWD/bn254_amd/$lib python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "$fmt" $lib 65536
done; done
for lib in "$@"; do
  BN254_LIB=$PWD/bn254_amd/$lib python bench.py --steps 3 --warmup 1 --no-cpu-baseline --batch 262144 2>/dev/null | python -c "$fmt" $lib 262144
done
