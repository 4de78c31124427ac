# A/B: fused (default) vs split Miller for the library given in BN254_LIB (default: in-tree build), + PMC traffic of the fused run
for mode in "" "--split-miller"; do
  python bench.py --steps 4 --warmup 1 --no-cpu-baseline $mode 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms']; print('mode=$mode : %.2f Mpair/s  step %.2f ms  hash %.2f miller %.2f fexp %.2f' % (d['value']/1e6, d['ms_per_step'], k['hash_to_g1'], k['miller_loop'], k['final_exp']))"
done
