for w in 1 2 3 4; do for mode in "" "--split-miller"; do
  BN254_LIB=$PWD/bn254_amd/libbn254hip_w$w.so python bench.py --steps 4 --warmup 1 --no-cpu-baseline $mode 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms']; print('w=$w mode=$mode n=65536 : %.2f Mpair/s  step %.2f ms  hash %.2f miller %.2f fexp %.2f' % (d['value']/1e6, d['ms_per_step'], k['hash_to_g1'], k['miller_loop'], k['final_exp']))"
done; done
for w in 2 4; do BN254_LIB=$PWD/bn254_amd/libbn254hip_w$w.so python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch 262144 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms']; print('w=$w n=262144 : %.2f Mpair/s  step %.2f ms  hash %.2f miller %.2f fexp %.2f' % (d['value']/1e6, d['ms_per_step'], k['hash_to_g1'], k['miller_loop'], k['final_exp']))"; done
