set -e
for b in 7 8; do
  ICP_GRID_BITS=$b ICP_GRID_DEBUG=gpurun_out/dbg$b.bin timeout -k 10 120 python bench.py --steps 20 --warmup 2 --no-cpu > gpurun_out/sw$b.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/sw$b.json')); print('bits',$b, 'ms/step', round(d['ms_per_step'],3), 'query_ms', round(d['roofline']['kernel_ms'],3))"
  python profiles/grid_wave_stats.py gpurun_out/dbg$b.bin | grep -E "waves: time|level 1 |level 2 "
  rm -f gpurun_out/dbg$b.bin
done
