set -e
for v in 0 1 2 3 4 5 6; do
  ICP_GRID_VARIANT=$v timeout -k 10 120 python bench.py --steps 20 --warmup 2 --no-cpu > gpurun_out/sw.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/sw.json')); print('variant $v', 'ms/step', round(d['ms_per_step'],3), 'query_ms', round(d['roofline']['kernel_ms'],3), 'err', d['config']['last_err'])"
done
