set -e
for cap in 0 1 2 3; do
  ICP_GRID_CAP=$cap timeout -k 10 120 python bench.py --steps 20 --warmup 2 --no-cpu > gpurun_out/sw.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/sw.json')); print('cap $cap', 'ms/step', round(d['ms_per_step'],3), 'query_ms', round(d['roofline']['kernel_ms'],3), 'err', d['config']['last_err'])"
done
