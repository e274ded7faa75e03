set -e
for cfg in "7 1" "8 1" "8 2" "6 1" "6 2"; do
  set -- $cfg
  ICP_GRID_BITS=$1 ICP_GRID_CAP=$2 timeout -k 10 120 python bench.py --steps 20 --warmup 2 --no-cpu > gpurun_out/sw.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/sw.json')); print('bits $1 cap $2', 'ms/step', round(d['ms_per_step'],3), 'query_ms', round(d['roofline']['kernel_ms'],3), 'err', d['config']['last_err'])"
done
