set -e
for cfg in "7 0 1" "8 2 2" "8 1 2" "8 3 2" "8 4 2" "8 2 3"; do
  set -- $cfg
  ICP_GRID_BITS=$1 ICP_GRID_TMIN=$2 ICP_GRID_CAP=$3 timeout -k 10 120 python bench.py --steps 30 --warmup 2 --no-cpu > gpurun_out/sw.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/sw.json')); print('bits $1 tmin $2 cap $3', 'ms/step', round(d['ms_per_step'],4), 'search_ms', round(d['roofline']['kernel_ms'],4), d['roofline']['kernel_ms_parts'], 'err', d['config']['last_err'])"
done
