for e in "0.3 0" "0.3 0.5" "0.3 0.7" "0.5 0.7" "0.4 0.6"; do
  set -- $e
  echo "== ICP_NICP_ETA=$1 ICP_NICP_ETA2=$2"
  ICP_NICP_ETA=$1 ICP_NICP_ETA2=$2 timeout -k 10 300 python3 benchmarks/nicp_bench.py --fixed 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('outer_iters','seconds','lm_iters','cg_iters','err_last')})" || exit 1
done
