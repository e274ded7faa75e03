for t in 0.3 0.5 0.2; do
  echo "== accel default, ICP_NICP_ETA=$t"
  ICP_NICP_ETA=$t timeout -k 10 300 python3 benchmarks/nicp_bench.py --fixed 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('outer_iters','seconds','lm_iters','cg_iters','err_first','err_last','stopped_by')})" || exit 1
done
echo "== compat (plain LM)"; timeout -k 10 300 python3 benchmarks/nicp_bench.py 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('outer_iters','seconds','lm_iters','cg_iters','err_first','err_last','stopped_by')})"
