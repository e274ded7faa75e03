for t in 1e-1 2e-1 5e-1; do
  echo "== ICP_NICP_GEO=1 cgtol $t"
  ICP_NICP_GEO=1 timeout -k 10 300 python3 benchmarks/nicp_bench.py --fixed --cgtol $t 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('outer_iters','seconds','lm_iters','cg_iters','err_first','err_last','stopped_by')})" || exit 1
done
