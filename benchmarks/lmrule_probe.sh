# experiment: damping update rules of the LM loop on the whole C3 schedule, self-consistent mode
for r in 0 1 2 3 4; do
  echo "== ICP_NICP_LMRULE=$r"
  ICP_NICP_LMRULE=$r timeout -k 10 240 python3 benchmarks/nicp_bench.py --fixed 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('outer_iters','seconds','lm_iters','cg_iters','err_first','err_last','stopped_by')})" || exit 1
done
