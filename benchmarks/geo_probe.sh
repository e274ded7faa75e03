# experiment: geodesic acceleration of the LM step on the whole C3 schedule (self-consistent mode) and at 2 500 vertices
for g in 0 1; do
  echo "== ICP_NICP_GEO=$g"
  ICP_NICP_GEO=$g timeout -k 10 300 python3 benchmarks/nicp_bench.py --fixed 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('outer_iters','seconds','lm_iters','cg_iters','err_first','err_last','stopped_by')}); print([ (p['lm'],p['cg'],round(p['err'],6)) for p in d['per_outer']])" || exit 1
done
