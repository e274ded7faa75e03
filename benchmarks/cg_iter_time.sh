# wall time per CG iteration at C3 size (10k vertices) and C5 size (200k), one GPU
for a in "--nx 100 --ny 100 --lm 3 --maxcg 2000" "--nx 448 --ny 448 --lm 2 --maxcg 300"; do
  timeout -k 10 200 python3 benchmarks/nicp_dist_bench.py $a | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['workload'][:40], 'cg', d['cg_iters'], 'us/iter %.1f' % d['us_per_cg_iter'], 'err', d['err'])"
done
