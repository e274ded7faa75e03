# Host cost of the sharded entry points on ONE GPU with library-side RCCL (profiles/r3_dist_rehearsal.txt):
#   bash benchmarks/rehearse_r3.sh
# rigid: plain icp_rigid_run | sharded entry point, one-rank RCCL communicator, every collective issued (library) | same through callbacks
for n in 1048576 262144 131072; do
for f in "" "--rehearse-comm --comm rccl" "--rehearse-comm --comm callback"; do
timeout -k 10 200 python bench.py --no-cpu --no-nicp --steps 30 --warmup 2 --n $n $f 2>/dev/null | tail -n 1 > gpurun_out/rh.json && python -c "
import json;d=json.load(open('gpurun_out/rh.json'));print('rigid',$n,'[$f]','ms/step first %.4f median %.4f marginal %.4f' % (d['ms_per_step'],d['extra']['repeats']['median_ms_per_step'],d['extra']['ms_per_step_without_setup']),'check',d['config']['check_idx_equal_bruteforce'])" || exit 1
done; done
# NICP, C5 size on one GPU: nicp_solve | nicp_solve_dist forced, callbacks | library RCCL | library RCCL + graph replay of the sharded chunk
for f in "--single" "--force" "--force --rccl" "--force --rccl --graph"; do
g=0; case "$f" in *--graph) g=1; f="--force --rccl";; esac
ICP_NICP_GRAPH_DIST=$g timeout -k 10 200 python benchmarks/nicp_dist_bench.py --nx 448 --ny 448 --lm 2 --maxcg 300 $f 2>/dev/null | tail -n 1 > gpurun_out/rh.json && python -c "
import json;d=json.load(open('gpurun_out/rh.json'));print('nicp 200k [$f graph_dist=$g]', d['path'], 'cg', d['cg_iters'], 'us/cg-iteration %.1f' % d['us_per_cg_iter'], 'graph launches', d['graph_launches'], 'callbacks', d['comm_calls'], 'err %.9g' % d['err'])" || exit 1
done
for f in "--single" "--force" "--force --rccl" "--force --rccl --graph"; do
g=0; case "$f" in *--graph) g=1; f="--force --rccl";; esac
ICP_NICP_GRAPH_DIST=$g timeout -k 10 200 python benchmarks/nicp_dist_bench.py --nx 158 --ny 158 --lm 2 --maxcg 300 $f 2>/dev/null | tail -n 1 > gpurun_out/rh.json && python -c "
import json;d=json.load(open('gpurun_out/rh.json'));print('nicp 25k [$f graph_dist=$g]', d['path'], 'cg', d['cg_iters'], 'us/cg-iteration %.1f' % d['us_per_cg_iter'], 'graph launches', d['graph_launches'], 'callbacks', d['comm_calls'], 'err %.9g' % d['err'])" || exit 1
done
