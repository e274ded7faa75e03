# host cost of the sharded entry point on one GPU (profiles/r2_rigid_dist_rehearsal.txt): bash benchmarks/rehearse_r2.sh
# (RCCL may print to stdout: the JSON line is the last one)
for n in 1048576 262144 131072; do
for f in "" "--rehearse-comm"; do
timeout -k 10 200 python bench.py --no-cpu --no-nicp --steps 30 --warmup 2 --n $n $f 2>/dev/null | tail -1 > gpurun_out/rh.json && python -c "
import json;d=json.load(open('gpurun_out/rh.json'));print($n,'$f',round(d['ms_per_step'],4),round(d['extra']['repeats']['median_ms_per_step'],4),round(d['extra']['ms_per_step_without_setup'],4),d['config']['check']['idx_equal_bruteforce'])"
done; done
