# kernel trace of the default bench after the fused loop (round 2, late).  From the repo root on the GPU box.
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 10 --warmup 2 --no-cpu --no-nicp --no-check --repeats 0"
timeout -s KILL 120 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r2c -o g -- $B > $R/gpurun_out/prof_r2c_bench.json 2>/dev/null
cd $R
python profiles/summarize_rocpd.py $(find gpurun_out/prof_r2c -name "*.db" | head -1) gpurun_out/r2c_kernel_stats_grid.txt > /dev/null
head -30 gpurun_out/r2c_kernel_stats_grid.txt | cut -c1-70,110-200
