# Round-2 profile set for the default bench (grid search).  Run on the GPU box from the repo root:
#   bash profiles/run_r2_profiles.sh
# Each rocprofv3 pass is a run of its own (kernel trace only, or one counter group).
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 10 --warmup 2 --no-cpu --no-nicp --no-check --repeats 0"
timeout -s KILL 120 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r2b -o g -- $B > $R/gpurun_out/prof_r2b_bench.json 2>/dev/null
timeout -s KILL 90 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_r2b_fetch -o p -- $B > /dev/null 2>&1
timeout -s KILL 90 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_r2b_write -o p -- $B > /dev/null 2>&1
timeout -s KILL 90 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --kernel-trace --output-format csv -d $R/gpurun_out/pmc_r2b_tcc -o p -- $B > /dev/null 2>&1
timeout -s KILL 90 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_r2b_sq -o p -- $B > /dev/null 2>&1
cd $R
python profiles/summarize_rocpd.py $(find gpurun_out/prof_r2b -name "*.db" | head -1) gpurun_out/r2b_kernel_stats_grid.txt > /dev/null
head -12 gpurun_out/r2b_kernel_stats_grid.txt | cut -c1-60,110-200
for d in fetch write tcc sq; do echo "== $d"; python profiles/pmc_summary.py $(find gpurun_out/pmc_r2b_$d -name "*counter_collection.csv" | head -1) grid_; done > gpurun_out/r2b_pmc.txt
cat gpurun_out/r2b_pmc.txt
