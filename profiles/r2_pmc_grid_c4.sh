# Round-2 PMC passes over the C4-size search on ONE GPU (2^24 queries x 2^24 targets).
# Run on the GPU box from the repo root:  bash profiles/r2_pmc_grid_c4.sh
# Each rocprofv3 pass is a run of its own (kernel trace only, or one counter group).
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --n 16777216 --steps 6 --warmup 1 --repeats 0 --no-cpu --no-nicp --no-check"
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_c4 -o g -- $B > $R/gpurun_out/prof_c4_bench.json 2>/dev/null
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_c4_fetch -o p -- $B > /dev/null 2>&1
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_c4_write -o p -- $B > /dev/null 2>&1
timeout -s KILL 150 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --kernel-trace --output-format csv -d $R/gpurun_out/pmc_c4_tcc -o p -- $B > /dev/null 2>&1
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_c4_sq -o p -- $B > /dev/null 2>&1
timeout -s KILL 150 rocprofv3 --pmc TA_BUSY_avr TA_BUSY_max --kernel-trace --output-format csv -d $R/gpurun_out/pmc_c4_ta -o p -- $B > /dev/null 2>&1
cd $R
python profiles/summarize_rocpd.py $(find gpurun_out/prof_c4 -name "*.db" | head -1) gpurun_out/c4_kernel_stats_grid.txt > /dev/null
head -14 gpurun_out/c4_kernel_stats_grid.txt | cut -c1-60,110-200
for d in fetch write tcc sq ta; do echo "== $d"; python profiles/pmc_summary.py $(find gpurun_out/pmc_c4_$d -name "*counter_collection.csv" | head -1) grid_; done > gpurun_out/c4_pmc.txt
cat gpurun_out/c4_pmc.txt
cat gpurun_out/prof_c4_bench.json
