# Round-3 profile set for the default bench (grid search).  Run on the GPU box from the repo root:
#   bash profiles/run_r3_profiles.sh
# Each rocprofv3 pass is a run of its own (kernel trace only, or one counter group within the per-block limits).
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --steps 10 --warmup 2 --no-cpu --no-nicp --no-check --repeats 0"
O=$R/gpurun_out/r3prof
rm -rf $O && mkdir -p $O
timeout -s KILL 120 rocprofv3 --kernel-trace --stats -d $O/trace -o g -- python3 $B > $O/bench.json 2>/dev/null
timeout -s KILL 90 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o p -- python3 $B > /dev/null 2>&1
timeout -s KILL 90 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o p -- python3 $B > /dev/null 2>&1
timeout -s KILL 90 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --kernel-trace --output-format csv -d $O/tcc -o p -- python3 $B > /dev/null 2>&1
timeout -s KILL 90 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq -o p -- python3 $B > /dev/null 2>&1
timeout -s KILL 90 rocprofv3 --pmc TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/ta -o p -- python3 $B > /dev/null 2>&1
cd $R
python3 profiles/summarize_rocpd.py $(find $O/trace -name "*.db" | head -1) gpurun_out/r3_kernel_stats_grid.txt > /dev/null
for d in fetch write tcc sq ta; do echo "== $d"; python3 profiles/pmc_summary.py $(find $O/$d -name "*counter_collection.csv" | head -1) grid_; done > gpurun_out/r3_pmc_grid_pool_raw.txt
python3 profiles/r3_traffic.py $O gpurun_out/r3_kernel_stats_grid.txt > gpurun_out/r3_traffic.json
cat gpurun_out/r3_traffic.json
