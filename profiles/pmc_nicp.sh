# counters for the NICP CG kernels at 200k vertices (one group per run)
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/benchmarks/nicp_dist_bench.py --lm 1 --maxcg 60"
timeout -s KILL 120 rocprofv3 --pmc TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --kernel-trace --output-format csv -d $R/gpurun_out/pmc_nicp1 -o p -- $B > /dev/null 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_nicp2 -o p -- $B > /dev/null 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_nicp3 -o p -- $B > /dev/null 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $R/gpurun_out/pmc_nicp4 -o p -- $B > /dev/null 2>&1
cd $R
for d in 1 2 3 4; do for k in jtu_kernel jp_kernel cg_update_prec cg_dir_prec; do python profiles/pmc_summary.py $(find gpurun_out/pmc_nicp$d -name "*counter_collection.csv" | head -1) $k; done; done > gpurun_out/nicp_pmc.txt
cat gpurun_out/nicp_pmc.txt
