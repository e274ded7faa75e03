# TA / TCP busy and stall counters for the grid query kernel (one pass per counter group)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export ICP_GRID_CAP=${ICP_GRID_CAP:-1}
timeout -s KILL 90 rocprofv3 --pmc TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE TCP_GATE_EN1_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum --kernel-trace --output-format csv -d $R/gpurun_out/pmc_ta1 -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu > /dev/null 2>&1
timeout -s KILL 90 rocprofv3 --pmc TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $R/gpurun_out/pmc_ta2 -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu > /dev/null 2>&1
cd $R
for d in pmc_ta1 pmc_ta2; do python profiles/pmc_summary.py $(find gpurun_out/$d -name "*counter_collection.csv" | head -1) grid_; done
