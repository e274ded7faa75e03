# Round 3 (same passes as round 2): kernel times + memory-side counters of the NICP CG kernels (one pass per counter group, as gpurun requires).
# usage (on the GPU box, from the repo root): bash profiles/r3_pmc_nicp.sh TAG [NX NY]   -> gpurun_out/r3_pmc_nicp_TAG.txt
set -e
TAG=${1:-run}
NX=${2:-448}
NY=${3:-448}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="$R/benchmarks/nicp_dist_bench.py --nx $NX --ny $NY --lm 1 --maxcg 100"
O=$R/gpurun_out/r3pmc_$TAG
rm -rf $O
timeout -s KILL 150 rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python3 $B > $O.bench.json 2> $O.trace.err
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum --kernel-trace --output-format csv -d $O/p1 -o p -- python3 $B > /dev/null 2>&1
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $O/p2 -o p -- python3 $B > /dev/null 2>&1
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/p3 -o p -- python3 $B > /dev/null 2>&1
cd $R
python3 profiles/r2_pmc_nicp_summary.py $O $NX $NY > gpurun_out/r3_pmc_nicp_$TAG.txt
cat gpurun_out/r3_pmc_nicp_$TAG.txt
