set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_c5c -o g -- python3 $R/benchmarks/nicp_phases.py --nx 448 --ny 448 > $R/gpurun_out/prof_c5c.json 2>/dev/null
cd $R
python profiles/summarize_rocpd.py $(find gpurun_out/prof_c5c -name "*.db" | head -1) gpurun_out/c5c_kernel_stats.txt > /dev/null
head -24 gpurun_out/c5c_kernel_stats.txt | cut -c1-60,100-200
