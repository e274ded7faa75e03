# per-kernel times of the 10k-vertex (C3-size) NICP solve
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_nicp10k
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_nicp10k -o nd -- python3 $R/benchmarks/nicp_dist_bench.py --nx 100 --ny 100 --lm 3 --maxcg 2000 > $R/gpurun_out/nd10k.json 2> $R/gpurun_out/nd10k.err
cd $R
python profiles/summarize_rocpd.py $(find gpurun_out/prof_nicp10k -name "*.db" | head -1) gpurun_out/nicp10k_kernel_stats.txt | cut -c1-60,110-200 | head -12
cut -c150-330 gpurun_out/nd10k.json
