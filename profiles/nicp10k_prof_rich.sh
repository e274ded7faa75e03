# per-kernel times of the 10k-vertex NICP solve with the enriched coarse space (default above 2048 vertices)
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_nicp10k_rich
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_nicp10k_rich -o nd -- python3 $R/benchmarks/nicp_dist_bench.py --nx 100 --ny 100 --lm 3 --maxcg 2000 > $R/gpurun_out/nd10k_rich.json 2> $R/gpurun_out/nd10k_rich.err
cd $R
python profiles/summarize_rocpd.py $(find gpurun_out/prof_nicp10k_rich -name "*.db" | head -1) gpurun_out/nicp10k_rich_kernel_stats.txt | cut -c1-60,110-200 | head -14
cut -c150-330 gpurun_out/nd10k_rich.json
