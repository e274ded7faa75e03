# per-kernel times of the 200k-vertex (C5-size) NICP solve on one GPU
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_nicp200k -o nd -- python3 $R/benchmarks/nicp_dist_bench.py > $R/gpurun_out/nd_prof.json 2> $R/gpurun_out/nd_prof.err
cd $R
python profiles/summarize_rocpd.py $(find gpurun_out/prof_nicp200k -name "*.db" | head -1) gpurun_out/nicp200k_kernel_stats.txt | cut -c1-60,110-200 | head -14
