# C3-size NICP: CG iterations and time per coarse space (ICP_NICP_COARSE = 1 | 4 | 10 fixed; unset = adaptive default),
# first two outer iterations
for m in 1 4 10 default; do
  echo "== coarse $m"
  if [ $m = default ]; then unset ICP_NICP_COARSE; else export ICP_NICP_COARSE=$m; fi
  timeout -k 10 200 python -u benchmarks/nicp_bench.py --fixed --cgtol 1e-6 --run 2 2>&1 | tail -1 | cut -c100-330 || exit 1
done
