set -e
b=${1:-7}
ICP_GRID_BITS=$b ICP_GRID_DEBUG=gpurun_out/dbg$b.bin timeout -k 10 120 python bench.py --steps 5 --warmup 1 --no-cpu > gpurun_out/sw$b.json 2>/dev/null
echo "== bits $b"
python profiles/grid_wave_stats.py gpurun_out/dbg$b.bin
rm -f gpurun_out/dbg$b.bin
