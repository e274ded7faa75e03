#!/usr/bin/env python3
"""profiles/r3_traffic.json from the passes of run_r3_profiles.sh: memory-side bytes per grid search (FETCH_SIZE x 2 on
gfx950 + WRITE_SIZE, KiB) and what binds the per-lane kernel -- vector-instruction issue (VALU instructions x 4 cycles
over the 1024 SIMDs against the kernel's cycles), the L1 address path (line accesses per compute unit against one line
per cycle; TA_BUSY as the hardware's own figure), and the share of wave cycles spent waiting.  bench.py reads this
file for roofline.traffic / valu_busy / ta_busy / wait_frac."""
import collections
import csv
import glob
import json
import sys

CLOCK_GHZ = 2.4


def counters(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(x) / len(x) for c, x in v.items()} for k, v in acc.items()}


def main(root, stats):
    c = {}
    for d in ("fetch", "write", "tcc", "sq", "ta"):
        for k, v in counters(root + "/" + d).items():
            c.setdefault(k, {}).update(v)
    dur = {}
    for ln in open(stats).read().splitlines()[1:]:
        name, rest = ln[:110], ln[110:].split()
        dur[name.strip()] = float(rest[2]) * 1e-3          # average, microseconds
    pick = lambda frag: [k for k in c if frag in k]
    loop = [k for k in pick("grid_pool_kernel") if "<3, true>" in k][0]
    hinted = [k for k in pick("grid_pool_kernel") if "<3, false>" in k][0]     # hinted and unhinted searches of bench.py's roofline leg, averaged
    defer = [k for k in pick("grid_defer_kernel") if "false, false>" in k][0]
    us = lambda k: [v for n, v in dur.items() if k[:100].startswith(n[:100]) or n[:100].startswith(k[:100])][0]
    bytes_of = lambda k: 2.0 * c[k]["FETCH_SIZE"] * 1024 + c[k]["WRITE_SIZE"] * 1024
    cyc = us(loop) * 1e3 * CLOCK_GHZ
    out = {
        "kernel": "grid search = grid_pool_kernel + grid_defer_kernel (bench.py's roofline leg; the counters average over its hinted and "
                  "unhinted searches, which share the kernel symbol; in_loop_with_hints_fused is the hinted search inside the loop alone)",
        "workload": "2^20 x 2^20 (C2)",
        "fetch_size_kb": {"grid_pool_kernel": c[hinted]["FETCH_SIZE"], "grid_defer_kernel": c[defer]["FETCH_SIZE"]},
        "fetch_correction": 2.0,
        "write_size_kb": {"grid_pool_kernel": c[hinted]["WRITE_SIZE"], "grid_defer_kernel": c[defer]["WRITE_SIZE"]},
        "traffic_bytes": bytes_of(hinted) + bytes_of(defer),
        "in_loop_with_hints_fused": {"traffic_bytes": bytes_of(loop) + bytes_of(defer), "kernel_us": us(loop), "defer_us": us(defer)},
        "valu_busy": c[loop]["SQ_INSTS_VALU"] * 4.0 / 1024.0 / cyc,
        "ta_busy": c[loop]["TCP_TOTAL_CACHE_ACCESSES_sum"] / 256.0 / cyc,
        "ta_busy_counter": (c[loop]["TA_BUSY_avr"] / cyc) if "TA_BUSY_avr" in c[loop] else None,   # TA_BUSY_avr counts busy cycles
        "wait_frac": c[loop]["SQ_WAIT_ANY"] / c[loop]["SQ_WAVE_CYCLES"],
        "l2_hit": c[loop]["TCC_HIT_sum"] / (c[loop]["TCC_HIT_sum"] + c[loop]["TCC_MISS_sum"]),
        "busy_note": "grid_pool_kernel<3, true> inside the ICP loop: valu_busy = SQ_INSTS_VALU x 4 cycles / 1024 SIMDs / kernel cycles at 2.4 GHz; "
                     "ta_busy = L1 line accesses per compute unit / kernel cycles (one line per cycle); wait_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES",
        "source": "profiles/r3_pmc_grid_pool.txt (rocprofv3 --pmc, one counter group per run: profiles/run_r3_profiles.sh; gfx950 x2 correction for FETCH_SIZE)",
    }
    print(json.dumps(out))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
