#!/usr/bin/env python3
"""Summary of profiles/r2_pmc_nicp.sh: per NICP CG kernel the average duration (rocprofv3 --kernel-trace), the
memory-side bytes per launch (FETCH_SIZE / WRITE_SIZE, separate --pmc passes), L2 hit rate and wait share.

Unit handling (guides/MI355X_MICROARCH.md, HBM section): FETCH_SIZE and WRITE_SIZE are reported in KILOBYTES by
rocprofv3 and derive from the L2's memory-side request counters (Infinity-Cache hits included).  On gfx950
FETCH_SIZE tallies 128-byte requests at 64 bytes for 16-B-per-lane loads (x2 correction); these kernels load
8 bytes per lane, a width the guide leaves uncalibrated, so the factor is CHECKED here on kernels of the same run
whose byte count is known and whose loads have exactly this shape and nothing else: cg_dir_prec (reads P, zb, p
and v once: 312 B per vertex) and cg_update_prec (x, r, p, Ap, P, the 78 packed block entries, v: 1128 B per
vertex) -- their working set is far beyond the 32 MB of L2, so every byte comes from the memory side.  The
ratios are printed (2.0 +- 0.05 in every run so far, the guide's factor) and the table applies x2 to FETCH_SIZE;
WRITE_SIZE is taken as it is (the guide: exact for streaming stores; axpby_kernel here: 0.985)."""
import collections
import csv
import glob
import sqlite3
import sys


def pmc(dirname):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(dirname + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


def main(root, nx, ny):
    n = nx * ny
    ne = (nx - 1) * ny + nx * (ny - 1) + (nx - 1) * (ny - 1)
    npar = 12 * n + 6
    db = glob.glob(root + "/trace/**/*.db", recursive=True)[0]
    rows = sqlite3.connect(db).execute(
        "select name, count(*), avg(end-start), min(end-start), max(end-start) from kernels group by name").fetchall()
    dur = {r[0]: r for r in rows}
    p1, p2, p3 = pmc(root + "/p1"), pmc(root + "/p2"), pmc(root + "/p3")

    def avg(tab, kern, ctr):
        for k, v in tab.items():
            if kern in k and ctr in v:
                # steady state: drop the first launches of a kernel (cold caches, warm-up solve)
                x = v[ctr][len(v[ctr]) // 4:]
                return sum(x) / len(x)
        return float("nan")

    # check of the FETCH_SIZE factor for 8-byte-per-lane coalesced loads on the two pure streaming kernels
    f_dir = avg(p1, "cg_dir_prec", "FETCH_SIZE") * 1024.0
    f_upd = avg(p1, "cg_update_prec", "FETCH_SIZE") * 1024.0
    w_ax = avg(p2, "icp::axpby_kernel", "WRITE_SIZE") * 1024.0
    print("mesh %d x %d: %d vertices, %d edges, %d parameters" % (nx, ny, n, ne, npar))
    print("FETCH_SIZE check (known streamed read bytes / reported): cg_dir_prec %.3f, cg_update_prec %.3f  -> x2 applied below; "
          "WRITE_SIZE check (axpby_kernel known / reported): %.3f -> taken as reported"
          % (n * 312.0 / f_dir if f_dir else float("nan"), n * 1128.0 / f_upd if f_upd else float("nan"),
             8.0 * npar / w_ax if w_ax else float("nan")))
    cal = 2.0
    B = 8.0
    # minimal bytes of each kernel AS IT IS WRITTEN (every array it must touch once), per launch
    alg = {
        "jp_kernel": n * (96 + 24 + 16 + 32 + 24 + 72) + ne * 8 + B * (9 * n + 6 * ne),
        "jtu_kernel<3, false, false>": B * (9 * n + 6 * ne) + n * (24 + 72 + 96 + 96 + 24 + 8 + 48) + ne * 2 * 8 + n * 96,
        "cg_update_prec": n * (5 * 96 + 78 * 8 + 24 + 3 * 96),
        "cg_dir_prec": n * (3 * 96 + 24 + 96),
        "fitrow_kernel": n * (96 + 24 + 16 + 32 + 24 + 24),
        "normal_kernel": n * (96 * 3 + 24 + 72 + 24 + 8 + 48 + 96) + ne * 2 * 8,
    }
    print("\n%-34s %6s %9s %11s %11s %11s %7s %7s %7s" % ("kernel", "calls", "avg us", "fetch MB", "write MB", "minimal MB", "ratio", "L2 hit", "wait"))
    tot_t = tot_b = 0.0
    for kern, a in alg.items():
        full = [k for k in dur if kern in k and dur[k][1] >= 20]      # kernels of the CG iteration only
        if not full:
            continue
        name = full[0]
        t = dur[name][2] / 1e3
        fe = avg(p1, kern, "FETCH_SIZE") * 1024.0 * cal
        wr = avg(p2, kern, "WRITE_SIZE") * 1024.0
        hit, miss = avg(p1, kern, "TCC_HIT_sum"), avg(p2, kern, "TCC_MISS_sum")
        wait = avg(p3, kern, "SQ_WAIT_ANY") / max(avg(p3, kern, "SQ_WAVE_CYCLES"), 1.0)
        print("%-34s %6d %9.1f %11.1f %11.1f %11.1f %7.2f %7.2f %7.2f" % (kern[:34], dur[name][1], t, fe / 1e6, wr / 1e6, a / 1e6,
                                                                       (fe + wr) / a, hit / max(hit + miss, 1.0), wait))
        tot_t += t
        tot_b += fe + wr
    sums = [k for k in dur if "cg_sum_" in k]
    ts = sum(dur[k][2] / 1e3 for k in sums)
    print("single-block sum + logic kernels: %.1f us per iteration (%s)" % (ts, ", ".join(k.split("(")[0] for k in sums)))
    it_t = tot_t + ts
    print("\nper CG iteration: %.1f us of kernel time; memory-side traffic %.1f MB = %.2f x the SURVEY 8(d) algorithmic %.1f MB "
          "(0.86 kB/vertex); algorithmic rate %.2f TB/s = %.3f of the 8 TB/s HBM peak"
          % (it_t, tot_b / 1e6, tot_b / (860.0 * n), 860.0 * n / 1e6, 860.0 * n / it_t / 1e6, 860.0 * n / it_t / 1e6 / 8.0))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]))
