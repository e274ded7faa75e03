#!/usr/bin/env python3
"""Turn a rocprofv3 (ROCm 7.2) rocpd sqlite result into the per-kernel stats table that
`rocprofv3 --kernel-trace --stats` describes: calls, total / average / min / max duration."""
import sqlite3
import sys


def main(db, out=None):
    c = sqlite3.connect(db)
    rows = c.execute(
        "select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
        "from kernels group by name order by sum(end-start) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    lines = ["%-110s %8s %14s %14s %14s %14s %7s" % ("kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct")]
    for name, calls, total, avg, mn, mx in rows:
        lines.append("%-110s %8d %14d %14.0f %14d %14d %6.2f%%" % (name[:110], calls, total, avg, mn, mx, 100.0 * total / tot))
    # bench.py's roofline leg sends two kinds of search through the same kernel symbol (with the previous answer as the
    # hint -- the dominant one -- and without): report the two populations of its durations separately
    for name, calls, total, avg, mn, mx in rows:
        if "grid_pool_kernel" in name and calls >= 8 and mx > 1.3 * mn:
            cut = 0.5 * (mn + mx)
            d = [r[0] for r in c.execute("select end-start from kernels where name = ?", (name,)).fetchall()]
            for label, part in (("<= %.0f us (hinted)" % (cut * 1e-3), [x for x in d if x <= cut]),
                                ("> %.0f us (no hint)" % (cut * 1e-3), [x for x in d if x > cut])):
                if part:
                    lines.append("#   %s  %s: calls %d  avg_ns %.0f  min_ns %d  max_ns %d" % (name[:40], label, len(part),
                                                                                             sum(part) / len(part), min(part), max(part)))
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    print(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
