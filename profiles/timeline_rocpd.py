#!/usr/bin/env python3
"""Print the kernels of the LAST burst (gap > 2 ms before it) of a rocprofv3 rocpd result in start order:
offset from the burst's first kernel, duration, queue/stream, name.  For reading the critical path of a call's set-up."""
import sqlite3
import sys


def main(db):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    extra = [x for x in ("stream_id", "queue_id") if x in cols]
    rows = c.execute("select start, end, name%s from kernels order by start" % "".join(", " + e for e in extra)).fetchall()
    first = 0
    for i in range(1, len(rows)):
        if rows[i][0] - rows[i - 1][1] > 2_000_000:
            first = i
    t0 = rows[first][0]
    print("columns:", cols)
    for r in rows[first:]:
        print("%9.1f us  %8.1f us  %s  %s" % ((r[0] - t0) / 1e3, (r[1] - r[0]) / 1e3, " ".join(str(x) for x in r[3:]), r[2][:100]))
    print("burst: %.1f us" % ((rows[-1][1] - t0) / 1e3))


if __name__ == "__main__":
    main(sys.argv[1])
