#!/usr/bin/env python3
"""Average per-dispatch PMC values per kernel from a rocprofv3 --pmc ... --output-format csv run."""
import collections
import csv
import sys


def main(path, pattern=""):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        if pattern in k:
            print(k[:90])
            for c, x in sorted(v.items()):
                print("    %-34s %16.1f  (n=%d)" % (c, sum(x) / len(x), len(x)))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
