#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc csv output (one row per dispatch and counter) into per-kernel averages.
   python tools/pmc_summary.py <dir with *counter_collection.csv ...> > profiles/rNN/pmc_xxx.csv"""
import collections
import csv
import glob
import os
import re
import sys


def short(name):
    name = re.sub(r"^void\s+", "", name)
    m = re.match(r"([A-Za-z_0-9]+(<[^(]*>)?)", name)
    return m.group(1) if m else name


def main():
    acc = collections.defaultdict(lambda: [0, 0.0])
    for root in sys.argv[1:]:
        for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
            with open(f, newline="") as fh:
                for row in csv.DictReader(fh):
                    k = (short(row["Kernel_Name"]), row["Counter_Name"])
                    acc[k][0] += 1
                    acc[k][1] += float(row["Counter_Value"])
    w = csv.writer(sys.stdout, lineterminator="\n")
    w.writerow(["kernel", "counter", "launches", "avg_per_launch", "total"])
    for (k, c), (n, tot) in sorted(acc.items()):
        w.writerow([k, c, n, "%.3f" % (tot / n), "%.3f" % tot])


if __name__ == "__main__":
    main()
