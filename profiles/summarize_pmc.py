#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per launch and kernel.

usage: summarize_pmc.py OUT.csv DIR [DIR ...]   (each DIR = one `rocprofv3 --pmc ... -d DIR` pass)
"""
import collections
import csv
import glob
import re
import sys


def short(name: str) -> str:
    m = re.search(r"::((?:k\d?w?_|k3_|t_)\w+)", name)
    if m:
        return m.group(1)
    if "segmented_radix_sort" in name:
        return "rocprim_segmented_radix_sort"
    return name[:40]


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    acc = collections.defaultdict(float)
    seen = collections.defaultdict(set)
    for d in dirs:
        for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                key = (short(r["Kernel_Name"]), r["Counter_Name"])
                acc[key] += float(r["Counter_Value"])
                seen[key].add(r["Dispatch_Id"])
    with open(out, "w") as f:
        f.write("kernel,counter,launches,value_per_launch\n")
        for (k, c) in sorted(acc):
            f.write("%s,%s,%d,%.1f\n" % (k, c, len(seen[(k, c)]), acc[(k, c)] / len(seen[(k, c)])))


if __name__ == "__main__":
    main()
