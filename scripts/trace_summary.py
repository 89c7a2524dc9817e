#!/usr/bin/env python3
"""Mean / min duration per (kernel, grid size) from a rocprofv3 --kernel-trace CSV.  usage: trace_summary.py <dir> [name-filter]"""
import collections
import csv
import glob
import os
import sys


def main():
    d = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if flt and flt not in k:
                continue
            acc[(k, int(r.get("Grid_Size_X") or r.get("Grid_Size") or 0) // max(1, int(r.get("Workgroup_Size_X") or 1)))].append(
                int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for (k, wgs), v in sorted(acc.items()):
        print(f"{k:40s} wgs={wgs:6d} n={len(v):5d} mean {sum(v) / len(v) / 1e3:8.2f} us   min {min(v) / 1e3:8.2f} us")


if __name__ == "__main__":
    main()
