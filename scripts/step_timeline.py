#!/usr/bin/env python3
"""Where one time step goes: from a rocprofv3 --kernel-trace CSV of bench.py, take the last complete step (from one launch of
the anchor kernel to the next), and print per kernel: launches, busy time, and the idle gap in front of its launches.
usage: step_timeline.py <dir> [anchor-kernel-substring] [--seq]"""
import collections
import csv
import glob
import os
import sys


def main():
    d = sys.argv[1]
    anchor = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else "k_dyn"
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("<")[0]))
    rows.sort()
    idx = [i for i, r in enumerate(rows) if anchor in r[2]]
    if len(idx) < 3:
        sys.exit("anchor kernel not found often enough")
    a, b = idx[-3], idx[-2]
    step = rows[a:b]
    span = rows[b][0] - rows[a][0]
    busy = collections.defaultdict(int); gap = collections.defaultdict(int); n = collections.defaultdict(int)
    prev_end = rows[a - 1][1] if a else step[0][0]
    for s, e, k in step:
        busy[k] += e - s; gap[k] += max(0, s - prev_end); n[k] += 1
        if "--seq" in sys.argv:
            print(f"  {k:36s} {(e - s) / 1e3:8.2f} us   gap {max(0, s - prev_end) / 1e3:7.2f} us")
        prev_end = max(prev_end, e)
    print(f"step span {span / 1e3:.1f} us, {len(step)} launches, busy {sum(busy.values()) / 1e3:.1f} us, idle {sum(gap.values()) / 1e3:.1f} us")
    for k in sorted(busy, key=lambda k: -(busy[k] + gap[k])):
        print(f"{k:36s} n={n[k]:4d} busy {busy[k] / 1e3:8.1f} us ({busy[k] / n[k] / 1e3:6.2f} each)  gap-before {gap[k] / 1e3:7.1f} us ({gap[k] / n[k] / 1e3:5.2f} each)")


if __name__ == "__main__":
    main()
