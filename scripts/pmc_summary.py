#!/usr/bin/env python3
"""Mean per-dispatch PMC counters per kernel from a rocprofv3 --pmc run.

usage: pmc_summary.py <dir with *_counter_collection.csv> [out.json] [name-filter]
"""
import csv, glob, json, os, sys
from collections import defaultdict

def main():
    d = sys.argv[1]
    out = sys.argv[2] if len(sys.argv) > 2 else None
    flt = sys.argv[3] if len(sys.argv) > 3 else ""
    acc = defaultdict(lambda: defaultdict(list))
    meta = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = row["Kernel_Name"].split("(")[0]
                if flt and flt not in k:
                    continue
                acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
                meta[k] = {"vgpr": int(row["VGPR_Count"]), "agpr": int(row["Accum_VGPR_Count"]), "sgpr": int(row["SGPR_Count"]),
                           "scratch": int(row["Scratch_Size"]), "lds": int(row["LDS_Block_Size"]), "grid": int(row["Grid_Size"]),
                           "wg": int(row["Workgroup_Size"])}
    res = {}
    for k, cs in acc.items():
        res[k] = {c: sum(v) / len(v) for c, v in sorted(cs.items())}
        res[k]["_launch"] = meta[k]
        res[k]["_dispatches"] = max(len(v) for v in cs.values())
    txt = json.dumps(res, indent=1)
    if out:
        open(out, "w").write(txt + "\n")
    print(txt)

if __name__ == "__main__":
    main()
