#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel-trace stats + PMC passes) into a short text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]


def short(name):
    return name.split("(")[0][-90:]


for f in sorted(glob.glob(os.path.join(root, "kt", "**", "*kernel_stats.csv"), recursive=True)):
    print("== kernel stats:", os.path.relpath(f, root))
    for row in list(csv.DictReader(open(f)))[:8]:
        print("  ", short(row.get("Name", "")), "| calls", row.get("Calls"), "| avg ns", row.get("AverageNs"), "| total ns", row.get("TotalDurationNs"),
              "| %", row.get("Percentage"))
for f in sorted(glob.glob(os.path.join(root, "kt", "**", "*kernel_trace.csv"), recursive=True)):
    rows = list(csv.DictReader(open(f)))
    by = defaultdict(list)
    for r in rows:
        by[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r))
    print("== kernel trace:", os.path.relpath(f, root), len(rows), "dispatches")
    for k, v in sorted(by.items(), key=lambda kv: -sum(e - s for s, e, _ in kv[1]))[:4]:
        d = sorted(e - s for s, e, _ in v)
        v.sort()
        gaps = sorted(v[i + 1][0] - v[i][1] for i in range(len(v) - 1))
        r0 = v[0][2]
        print(f"   {short(k)}\n      n={len(d)} dur ns min/med/mean/max = {d[0]}/{d[len(d)//2]}/{sum(d)//len(d)}/{d[-1]}  gap ns med = {gaps[len(gaps)//2] if gaps else None}"
              f"  grid={r0.get('Grid_Size_X', r0.get('Grid_Size'))} wg={r0.get('Workgroup_Size_X', r0.get('Workgroup_Size'))} vgpr={r0.get('VGPR_Count')} sgpr={r0.get('SGPR_Count')} lds={r0.get('LDS_Block_Size')}")
for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        rows = list(csv.DictReader(open(f)))
        acc = defaultdict(lambda: defaultdict(list))
        for r in rows:
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        print("== pmc:", os.path.relpath(f, root), len(rows), "rows")
        for k, cs in sorted(acc.items(), key=lambda kv: -len(next(iter(kv[1].values()))))[:2]:
            print("   ", short(k))
            for c, vals in cs.items():
                vals.sort()
                print(f"       {c}: n={len(vals)} median={vals[len(vals)//2]:.4g} mean={sum(vals)/len(vals):.4g}")
