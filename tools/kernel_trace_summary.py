#!/usr/bin/env python3
"""rocprofv3 kernel trace (csv) -> per (kernel, grid, block) launches and average / min / max duration in microseconds"""
import csv
import sys
from collections import defaultdict

rows = defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"]
    name = name[:name.index("(")] if "(" in name else name
    key = (name[:110], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")))
    rows[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("%-112s %10s %8s %6s %8s %10s %10s %10s" % ("kernel", "grid.x", "grid.y", "block", "launches", "avg_us", "min_us", "max_us"))
for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    print("%-112s %10s %8s %6s %8d %10.2f %10.2f %10.2f" % (k[0], k[1], k[2], k[3], len(v), sum(v) / len(v), min(v), max(v)))
