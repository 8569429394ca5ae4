#!/usr/bin/env python3
"""
Reduce a rocprofv3 --kernel-trace CSV (or a directory of them) to one row per (kernel, launch shape):

    python tools/kernel_trace_by_shape.py gpurun_out/r03/prof_bench > profiles/r03_bench_kernel_by_shape.csv

The --stats summary pools every launch of a kernel in one row; the bench scores slabs of different sizes
(20.019M + 20.019M + 9.962M rows), so the average of the dominant launch cannot be read from it.  The launch
shape (grid and workgroup size) identifies the slab: k_fast's grid.y is its number of parts, which follows the rows.
Only the library's own kernels (snpm::) are kept unless --all is given.

--phase-marker NAME: launches of the same shape are further split by how many launches of kernel NAME came before
them (bench.py regenerates the resident buffer with k_synth before every slab, and k_fast's grid does not change with
the slab's row count: "phase" then numbers the slabs in the order the bench walks them).
"""
import csv
import glob
import os
import sys
from collections import OrderedDict


def main():
    argv = sys.argv[1:]
    marker = None
    if "--phase-marker" in argv:
        i = argv.index("--phase-marker")
        marker = argv[i + 1]
        del argv[i:i + 2]
    args = [a for a in argv if not a.startswith("--")]
    keep_all = "--all" in argv
    paths = []
    for a in args:
        if os.path.isdir(a):
            paths += sorted(glob.glob(os.path.join(a, "**", "*kernel_trace.csv"), recursive=True))
        else:
            paths.append(a)
    rows = OrderedDict()
    for path in paths:
        with open(path, newline="") as fh:
            recs = sorted(csv.DictReader(fh), key=lambda r: int(r["Start_Timestamp"]))
            phase = 0
            for r in recs:
                name = r["Kernel_Name"]
                if marker and marker in name.split("(")[0]:
                    phase += 1
                if not keep_all and "snpm::" not in name:
                    continue
                short = name.split("(")[0].replace("void ", "").replace("snpm::", "")
                grid = "%sx%sx%s" % (r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])
                wg = "%sx%sx%s" % (r["Workgroup_Size_X"], r["Workgroup_Size_Y"], r["Workgroup_Size_Z"])
                key = (short, grid, wg, r.get("VGPR_Count", ""), r.get("LDS_Block_Size", ""), phase if marker else "")
                dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
                e = rows.setdefault(key, [0, 0.0, 1e30, 0.0])
                e[0] += 1
                e[1] += dur
                e[2] = min(e[2], dur)
                e[3] = max(e[3], dur)
    out = csv.writer(sys.stdout)
    out.writerow(["kernel", "grid", "workgroup", "vgprs", "lds_bytes", "phase", "launches", "total_ms", "avg_ms", "min_ms", "max_ms"])
    for (short, grid, wg, vgpr, lds, ph), (n, tot, lo, hi) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
        out.writerow([short, grid, wg, vgpr, lds, ph, n, "%.4f" % tot, "%.4f" % (tot / n), "%.4f" % lo, "%.4f" % hi])


if __name__ == "__main__":
    main()
