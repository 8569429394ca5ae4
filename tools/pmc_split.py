#!/usr/bin/env python3
"""
HBM read traffic per kernel launch from the REQUEST-SIZE split of the L2's memory-side read counters
(TCC_EA0_RDREQ_32B / _64B / _128B, rocprofv3 --pmc, csv output): bytes = 32 n32 + 64 n64 + 128 n128 -- no calibration
factor, valid at any row pitch (FETCH_SIZE tallies every non-32-B request at 64 B: exactly half of a 128-B-request stream and
something in between for rows whose pitch leaves 64-B requests).  The sum of the three classes is checked
against TCC_EA0_RDREQ, and the whole scheme against k_calib_read, whose byte count is known.

usage: pmc_split.py <dir with *counter_collection.csv> [<second dir> ...] > table
       (rows: kernel, grid, launches, n32 / n64 / n128 / all requests per launch, bytes per launch)
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

WANT = ("TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum", "TCC_EA0_RDREQ_sum", "TCC_BUBBLE_sum",
        "TCC_HIT_sum", "TCC_MISS_sum", "FETCH_SIZE", "WRITE_SIZE")


def short(name):
    name = name[:name.index("(")] if "(" in name else name
    return name.replace("void ", "").replace("snpm::", "")


def collect(dirs):
    # (kernel, grid, block) -> counter -> list of per-dispatch values
    acc = defaultdict(lambda: defaultdict(dict))
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                c = row.get("Counter_Name")
                if c not in WANT:
                    continue
                key = (short(row["Kernel_Name"]), row.get("Grid_Size", ""), row.get("Workgroup_Size", ""))
                disp = (d, row["Dispatch_Id"])
                acc[key][c][disp] = acc[key][c].get(disp, 0.0) + float(row["Counter_Value"])
    return acc


def main():
    acc = collect(sys.argv[1:])
    out = []
    for key, ctrs in acc.items():
        mean = {c: (sum(v.values()) / len(v)) for c, v in ctrs.items()}
        n = max(len(v) for v in ctrs.values())
        rec = {"kernel": key[0], "grid": key[1], "block": key[2], "launches": n}
        rec.update({c: mean.get(c) for c in WANT if c in mean})
        if all(k in mean for k in WANT[:3]):
            rec["read_bytes_per_launch"] = 32 * mean[WANT[0]] + 64 * mean[WANT[1]] + 128 * mean[WANT[2]]
            if "TCC_EA0_RDREQ_sum" in mean and mean["TCC_EA0_RDREQ_sum"] > 0:
                rec["classes_over_all_requests"] = (mean[WANT[0]] + mean[WANT[1]] + mean[WANT[2]]) / mean["TCC_EA0_RDREQ_sum"]
        if "FETCH_SIZE" in mean:
            rec["FETCH_SIZE_bytes_raw"] = mean["FETCH_SIZE"] * 1024.0
        if "WRITE_SIZE" in mean:
            rec["WRITE_SIZE_bytes"] = mean["WRITE_SIZE"] * 1024.0
        out.append(rec)
    out.sort(key=lambda r: -(r.get("read_bytes_per_launch") or r.get("FETCH_SIZE_bytes_raw") or 0))
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
