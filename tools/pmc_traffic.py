#!/usr/bin/env python3
"""
Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; csv output) of tools/pmc_run.py into HBM
bytes per k_fast launch, following MI355X_MICROARCH.md (HBM section): counters are in KiB, and
FETCH_SIZE is calibrated on the known byte count of k_calib_read, which has the access shape of k_fast.
usage: pmc_traffic.py <fetch_dir> <write_dir> <n_gpus_key> <out.json>
"""
import csv
import glob
import json
import os
import sys


def per_kernel(d, counter):
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            name = row["Kernel_Name"]
            key = "k_fast" if "k_fast" in name else ("k_calib_read" if "k_calib_read" in name else None)
            if key:
                out.setdefault(key, {}).setdefault(row["Dispatch_Id"], 0.0)
                out[key][row["Dispatch_Id"]] += float(row["Counter_Value"])
    return {k: sorted(v.values()) for k, v in out.items()}


fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
n_acc = int(os.environ.get("PMC_N_ACC", "10000"))
n_snp = int(os.environ.get("PMC_N_SNP", "6250000"))
packed = os.environ.get("PMC_PACKED", "0") == "1"
align = int(os.environ.get("SNPM_PITCH_ALIGN", "256"))          # bytes a panel row is padded to
pitch = ((n_acc + 3) // 4 + align - 1) // align * align if packed else (n_acc + align - 1) // align * align
calib_bytes = n_snp * pitch
calib_fetch = sum(fetch["k_calib_read"]) / len(fetch["k_calib_read"]) * 1024.0
factor = calib_bytes / calib_fetch
kf = sum(fetch["k_fast"]) / len(fetch["k_fast"]) * 1024.0
kw = sum(write["k_fast"]) / len(write["k_fast"]) * 1024.0
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
try:
    from snpmatch_amd import _lib
    build_id = _lib.build_id()
except Exception:          # noqa: BLE001
    build_id = None
res = {
    "build_id": build_id,
    "n_acc": n_acc, "n_snp": n_snp, "panel_format": "packed2" if packed else "int8",
    "sample": "hard calls" if os.environ.get("PMC_HARD", "0") == "1" else "80% PL weights",
    "calibration": {"kernel": "k_calib_read", "known_bytes": calib_bytes, "FETCH_SIZE_bytes_raw": calib_fetch,
                    "correction_factor": factor},
    "k_fast": {"FETCH_SIZE_bytes_raw": kf, "fetch_bytes_corrected": kf * factor, "WRITE_SIZE_bytes": kw,
               "algorithmic_bytes": n_snp * ((n_acc / 4.0 if packed else n_acc) + 24.0)},
    "hbm_bytes_per_launch": kf * factor + kw,
}
out = {}
if os.path.exists(sys.argv[4]):
    out = json.load(open(sys.argv[4]))
out[sys.argv[3]] = res
json.dump(out, open(sys.argv[4], "w"), indent=1)
print(json.dumps(res, indent=1))
