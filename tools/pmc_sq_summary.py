#!/usr/bin/env python3
"""Average the SQ counters of the dispatches of one kernel (argv[2], default k_fast) in a rocprofv3 --pmc csv directory."""
import csv
import glob
import os
import sys
from collections import defaultdict

KERNEL = sys.argv[2] if len(sys.argv) > 2 else "k_fast"
acc = defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if KERNEL in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
# rocprofv3 emits one row per (dispatch, counter, dimension instance): sum per dispatch first
per = {}
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    d = defaultdict(lambda: defaultdict(float))
    for row in csv.DictReader(open(f)):
        if KERNEL in row["Kernel_Name"]:
            d[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
    for k, v in d.items():
        per[k] = sum(v.values()) / len(v)
for k in sorted(per):
    print("%-28s %.4g" % (k, per[k]))
wc = per.get("SQ_WAVE_CYCLES")
if wc:
    for k in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS"):
        if k in per:
            print("%-28s / SQ_WAVE_CYCLES = %.3f" % (k, per[k] / wc))
