#!/usr/bin/env python3
"""Workload of the PMC passes on dense scans: panel PMC_N_ACC x PMC_N_SNP (PMC_PACKED, PMC_HARD as tools/pmc_run.py), one
k_calib_read over the whole panel allocation (known byte count: rows x pitch) and three fast-pass launches."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snpmatch_amd import engine  # noqa: E402
import torch  # noqa: E402

n_acc = int(os.environ.get("PMC_N_ACC", "1135"))
n_snp = int(os.environ.get("PMC_N_SNP", "40000000"))
packed = os.environ.get("PMC_PACKED", "0") == "1"
ctx = engine.Context(0)
panel = engine.Panel(ctx, n_snp, n_acc, packed=packed)
panel.fill_synthetic(10050)
wei = torch.empty((n_snp, 3), dtype=torch.float64, device="cuda:0")
ctx.sample_synthetic(10050, 0, n_snp, 417, wei.data_ptr(), frac_pl=0.0 if os.environ.get("PMC_HARD", "0") == "1" else 0.8)
q = engine.Query.from_device(panel, None, wei.data_ptr(), n_snp)
for _ in range(2):
    print("calib bytes", panel.stream_read(), "pitch", panel.pitch)
for _ in range(3):
    q.run(1000, False, engine.MODE_FAST)
print("kernel", q.last_kernel(), "alg bytes per launch", n_snp * ((n_acc / 4 if packed else n_acc) + 24))
