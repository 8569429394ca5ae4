#!/usr/bin/env python3
"""Workload for the PMC passes (run under `rocprofv3 --pmc ...`): the bench panel shard, one calibration
read of known size (k_calib_read) and a few launches of the scoring kernel (k_fast)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from snpmatch_amd import engine  # noqa: E402

n_acc = int(os.environ.get("PMC_N_ACC", "10000"))
n_snp = int(os.environ.get("PMC_N_SNP", "6250000"))
ctx = engine.Context(0)
packed = os.environ.get("PMC_PACKED", "0") == "1"              # 2-bit packed panel (k_fast_packed_q4; PMC_HARD=1: k_fast_bits)
panel = engine.Panel(ctx, n_snp, n_acc, packed=packed)
panel.fill_synthetic(bench.SEED)
import torch  # noqa: E402
wei = torch.empty((n_snp, 3), dtype=torch.float64, device="cuda:0")
ctx.sample_synthetic(bench.SEED, 0, n_snp, bench.PLANTED, wei.data_ptr(), frac_pl=0.0 if os.environ.get("PMC_HARD", "0") == "1" else 0.8)
q = engine.Query.from_device(panel, None, wei.data_ptr(), n_snp)
for _ in range(2):
    print("calib bytes", panel.stream_read())
for _ in range(3):
    q.run(1000, False, engine.MODE_FAST)
print("alg bytes per fast-pass launch", n_snp * ((n_acc / 4 if packed else n_acc) + 24))
