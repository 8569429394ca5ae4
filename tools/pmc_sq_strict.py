#!/usr/bin/env python3
"""Workload for SQ counter passes on the reference-order kernel: a few strict passes on a 10k x 2M panel."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401
import bench  # noqa: E402
from snpmatch_amd import engine  # noqa: E402

n_acc, n_snp = 10000, 2_000_000
ctx = engine.Context(0)
panel = engine.Panel(ctx, n_snp, n_acc, packed=os.environ.get("PMC_PACKED", "0") == "1")
panel.fill_synthetic(bench.SEED)
wei = torch.empty((n_snp, 3), dtype=torch.float64, device="cuda:0")
ctx.sample_synthetic(bench.SEED, 0, n_snp, bench.PLANTED, wei.data_ptr())
q = engine.Query.from_device(panel, None, wei.data_ptr(), n_snp)
for _ in range(3):
    q.run(1000, False, engine.MODE_STRICT)
