#!/usr/bin/env python3
"""Workload for SQ counter passes: a few k_fast launches on the bench shard (PMC_PACKED=1 for the 2-bit panel)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from snpmatch_amd import engine  # noqa: E402

n_acc = int(os.environ.get("PMC_N_ACC", "10000"))
n_snp = int(os.environ.get("PMC_N_SNP", "6250000"))
ctx = engine.Context(0)
panel = engine.Panel(ctx, n_snp, n_acc, packed=os.environ.get("PMC_PACKED", "0") == "1")
panel.fill_synthetic(bench.SEED)
wei = bench.make_sample(n_snp, bench.SEED, bench.PLANTED)
q = engine.Query(panel, None, wei)
for _ in range(3):
    q.run(1000, False, engine.MODE_FAST)
