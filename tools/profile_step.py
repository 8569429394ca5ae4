#!/usr/bin/env python3
"""Per-kernel time of one scoring step (library HIP events): tools/profile_step.py N_ACC N_SNP [mode]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from snpmatch_amd import engine  # noqa: E402

n_acc, n_snp = int(sys.argv[1]), int(sys.argv[2])
mode = {"exact": engine.MODE_EXACT, "strict": engine.MODE_STRICT, "fast": engine.MODE_FAST}[sys.argv[3] if len(sys.argv) > 3 else "exact"]
ctx = engine.Context(0)
panel = engine.Panel(ctx, n_snp, n_acc)
panel.fill_synthetic(bench.SEED)
wei = bench.make_sample(n_snp, bench.SEED, bench.PLANTED)
q = engine.Query(panel, None, wei)
print("error bound", q.error_bound(1000))
for _ in range(2):
    q.run_device(1000, False, mode)
ctx.synchronize()
ctx.profile(True)
ctx.profile_reset()
t0 = time.perf_counter()
K = 5
nre = 0
for _ in range(K):
    q.run_device(1000, False, mode)
    nre += q.last_reeval()
ctx.synchronize()
dt = (time.perf_counter() - t0) / K * 1e3
print("wall ms/step %.3f  re-evaluated columns/step %.1f" % (dt, nre / K))
for k in ("fast", "reduce", "strict", "scan", "lut"):
    n, ms = ctx.profile_read(k)
    if n:
        print("  %-8s %3d launches  %.3f ms each" % (k, n, ms / n))
