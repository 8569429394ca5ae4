#!/usr/bin/env python3
"""Per-sample latency at the 1001-Genomes shape: query create / run / likelihood for a 200k-SNP sample."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snpmatch_amd import engine, synth  # noqa: E402

n_snp, n_acc, n_s = 11_000_000, 1135, 200_000
ctx = engine.Context(0)
panel = engine.Panel(ctx, n_snp, n_acc)
panel.fill_synthetic(1001)
rng = np.random.default_rng(5)
rows = np.sort(rng.choice(n_snp, size=n_s, replace=False)).astype(np.int64)
codes, wei = synth.planted_sample(rng, synth.panel_rows(1001, rows, 416, 4)[:, 1], 0.02)
ctx.synchronize()
for rep in range(4):
    t0 = time.perf_counter()
    q = engine.Query(panel, rows, wei)
    t1 = time.perf_counter()
    s, n = q.run(1000, False, engine.MODE_EXACT)
    t2 = time.perf_counter()
    lik, lrt = ctx.likelihood(s, n, truncate=True)
    t3 = time.perf_counter()
    s2, n2 = q.run(1000, False, engine.MODE_EXACT)
    t4 = time.perf_counter()
    q.free()
    print("rep %d: create %.2f ms  first run %.2f ms  likelihood %.2f ms  second run %.2f ms  (top %d)" %
          (rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, int(np.nanargmin(lik))))
ctx.profile(True)
ctx.profile_reset()
q = engine.Query(panel, rows, wei)
q.run(1000, False, engine.MODE_EXACT)
for k in ("lut", "fast", "reduce", "strict", "scan"):
    n, ms = ctx.profile_read(k)
    if n:
        print("  kernel %-7s %.3f ms" % (k, ms / n))
