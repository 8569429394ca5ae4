#!/usr/bin/env python3
"""Read-only streaming ceiling: time k_calib_read (4 B/lane, non-temporal, no compute) over the bench panel."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snpmatch_amd import engine  # noqa: E402

ctx = engine.Context(0)
panel = engine.Panel(ctx, 6_250_000, 10000)
panel.fill_synthetic(1)
ctx.synchronize()
for _ in range(2):
    panel.stream_read()
ts = []
for _ in range(8):
    t0 = time.perf_counter()
    nbytes = panel.stream_read()          # synchronises
    ts.append(time.perf_counter() - t0)
best, med = min(ts), sorted(ts)[len(ts) // 2]
print("k_calib_read: %.1f GB  best %.3f ms = %.0f GB/s   median %.3f ms = %.0f GB/s" %
      (nbytes / 1e9, best * 1e3, nbytes / best / 1e9, med * 1e3, nbytes / med / 1e9))
