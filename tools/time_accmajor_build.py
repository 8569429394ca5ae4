#!/usr/bin/env python3
"""One-time cost of the accession-major packed copy: first vs second exact run with a forced re-evaluation."""
import os
import sys
import time

os.environ["SNPM_DEBUG_REEVAL"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from snpmatch_amd import engine  # noqa: E402

n_acc, n_snp = int(sys.argv[1]), int(sys.argv[2])
ctx = engine.Context(0)
panel = engine.Panel(ctx, n_snp, n_acc)
panel.fill_synthetic(bench.SEED)
wei = bench.make_sample(n_snp, bench.SEED, bench.PLANTED)
q = engine.Query(panel, None, wei)
q.run(1000, False, engine.MODE_FAST)
ctx.synchronize()
for i in range(3):
    t0 = time.perf_counter()
    q.run(1000, False, engine.MODE_EXACT)
    print("exact run %d: %.1f ms" % (i, (time.perf_counter() - t0) * 1e3))
