#!/usr/bin/env python3
"""Cost of the re-evaluation tier behind a certified pass on the N=8 shard shape (1252 accessions x 50M SNPs): HIP-event times
of the fast pass and of everything after it, with the planted accession's exact-integer score flagged in every step.
usage: tools/time_reeval_tail.py [n_acc=1252] [n_snp=50000000]   (SNPMATCH_HIP_LIB selects another build)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401
import bench  # noqa: E402
from snpmatch_amd import engine  # noqa: E402

n_acc = int(sys.argv[1]) if len(sys.argv) > 1 else 1252
n_snp = int(sys.argv[2]) if len(sys.argv) > 2 else 50_000_000
ctx = engine.Context(0)
panel = engine.Panel(ctx, n_snp, n_acc)
panel.fill_synthetic(bench.SEED)
wei = torch.empty((n_snp, 3), dtype=torch.float64, device="cuda:0")
ctx.sample_synthetic(bench.SEED, 0, n_snp, bench.PLANTED, wei.data_ptr())
q = engine.Query.from_device(panel, None, wei.data_ptr(), n_snp)
for _ in range(3):
    q.run_device(1000, False, engine.MODE_EXACT)
ctx.synchronize()
ctx.profile(True)
ctx.profile_reset()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
K = 10
ctx.synchronize()
import time
t0 = time.perf_counter()
for _ in range(K):
    q.run_device(1000, False, engine.MODE_EXACT)
ctx.synchronize()
dt = (time.perf_counter() - t0) / K * 1e3
out = ["step %.3f ms, flagged %d" % (dt, q.last_reeval())]
for kind in ("fast", "reduce", "strict", "scan"):
    n, ms = ctx.profile_read(kind)
    out.append("%s %.3f ms/step (%d launches)" % (kind, ms / K, n))
print(os.environ.get("SNPMATCH_HIP_LIB", "default lib"), "|", "; ".join(out))
