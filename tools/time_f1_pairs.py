"""Time snpm_query_f1_pairs (45 in-silico crosses of 10 accessions) on synthetic panels."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from snpmatch_amd import engine, synth  # noqa: E402

ctx = engine.Context(0)
for n_snp, n_acc, n_match in ((11000000, 1135, 200000), (6250000, 10000, None), (25000000, 2500, None)):
    panel = engine.Panel(ctx, n_snp, n_acc)
    panel.fill_synthetic(1001)
    rng = np.random.default_rng(1)
    rows = None if n_match is None else np.sort(rng.choice(n_snp, size=n_match, replace=False)).astype(np.int64)
    n = n_snp if n_match is None else n_match
    wei = synth.sample_weights(rng, rng.choice(np.array([0, 1, 2], dtype=np.int8), size=n, p=[0.6, 0.35, 0.05]), 0.8)
    q = engine.Query(panel, rows, wei)
    sel = rng.choice(n_acc, size=10, replace=False)
    q.f1_pairs(sel)
    t0 = time.perf_counter()
    s, ni = q.f1_pairs(sel)
    dt = time.perf_counter() - t0
    print("panel %d x %d, %d matched SNPs: 45 crosses in %.2f ms (ninfo[0] %d)" % (n_snp, n_acc, n, dt * 1e3, ni[0]), flush=True)
    q.free()
    panel.free()
