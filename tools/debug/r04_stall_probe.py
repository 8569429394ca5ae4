"""How often does a short call stall?  N one-call samples in a row (1135 x 2M int8, 200k-SNP coded sample); prints the median,
the mean, and where the calls above 2 ms sit (period?).  Environment knobs of the HIP runtime are tried from outside."""
import os, sys, time, numpy as np
sys.path.insert(0, '.')
from snpmatch_amd import engine, synth
n_snp, n_acc, n = 2_000_000, 1135, 200_000
N = int(os.environ.get("PROBE_CALLS", 3000))
rng = np.random.default_rng(5)
rows = np.sort(rng.choice(n_snp, size=n, replace=False)).astype(np.int64)
wei = synth.planted_sample(rng, synth.panel_rows(1001, rows, 416, 4)[:, 1], 0.02)[1]
tab = np.concatenate([engine.pl_table(256), [0.0]])
codes = engine.weight_codes(wei, tab)
ctx = engine.Context(0)
panel = engine.Panel(ctx, n_snp, n_acc); panel.fill_synthetic(1001)
q = engine.Query(panel, rows, wei)
for what in os.environ.get("PROBE_WHAT", "once,rerun").split(","):
    run = (lambda: panel.genotype_once(rows, codes, None, table=tab)) if what == "once" else (lambda: q.run(1000, False, engine.MODE_EXACT))
    for _ in range(20):
        run()
    t = np.empty(N)
    for i in range(N):
        t0 = time.perf_counter(); run(); t[i] = time.perf_counter() - t0
    slow = np.flatnonzero(t > 2e-3)
    print("%-6s %d calls: median %.3f ms  mean %.3f ms  p99 %.3f ms  max %.1f ms  calls > 2 ms: %d at %s (%s ms)" % (
        what, N, np.median(t) * 1e3, t.mean() * 1e3, np.quantile(t, 0.99) * 1e3, t.max() * 1e3, len(slow), slow[:12].tolist(),
        " ".join("%.0f" % (x * 1e3) for x in t[slow[:12]])), flush=True)
