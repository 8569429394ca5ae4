import sys, time, os, numpy as np
sys.path.insert(0, '.')
from snpmatch_amd import engine, synth
ctx = engine.Context(0)
n_snp, n_acc, n = 11_000_000, 1135, 200_000
panel = engine.Panel(ctx, n_snp, n_acc); panel.fill_synthetic(1001); ctx.synchronize()
rng = np.random.default_rng(5)
rows = np.sort(rng.choice(n_snp, size=n, replace=False)).astype(np.int64)
wei = synth.planted_sample(rng, synth.panel_rows(1001, rows, 416, 4)[:, 1], 0.02)[1]
n_in = 250000
sidx = np.sort(rng.choice(n_in, size=n, replace=False)).astype(np.int64)
wall = np.zeros((n_in, 3)); wall[sidx] = wei
def three():
    q = engine.Query(panel, rows, wei); s, nn = q.run(1000, False, engine.MODE_EXACT); lik, _ = ctx.likelihood(s, nn, truncate=True); q.free()
for phase in ("fresh", "after three-call queries", "with profiling on"):
    if phase == "after three-call queries":
        for _ in range(12): three()
    if phase == "with profiling on":
        ctx.profile(True); ctx.profile_reset()
    ts = []
    for i in range(8):
        t0 = time.perf_counter(); out = panel.genotype_once(rows, wall, sidx); ts.append((time.perf_counter() - t0) * 1e3)
    print(phase, " ".join("%.2f" % t for t in ts))
