import sys, time, os, numpy as np
sys.path.insert(0, '.')
os.environ["SNPM_ONCE_TRACE"] = "1"
from snpmatch_amd import engine, synth
ctx = engine.Context(0)
n_snp, n_acc, n = 11_000_000, 1135, 200_000
panel = engine.Panel(ctx, n_snp, n_acc); panel.fill_synthetic(1001)
rng = np.random.default_rng(5)
rows = np.sort(rng.choice(n_snp, size=n, replace=False)).astype(np.int64)
wei = synth.planted_sample(rng, synth.panel_rows(1001, rows, 416, 4)[:, 1], 0.02)[1]
n_in = 250000
sidx = np.sort(rng.choice(n_in, size=n, replace=False)).astype(np.int64)
wall = np.zeros((n_in, 3)); wall[sidx] = wei
for i in range(8):
    t0 = time.perf_counter()
    out = panel.genotype_once(rows, wall, sidx)
    print("wall %.3f ms top %d" % ((time.perf_counter() - t0) * 1e3, int(np.nanargmin(out["lik"]))), flush=True)
tab = np.concatenate([engine.pl_table(256), [0.0]])
codes = engine.weight_codes(wall, tab)
print("coded weights:", codes is not None)
for i in range(8):
    t0 = time.perf_counter()
    out = panel.genotype_once(rows, codes, sidx, table=tab)
    print("coded wall %.3f ms top %d" % ((time.perf_counter() - t0) * 1e3, int(np.nanargmin(out["lik"]))), flush=True)
