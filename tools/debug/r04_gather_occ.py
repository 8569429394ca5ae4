"""k_fast<GATHER> (int8 panel): kernel time against the resident blocks per CU the part count is derived from (SNPM_OCC_CAP)"""
import os, sys, numpy as np
sys.path.insert(0, '.')
from snpmatch_amd import engine, synth
n_snp = 6_000_000
rng = np.random.default_rng(5)
for n_acc in [int(a) for a in os.environ.get("WIDTHS", "300,600,900,1135,1500,2029,4000,10000").split(",")]:
    for n in (200_000, 1_000_000):
        rows = np.sort(rng.choice(n_snp, size=n, replace=False)).astype(np.int64)
        wei = synth.planted_sample(rng, synth.panel_rows(1001, rows, 0, 4)[:, 1], 0.02)[1]
        line = []
        for cap in (0, 1, 2, 3, 4, 5, 6, 8):
            if cap:
                os.environ["SNPM_OCC_CAP"] = str(cap)
            else:
                os.environ.pop("SNPM_OCC_CAP", None)
            ctx = engine.Context(0)
            panel = engine.Panel(ctx, n_snp, n_acc); panel.fill_synthetic(1001)
            q = engine.Query(panel, rows, wei)
            q.run(1000, False, engine.MODE_EXACT)
            ctx.synchronize(); ctx.profile(True); ctx.profile_reset()
            for _ in range(20):
                q.run(1000, False, engine.MODE_EXACT)
            ctx.synchronize()
            l, ms = ctx.profile_read("fast")
            line.append("%d: %.4f" % (cap, ms / max(l, 1)))
            q.free(); panel.free(); ctx.close()
        print("n_acc %5d n %7d  ms by occ cap  %s" % (n_acc, n, "  ".join(line)), flush=True)
