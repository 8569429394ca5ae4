"""k_fast<GATHER> on one 200k-row sample (1135 x 11M): kernel time against the number of parts (SNPM_DEBUG_MAX_PARTS caps it)"""
import os, sys, numpy as np
sys.path.insert(0, '.')
from snpmatch_amd import engine, synth
n_snp, n_acc = 11_000_000, 1135
rng = np.random.default_rng(5)
for n in (200_000, 50_000, 1_000_000):
    rows = np.sort(rng.choice(n_snp, size=n, replace=False)).astype(np.int64)
    wei = synth.planted_sample(rng, synth.panel_rows(1001, rows, 416, 4)[:, 1], 0.02)[1]
    for packed in (False, True):
        for cap in [0] + [int(c) for c in os.environ.get("CAPS", "2048,1563,1280,1042,1024,782,768,626,521,512,391,256").split(",")]:
            if cap:
                os.environ["SNPM_DEBUG_MAX_PARTS"] = str(cap)
            else:
                os.environ.pop("SNPM_DEBUG_MAX_PARTS", None)
            ctx = engine.Context(0)
            panel = engine.Panel(ctx, n_snp, n_acc, packed=packed); panel.fill_synthetic(1001)
            q = engine.Query(panel, rows, wei)
            q.run(1000, False, engine.MODE_EXACT)
            ctx.synchronize(); ctx.profile(True); ctx.profile_reset()
            for _ in range(20):
                q.run(1000, False, engine.MODE_EXACT)
            ctx.synchronize()
            l, ms = ctx.profile_read("fast")
            lr, msr = ctx.profile_read("reduce")
            print("n %7d packed %d cap %5d: k_fast %.4f ms  reduce %.4f ms" % (n, packed, cap, ms / max(l, 1), msr / max(lr, 1) * (lr / max(l, 1))), flush=True)
            q.free(); panel.free(); ctx.close()
