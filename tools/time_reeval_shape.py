"""step time of a certified run with k forced re-evaluations vs the fast pass alone, any shape
usage: python tools/time_reeval_shape.py N_SNP N_ACC [packed=0] [reeval=6]"""
import os, sys, time
n_snp, n_acc = int(sys.argv[1]), int(sys.argv[2])
packed = len(sys.argv) > 3 and sys.argv[3] == "1"
k = int(sys.argv[4]) if len(sys.argv) > 4 else 6
os.environ["SNPM_DEBUG_REEVAL"] = str(k)
sys.path.insert(0, os.getcwd())
import numpy as np
import bench
from snpmatch_amd import engine
ctx = engine.Context(0)
panel = engine.Panel(ctx, n_snp, n_acc, packed=packed)
panel.fill_synthetic(bench.SEED)
wei = bench.make_sample(n_snp, bench.SEED, bench.PLANTED)
q = engine.Query(panel, None, wei)
for mode, name in ((engine.MODE_FAST, "fast"), (engine.MODE_EXACT, "exact")):
    q.run_device(1000, False, mode); ctx.synchronize()
    ctx.profile(True); ctx.profile_reset()
    t0 = time.perf_counter()
    for i in range(5):
        q.run_device(1000, False, mode)
    ctx.synchronize()
    nre = q.last_reeval()
    dt = (time.perf_counter() - t0) / 5 * 1e3
    parts = {kk: ctx.profile_read(kk) for kk in ("fast", "reduce", "strict", "scan")}
    ctx.profile(False)
    print("%d x %d packed=%d %s: %.3f ms/step  reeval %d  " % (n_snp, n_acc, packed, name, dt, nre) +
          "  ".join("%s %.3f" % (kk, v[1] / max(v[0], 1)) for kk, v in parts.items()), flush=True)
