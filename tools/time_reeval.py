import os, sys, time
os.environ["SNPM_DEBUG_REEVAL"] = "1"
sys.path.insert(0, os.getcwd())
import bench
from snpmatch_amd import engine
ctx = engine.Context(0)
panel = engine.Panel(ctx, 50000000, 1252)
panel.fill_synthetic(bench.SEED)
wei = bench.make_sample(50000000, bench.SEED, bench.PLANTED)
q = engine.Query(panel, None, wei)
q.run(1000, False, engine.MODE_EXACT)
ctx.profile(True); ctx.profile_reset()
for i in range(5): q.run(1000, False, engine.MODE_EXACT)
for k in ("fast","strict","scan"):
    n, ms = ctx.profile_read(k); print(os.environ.get("SNPMATCH_HIP_LIB","default")[-12:], k, n, "%.3f ms avg" % (ms/max(n,1)))
