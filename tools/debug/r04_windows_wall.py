import sys, time, numpy as np
sys.path.insert(0, '.')
import importlib.util
spec = importlib.util.spec_from_file_location("brp", "tools/bench_real_panel.py"); brp = importlib.util.module_from_spec(spec); spec.loader.exec_module(brp)
from snpmatch_amd import engine, synth
ctx = engine.Context(0)
g0, bounds, positions = brp.tair10_layout(brp.N_SNP)
rng = np.random.default_rng(5)
rows = np.sort(rng.choice(brp.N_SNP, size=200000, replace=False)).astype(np.int64)
wei = synth.planted_sample(rng, synth.panel_rows(1001, rows, 416, 4)[:, 1], 0.02)[1]
off = brp.window_offsets(g0, bounds, positions, rows)
for packed in (False, True):
    panel = engine.Panel(ctx, brp.N_SNP, 1135, packed=packed); panel.fill_synthetic(1001)
    q = engine.Query(panel, rows, wei)
    for rep in range(6):
        ctx.profile(True); ctx.profile_reset()
        t0 = time.perf_counter()
        q.run_windows(off, False, fast=True)
        dt = time.perf_counter() - t0
        print("packed", packed, "rep", rep, "wall %.3f ms" % (dt * 1e3), q.last_windows_info, {k: round(ctx.profile_read(k)[1], 4) for k in ("fast", "reduce", "strict", "scan", "lut")})
    q.free(); panel.free()
