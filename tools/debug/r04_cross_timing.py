"""where the wall time of warm `cross` runs goes, run by run (the two-rep tool sees 13 ... 50 ms)"""
import os, sys, tempfile, time, gc
import numpy as np
sys.path.insert(0, '.')
from snpmatch_amd import engine, synth
from snpmatch_amd.core import csmatch, genomes, parsers, snp_genotype, snpmatch, _report
import pandas as pd
n_snp, n_acc, n_s = 11_000_000, 1135, 200_000
g0 = genomes.Genome("athaliana_tair10")
frac = np.cumsum(g0.chrlen) / g0.chrlen.sum()
bounds = np.concatenate([[0], np.round(frac * n_snp).astype(np.int64)])
positions = np.concatenate([1 + (np.arange(bounds[c + 1] - bounds[c]) * int(g0.chrlen[c] - 1)) // int(bounds[c + 1] - bounds[c]) for c in range(5)])
regions = [(int(bounds[c]), int(bounds[c + 1])) for c in range(5)]
ctx = engine.default_context()
panel = engine.Panel(ctx, n_snp, n_acc, packed=os.environ.get("PACKED", "0") == "1"); panel.fill_synthetic(1001)
g = snp_genotype.Genotype.from_arrays(np.zeros((0, n_acc), dtype=np.int8), [str(i) for i in range(n_acc)], positions, ["1", "2", "3", "4", "5"], regions)
g._panel = panel
rng = np.random.default_rng(5)
rows = np.sort(rng.choice(n_snp, size=n_s, replace=False))
codes, wei = synth.planted_sample(rng, synth.panel_rows(1001, rows, 416, 4)[:, 1], 0.02)
chr_of = np.searchsorted(bounds[1:], rows, side="right")
inp = parsers.ParseInputs("")
inp.load_snp_info(np.array(["Chr%d" % (c + 1) for c in chr_of]), positions[rows], np.array(["0/0", "1/1", "0/1"])[codes], wei, rng.integers(1, 40, n_s))
acc = {}
def timed(obj, name, label):
    f = getattr(obj, name)
    def w(*a, **k):
        t0 = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            acc[label] = acc.get(label, 0.0) + time.perf_counter() - t0
    setattr(obj, name, w)
timed(csmatch, "_window_segments", "segments"); timed(engine.Query, "__init__", "query"); timed(engine.Query, "run_windows", "run_windows")
timed(engine.Query, "free", "free"); timed(engine.Context, "likelihood", "likelihood"); timed(engine.Context, "binom_identity", "identity")
timed(_report, "window_table", "table"); timed(pd.DataFrame, "to_csv", "to_csv"); timed(csmatch.CrossIdentifier, "__init__", "init")
if os.environ.get("NOGC") == "1":
    gc.disable()
with tempfile.TemporaryDirectory() as tmp:
    for rep in range(int(os.environ.get("REPS", 12))):
        acc.clear()
        t0 = time.perf_counter()
        ci = csmatch.CrossIdentifier(inp, g, "athaliana_tair10", 300000, os.path.join(tmp, "c"), run_identifier=False)
        ci.window_genotyper(os.path.join(tmp, "c.windowscore.txt"))
        tot = time.perf_counter() - t0
        print("rep %2d: %.1f ms  " % (rep, tot * 1e3) + "  ".join("%s %.1f" % (k, v * 1e3) for k, v in acc.items()) + "  other %.1f" % ((tot - sum(acc.values())) * 1e3), flush=True)
