"""cProfile of CrossIdentifier.window_genotyper at the 1001-Genomes shape (host side of `snpmatch cross`)"""
import cProfile, os, pstats, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from snpmatch_amd import engine, synth
from snpmatch_amd.core import csmatch, genomes, parsers, snp_genotype
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
with tempfile.TemporaryDirectory() as tmp:
    for rep in range(3):
        t0 = time.perf_counter()
        ci = csmatch.CrossIdentifier(inp, g, "athaliana_tair10", 300000, os.path.join(tmp, "c"), run_identifier=False)
        ci.window_genotyper(os.path.join(tmp, "c.windowscore.txt"))
        print("window_genotyper %.1f ms" % ((time.perf_counter() - t0) * 1e3))
    pr = cProfile.Profile(); pr.enable()
    ci = csmatch.CrossIdentifier(inp, g, "athaliana_tair10", 300000, os.path.join(tmp, "c"), run_identifier=False)
    ci.window_genotyper(os.path.join(tmp, "c.windowscore.txt"))
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
    t0 = time.perf_counter()
    ci2 = csmatch.CrossIdentifier(inp, g, "athaliana_tair10", 300000, os.path.join(tmp, "d"), run_identifier=True)
    print("whole cross_identifier %.1f ms" % ((time.perf_counter() - t0) * 1e3))
    pr = cProfile.Profile(); pr.enable()
    ci2 = csmatch.CrossIdentifier(inp, g, "athaliana_tair10", 300000, os.path.join(tmp, "d"), run_identifier=True)
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
    # where do the 18 ms per Query of cross_identifier go?
    import time as _t
    real_init = engine.Query.__init__
    def timed_init(self, *a, **k):
        ctx.synchronize(); t0 = _t.perf_counter(); real_init(self, *a, **k); ctx.synchronize()
        print("  Query.__init__ %.2f ms (n=%d)" % ((_t.perf_counter() - t0) * 1e3, self.n))
    engine.Query.__init__ = timed_init
    real_f1 = engine.Query.f1_pairs
    def timed_f1(self, *a, **k):
        ctx.synchronize(); t0 = _t.perf_counter(); r = real_f1(self, *a, **k); ctx.synchronize()
        print("  f1_pairs %.2f ms" % ((_t.perf_counter() - t0) * 1e3)); return r
    engine.Query.f1_pairs = timed_f1
    for rep in range(2):
        t0 = _t.perf_counter()
        ci2 = csmatch.CrossIdentifier(inp, g, "athaliana_tair10", 300000, os.path.join(tmp, "d"), run_identifier=True)
        print("whole cross_identifier %.1f ms" % ((_t.perf_counter() - t0) * 1e3))
