"""cProfile of whole warm `cross` runs (CrossIdentifier(..., run_identifier=True): windows, totals, in-silico F1s, interpretation)
at the 1001-Genomes shape"""
import os, sys, tempfile, time, cProfile, pstats
import numpy as np
sys.path.insert(0, '.')
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
# an F2-like sample: windows alternate between two parents (accessions 417 and 800), so the interpretation has work to do
cols = synth.panel_rows(1001, rows, 416, 4)[:, 1]
cols2 = synth.panel_rows(1001, rows, 800, 4)[:, 0]
win = (positions[rows] // 3_000_000) % 2 == 0
codes, wei = synth.planted_sample(rng, np.where(win, cols, cols2), 0.02)
chr_of = np.searchsorted(bounds[1:], rows, side="right")
inp = parsers.ParseInputs("")
inp.load_snp_info(np.array(["Chr%d" % (c + 1) for c in chr_of]), positions[rows], np.array(["0/0", "1/1", "0/1"])[codes], wei, rng.integers(1, 40, n_s))
from snpmatch_amd.core import snpmatch as sm, _report
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
timed(csmatch.CrossIdentifier, "window_genotyper", "windows"); timed(sm.GenotyperOutput, "print_json_output", "json"); timed(sm, "getHeterozygosity", "het")
timed(csmatch.CrossIdentifier, "match_insilico_f1s", "f1s"); timed(csmatch.CrossIdentifier, "cross_interpreter", "interpret")
timed(_report, "dump_json", "(dump_json)"); timed(sm.GenotyperOutput, "print_out_table", "(table)"); timed(engine.Query, "f1_pairs", "(f1 device)")
timed(snp_genotype.Genotype, "get_positions_idxs", "(positions)")
with tempfile.TemporaryDirectory() as tmp:
    for rep in range(5):
        acc.clear()
        t0 = time.perf_counter()
        ci = csmatch.CrossIdentifier(inp, g, "athaliana_tair10", 300000, os.path.join(tmp, "c"), run_identifier=True)
        tot = time.perf_counter() - t0
        print("staged rep %d: %.1f ms  " % (rep, tot * 1e3) + "  ".join("%s %.1f" % (k, v * 1e3) for k, v in acc.items()), flush=True)
    for rep in range(0):
        t0 = time.perf_counter()
        ci = csmatch.CrossIdentifier(inp, g, "athaliana_tair10", 300000, os.path.join(tmp, "c"), run_identifier=True)
        print("rep %d: whole cross %.1f ms, %d window rows" % (rep, (time.perf_counter() - t0) * 1e3, len(ci.windows_data)), flush=True)
    pr = cProfile.Profile(); pr.enable()
    for _ in range(3):
        csmatch.CrossIdentifier(inp, g, "athaliana_tair10", 300000, os.path.join(tmp, "c"), run_identifier=True)
    pr.disable(); pstats.Stats(pr).sort_stats("cumulative").print_stats(40)
