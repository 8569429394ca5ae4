#!/usr/bin/env python3
"""End-to-end timing of Genotyper / CrossIdentifier at the 1001-Genomes shape (config 2 / 3)."""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snpmatch_amd import engine, synth  # noqa: E402
from snpmatch_amd.core import csmatch, genomes, parsers, snp_genotype, snpmatch  # noqa: E402

n_snp, n_acc, n_s = int(os.environ.get("N_SNP", 11_000_000)), 1135, 200_000
g0 = genomes.Genome("athaliana_tair10")
frac = np.cumsum(g0.chrlen) / g0.chrlen.sum()
bounds = np.concatenate([[0], np.round(frac * n_snp).astype(np.int64)])
positions = np.concatenate([1 + (np.arange(bounds[c + 1] - bounds[c]) * int(g0.chrlen[c] - 1)) // int(bounds[c + 1] - bounds[c])
                            for c in range(5)])
regions = [(int(bounds[c]), int(bounds[c + 1])) for c in range(5)]
t = time.perf_counter()
ctx = engine.default_context()
panel = engine.Panel(ctx, n_snp, n_acc, packed=os.environ.get("PACKED", "0") == "1")     # PACKED=1: the 2-bit panel (the drop-in classes' default residency)
panel.fill_synthetic(1001)
g = snp_genotype.Genotype.from_arrays(np.zeros((0, n_acc), dtype=np.int8), [str(i) for i in range(n_acc)], positions,
                                      ["1", "2", "3", "4", "5"], regions)
g._panel = panel                       # DB already resident (device-generated)
print("setup %.2f s" % (time.perf_counter() - t))
rng = np.random.default_rng(5)
rows = np.sort(rng.choice(n_snp, size=n_s, replace=False))
col = synth.panel_rows(1001, rows, 416, 4)[:, 1]
codes, wei = synth.planted_sample(rng, col, 0.02)
chr_of = np.searchsorted(bounds[1:], rows, side="right")
inp = parsers.ParseInputs("")
inp.load_snp_info(np.array(["Chr%d" % (c + 1) for c in chr_of]), positions[rows],
                  np.array(["0/0", "1/1", "0/1"])[codes], wei, rng.integers(1, 40, n_s))
with tempfile.TemporaryDirectory() as tmp:
    for rep in range(2):
        t0 = time.perf_counter()
        gt = snpmatch.Genotyper(inp, g, os.path.join(tmp, "o"), run_genotyper=False)
        t1 = time.perf_counter()
        gt.get_common_positions()
        t2 = time.perf_counter()
        res = gt.genotyper()
        t3 = time.perf_counter()
        gt.write_genotyper_output(res)
        t4 = time.perf_counter()
        print("inbred rep%d: init %.3f  common_positions %.3f  genotyper(total) %.3f  outputs %.3f  | total %.3f s"
              % (rep, t1 - t0, t2 - t1, t3 - t2, t4 - t3, t4 - t0))
    assert len(gt.commonSNPs[0]) == n_s and int(np.nanargmin(res.likelis)) == 417
    for rep in range(2):
        t0 = time.perf_counter()
        ci = csmatch.CrossIdentifier(inp, g, "athaliana_tair10", 300000, os.path.join(tmp, "c"), run_identifier=False)
        r = ci.window_genotyper(os.path.join(tmp, "c.windowscore.txt"))
        t1 = time.perf_counter()
        print("cross  rep%d: window_genotyper %.3f s (%d windows, %d table rows)" % (rep, t1 - t0, 399, len(ci.windows_data)))
    if os.environ.get("PROFILE") == "1":             # where the host time of the warm runs goes (cProfile, 5 runs each)
        import cProfile
        import pstats

        def inbred_once():
            gt = snpmatch.Genotyper(inp, g, os.path.join(tmp, "o"), run_genotyper=False)
            gt.get_common_positions()
            gt.write_genotyper_output(gt.genotyper())

        def cross_once():
            ci = csmatch.CrossIdentifier(inp, g, "athaliana_tair10", 300000, os.path.join(tmp, "c"), run_identifier=False)
            ci.window_genotyper(os.path.join(tmp, "c.windowscore.txt"))

        for name, fn in (("inbred", inbred_once), ("cross", cross_once)):
            pr = cProfile.Profile()
            pr.enable()
            for _ in range(5):
                fn()
            pr.disable()
            print("== cProfile, 5 warm runs of %s" % name)
            pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
