"""the --refine steps at the 1001-Genomes shape: identify_segregating_snps over the resident panel, then the second genotyper pass"""
import os, sys, tempfile, time
import numpy as np
sys.path.insert(0, '.')
from snpmatch_amd import engine, synth
from snpmatch_amd.core import genomes, parsers, snp_genotype, snpmatch
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
    gt = snpmatch.Genotyper(inp, g, os.path.join(tmp, "o"), run_genotyper=False)
    top = np.array([100, 417, 800])
    others = np.setdiff1d(np.arange(n_acc), top)
    for rep in range(3):
        t0 = time.perf_counter()
        seg = g.identify_segregating_snps(top)
        t1 = time.perf_counter()
        res = gt.genotyper(filter_pos_ix=seg, mask_acc_ix=others)
        t2 = time.perf_counter()
        ta = time.perf_counter()
        res2 = gt.genotyper(mask_acc_ix=others, _filter_mask=g.segregating_mask(top))
        tb = time.perf_counter()
        assert np.array_equal(res.scores, res2.scores) and np.array_equal(res.ninfo, res2.ninfo) and res.num_snps == res2.num_snps
        print("   the same through the mask (filter_tophits' path): %.1f ms" % ((tb - ta) * 1e3))
        res.print_out_table(os.path.join(tmp, "o.refined.scores.txt"))
        t3 = time.perf_counter()
        print("rep %d: identify_segregating_snps %.1f ms (%d of %d rows segregate), genotyper(filter, mask) %.1f ms (%d rows), table %.1f ms" % (
            rep, (t1 - t0) * 1e3, len(seg), n_snp, (t2 - t1) * 1e3, res.num_snps, (t3 - t2) * 1e3), flush=True)
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    seg = g.identify_segregating_snps(top); res = gt.genotyper(filter_pos_ix=seg, mask_acc_ix=others)
    pr.disable(); pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
