#!/usr/bin/env python3
"""
Generate the golden vectors under tests/golden/ by RUNNING THE UNMODIFIED REFERENCE
(/root/reference, SNPmatch v5.0.1) in the build container.

The reference cannot travel to the GPU box; these small fixtures (inputs + the reference's
outputs) can.  Run from the repo root:   python tests/golden/make_golden.py

How the reference is made importable here (nothing in /root/reference is modified or copied):
  * `allel`, `h5py`, `hmmlearn(.hmm)` are absent from this image and are only imported at the top
    of reference files for VCF/HDF5 reading and the HMM (none of which is on the scoring path).
    Empty placeholder modules are put into sys.modules so the import statements succeed; none
    of their attributes is ever touched by the code run below.
  * pandas >= 2 removed DataFrame.append, which `window_genotyper` (core/csmatch.py:91) still
    calls; a two-line shim (concat) is installed for the cross goldens only, and those fixtures
    are labelled "pandas2-append-shim".
  * the HDF5-backed `g.g` / `g.g_acc` objects are replaced by an in-memory class exposing the
    same attributes (`snps`, `accessions`, `positions`, `chrs`, `chr_regions`, `chromosomes`),
    as SURVEY.md section 8b/8c describes.

Outputs (all under tests/golden/):
  g1_match.npz       matchGTsAccs inputs and outputs (fp64 bit patterns, ninfo)
  g2_inbred.npz/json Genotyper end to end on a toy DB: commonSNPs, ScoreList, files' text
  g3_refine.json     --refine path outputs
  g4_likelihood.npz  likeliTest / calculate_likelihoods grid
  g5_cross.npz/json  per-window scores, get_window_data rows, np_test_identity, cross files' text
  g6_common.npz      get_common_positions edge cases
  g7_cross_*         F1-like and F2-like samples through the whole cross pipeline (interpreter cases >= 3)
  g1b_single_acc.npz, g2b_g5b_single_acc.npz/json, toy_db_single.npz
                     the same for panels of ONE accession (numpy sums a contiguous axis pairwise: another order)
"""
import io
import json
import os
import sys
import tempfile
import types
import warnings

import numpy as np

sys.dont_write_bytecode = True
for _m in ("allel", "h5py", "hmmlearn", "hmmlearn.hmm"):
    sys.modules[_m] = types.ModuleType(_m)
sys.modules["hmmlearn"].hmm = sys.modules["hmmlearn.hmm"]
sys.path.insert(0, "/root/reference")
warnings.filterwarnings("ignore")

import pandas as pd  # noqa: E402
from snpmatch.core import csmatch as ref_cs  # noqa: E402
from snpmatch.core import genomes as ref_genomes  # noqa: E402
from snpmatch.core import parsers as ref_parsers  # noqa: E402
from snpmatch.core import snp_genotype as ref_sg  # noqa: E402
from snpmatch.core import snpmatch as ref_sm  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
DB_P = [0.05, 0.60, 0.33, 0.02]          # P(-1, 0, 1, 2), SURVEY.md 8d
DB_V = np.array([-1, 0, 1, 2], dtype=np.int8)


# ----------------------------------------------------------------------------- helpers
def make_weights(rng, n, frac_pl=0.8, codes=None):
    """80 % PL-derived exp(-PL/10) rows (integer PL, min 0), 20 % hard one-hot rows."""
    if codes is None:
        codes = rng.choice(np.array([0, 1, 2], dtype=np.int8), size=n, p=[0.6, 0.35, 0.05])
    col_of = {0: 0, 2: 1, 1: 2}
    wei = np.zeros((n, 3))
    is_pl = rng.random(n) < frac_pl
    pl = rng.integers(1, 256, size=(n, 3))
    for i in range(n):
        c = col_of[int(codes[i])]
        if is_pl[i]:
            row = pl[i].astype(float)
            row[c] = 0
            wei[i] = np.exp(row / (-10))          # core/parsers.py:149-150
        else:
            wei[i, c] = 1.0                       # core/parsers.py:135-138
    return wei, codes


class MemGeno(object):
    """In-memory stand-in for pygwas HDF5Genotype (attributes used by the path only)."""

    def __init__(self, snps, accessions, positions, chrs, chr_regions):
        self.snps = snps
        self.accessions = np.array(accessions, dtype="S")
        self.positions = np.array(positions, dtype="i4")
        self.chrs = np.array(chrs, dtype="U")
        self.chr_regions = np.array(chr_regions, dtype=int)

    @property
    def chromosomes(self):
        out = []
        for i, reg in enumerate(self.chr_regions):
            out.extend([self.chrs[i]] * int(reg[1] - reg[0]))
        return out


def make_genotype(snps, accessions, positions, chrs, chr_regions):
    g = ref_sg.Genotype.__new__(ref_sg.Genotype)
    g.g = MemGeno(snps, accessions, positions, chrs, chr_regions)
    g.g_acc = MemGeno(snps, accessions, positions, chrs, chr_regions)
    g.accessions = g.g.accessions.astype("U")
    g.chrs = g.g.chrs.astype("U")
    return g


def make_inputs(chrs, pos, gt, wei, dp):
    inp = ref_parsers.ParseInputs("")
    inp.load_snp_info(chrs, pos, gt, wei, dp)
    return inp


def toy_db(rng, chrlens, n_per_chr, n_acc, near_identical=()):
    positions, regions, start = [], [], 0
    for L in chrlens:
        p = np.sort(rng.choice(np.arange(1, L + 1), size=n_per_chr, replace=False))
        positions.append(p)
        regions.append((start, start + n_per_chr))
        start += n_per_chr
    positions = np.concatenate(positions)
    snps = rng.choice(DB_V, size=(len(positions), n_acc), p=DB_P)
    for (src, dst, rate) in near_identical:
        col = snps[:, src].copy()
        flip = rng.random(len(col)) < rate
        col[flip] = rng.choice(DB_V, size=int(flip.sum()), p=DB_P)
        snps[:, dst] = col
    accessions = [str(6000 + 7 * i) for i in range(n_acc)]
    chrs = [str(i + 1) for i in range(len(chrlens))]
    return snps, accessions, positions, chrs, regions


def toy_sample(rng, snps, positions, regions, chrlens, n_hit, n_miss, planted, err, prefix="Chr"):
    n_db = len(positions)
    hit = np.sort(rng.choice(n_db, size=n_hit, replace=False))
    chr_of_row = np.zeros(n_db, dtype=int)
    for i, (a, b) in enumerate(regions):
        chr_of_row[a:b] = i
    rows = []
    for r in hit:
        rows.append((chr_of_row[r], int(positions[r]), int(r)))
    dbset = set((int(chr_of_row[r]), int(positions[r])) for r in range(n_db))
    while len(rows) < n_hit + n_miss:
        c = int(rng.integers(0, len(chrlens)))
        p = int(rng.integers(1, chrlens[c] + 1))
        if (c, p) not in dbset:
            dbset.add((c, p))
            rows.append((c, p, -1))
    rows.sort()
    codes = np.zeros(len(rows), dtype=np.int8)
    for i, (c, p, r) in enumerate(rows):
        if r >= 0 and snps[r, planted] >= 0:
            codes[i] = snps[r, planted]
        else:
            codes[i] = rng.choice(np.array([0, 1], dtype=np.int8))
        if rng.random() < err:
            codes[i] = rng.choice(np.array([0, 1, 2], dtype=np.int8))
    wei, _ = make_weights(rng, len(rows), codes=codes)
    gt_of = {0: "0/0", 1: "1/1", 2: "0/1"}
    chrs = np.array(["%s%d" % (prefix, c + 1) for (c, p, r) in rows])
    pos = np.array([p for (c, p, r) in rows], dtype=int)
    gt = np.array([gt_of[int(k)] for k in codes])
    dp = rng.integers(1, 40, size=len(rows))
    return chrs, pos, gt, wei, dp


def read_text(path):
    with open(path) as fh:
        return fh.read()


# ----------------------------------------------------------------------------- G1
def g1_match():
    rng = np.random.default_rng(701501)
    out = {}
    cases = [(1, 1), (7, 3), (64, 1), (337, 64), (1000, 257), (129, 1135)]
    names = []
    for (n, n_acc) in cases:
        db = rng.choice(DB_V, size=(n, n_acc), p=DB_P)
        for kind in ("pl", "hard"):
            wei, _ = make_weights(rng, n, frac_pl=0.8 if kind == "pl" else 0.0)
            for skip in (False, True):
                s, ni = ref_sm.matchGTsAccs(wei, db.copy(), skip)
                key = "n%d_a%d_%s_%d" % (n, n_acc, kind, int(skip))
                names.append(key)
                out[key + "_db"] = db
                out[key + "_wei"] = wei
                out[key + "_score"] = np.asarray(s, dtype=np.float64)
                out[key + "_ninfo"] = np.asarray(ni, dtype=np.int64)
    # edge: all-missing column, values outside {-1,0,1,2}, and a row with zero weights
    db = rng.choice(DB_V, size=(50, 9), p=DB_P)
    db[:, 0] = -1
    db[:, 1] = 3
    db[5, :] = -2
    wei, _ = make_weights(rng, 50)
    wei[7] = 0.0
    for skip in (False, True):
        s, ni = ref_sm.matchGTsAccs(wei, db.copy(), skip)
        key = "edge_%d" % int(skip)
        names.append(key)
        out[key + "_db"] = db
        out[key + "_wei"] = wei
        out[key + "_score"] = np.asarray(s, dtype=np.float64)
        out[key + "_ninfo"] = np.asarray(ni, dtype=np.int64)
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "g1_match.npz"), **out)
    print("g1: %d cases" % len(names))


# ----------------------------------------------------------------------------- G2 / G3
TOY_CHRLENS = [30427671, 19698289, 23459830, 18585056, 26975502]   # TAIR10 lengths (genome json)


def build_toy(seed, n_acc=50, near=(), planted=17):
    rng = np.random.default_rng(seed)
    snps, accs, positions, chrs, regions = toy_db(rng, TOY_CHRLENS, 2000, n_acc, near_identical=near)
    s_chrs, s_pos, s_gt, s_wei, s_dp = toy_sample(rng, snps, positions, regions, TOY_CHRLENS,
                                                  n_hit=2400, n_miss=600, planted=planted, err=0.03)
    return dict(snps=snps, accs=np.array(accs), positions=positions, chrs=np.array(chrs),
                regions=np.array(regions), s_chrs=s_chrs, s_pos=s_pos, s_gt=s_gt, s_wei=s_wei, s_dp=s_dp)


def run_inbred(toy, skip_db_hets, refine, tmp):
    g = make_genotype(toy["snps"].copy(), toy["accs"], toy["positions"], toy["chrs"], toy["regions"])
    inputs = make_inputs(toy["s_chrs"], toy["s_pos"], toy["s_gt"], toy["s_wei"], toy["s_dp"])
    outp = os.path.join(tmp, "inbred")
    files = {}
    if refine:
        gt = ref_sm.Genotyper(inputs, g, outp, run_genotyper=False, skip_db_hets=skip_db_hets)
        gt.filter_tophits()
        if os.path.exists(outp + ".refined.scores.txt"):
            files["refined.scores.txt"] = read_text(outp + ".refined.scores.txt")
    else:
        gt = ref_sm.Genotyper(inputs, g, outp, run_genotyper=True, skip_db_hets=skip_db_hets)
    files["scores.txt"] = read_text(outp + ".scores.txt")
    files["matches.json"] = read_text(outp + ".matches.json")
    return gt, files


def g2_g3():
    with tempfile.TemporaryDirectory() as tmp:
        toy = build_toy(1001)
        np.savez_compressed(os.path.join(OUT, "toy_db.npz"), **toy)
        res = {}
        arrays = {}
        for skip in (False, True):
            gt, files = run_inbred(toy, skip, False, tmp)
            res["inbred_skip%d" % int(skip)] = files
            # raw accumulators before truncation: re-run the chunk loop through the reference kernel
            score = np.zeros(len(toy["accs"]))
            ninfo = np.zeros(len(toy["accs"]), dtype="uint32")
            c0, c1 = gt.commonSNPs
            for j in range(0, len(c0), 1000):
                s, ni = ref_sm.matchGTsAccs(toy["s_wei"][c1[j:j + 1000]], toy["snps"][c0[j:j + 1000], :].copy(), skip)
                score = score + s
                ninfo = ninfo + ni
            arrays["common_db"] = np.asarray(c0, dtype=np.int64)
            arrays["common_sample"] = np.asarray(c1, dtype=np.int64)
            arrays["score_skip%d" % int(skip)] = score
            arrays["ninfo_skip%d" % int(skip)] = np.asarray(ninfo, dtype=np.int64)
            assert np.array_equal(np.array(score, dtype=int), gt.result.scores)
        np.savez_compressed(os.path.join(OUT, "g2_inbred.npz"), **arrays)
        with open(os.path.join(OUT, "g2_inbred.json"), "w") as fh:
            json.dump(res, fh, indent=1, sort_keys=True)
        print("g2: inbred files captured, %d common SNPs" % len(arrays["common_db"]))

        # G3: near-identical accessions so that --refine has work to do
        toy3 = build_toy(1002, n_acc=40, near=((17, 18, 0.004), (17, 19, 0.006), (17, 20, 0.01)))
        np.savez_compressed(os.path.join(OUT, "toy_db_refine.npz"), **toy3)
        gt, files = run_inbred(toy3, False, True, tmp)
        files["has_result_fine"] = hasattr(gt, "result_fine")
        with open(os.path.join(OUT, "g3_refine.json"), "w") as fh:
            json.dump(files, fh, indent=1, sort_keys=True)
        print("g3: refine captured (refined table: %s)" % ("refined.scores.txt" in files))


# ----------------------------------------------------------------------------- G4
def g4_likelihood():
    ns, ys = [], []
    for n in (0, 1, 2, 10, 11, 4987, 5194, 4933, 100000, 7545):
        for y in sorted(set([0, 1, n // 3, n // 2, max(n - 1, 0), n])):
            if y <= n:
                ns.append(n)
                ys.append(float(y))
    rng = np.random.default_rng(4)
    for _ in range(200):                      # fractional y as in cross windows (csmatch.py:47)
        n = int(rng.integers(1, 3000))
        ns.append(n)
        ys.append(float(rng.random() * n))
    ns = np.array(ns, dtype=np.int64)
    ys = np.array(ys)
    lik = np.array([ref_sm.likeliTest(int(n), y if y != int(y) else int(y)) for n, y in zip(ns, ys)], dtype=float)
    # calculate_likelihoods on vectors (nanmin + ratio), int and float scores
    sc_i = np.array([4946, 4861, 4368, 0, 10, 3], dtype=int)
    ni_i = np.array([4987, 5194, 4933, 10, 10, 10], dtype=int)
    l_i, r_i = ref_sm.GenotyperOutput.calculate_likelihoods(sc_i, ni_i)
    sc_f = np.array([10.25, 3.5, 8.0, 0.0, 11.0])
    ni_f = np.array([11, 11, 11, 11, 11], dtype=int)
    l_f, r_f = ref_sm.GenotyperOutput.calculate_likelihoods(sc_f, ni_f)
    l_a, r_a = ref_sm.GenotyperOutput.calculate_likelihoods(sc_i, ni_i, amin=517.0)
    np.savez_compressed(os.path.join(OUT, "g4_likelihood.npz"), n=ns, y=ys, lik=lik, sc_i=sc_i, ni_i=ni_i,
                        l_i=l_i, r_i=r_i, sc_f=sc_f, ni_f=ni_f, l_f=l_f, r_f=r_f, l_a=l_a, r_a=r_a)
    print("g4: %d likelihood points" % len(ns))


# ----------------------------------------------------------------------------- G5 / G7
def g5_cross():
    def _append(self, other, ignore_index=False):
        return pd.concat([self, other], ignore_index=ignore_index)
    pd.DataFrame.append = _append             # pandas >= 2 (see module docstring)

    toy = build_toy(1003, n_acc=30, near=((17, 5, 0.5),))
    np.savez_compressed(os.path.join(OUT, "toy_db_cross.npz"), **toy)
    res = {"note": "pandas2-append-shim"}
    arrays = {}
    with tempfile.TemporaryDirectory() as tmp:
        for skip in (False, True):
            g = make_genotype(toy["snps"].copy(), toy["accs"], toy["positions"], toy["chrs"], toy["regions"])
            inputs = make_inputs(toy["s_chrs"], toy["s_pos"], toy["s_gt"], toy["s_wei"], toy["s_dp"])
            outp = os.path.join(tmp, "cross%d" % int(skip))
            stderr = sys.stderr
            sys.stderr = io.StringIO()
            try:
                ci = ref_cs.CrossIdentifier(inputs, g, "athaliana_tair10", 300000, outp, run_identifier=True,
                                            skip_db_hets=skip)
            finally:
                sys.stderr = stderr
            files = {}
            for suf in (".windowscore.txt", ".scores.txt", ".scores.txt.matches.json", ".matches.json"):
                if os.path.exists(outp + suf):
                    files[suf] = read_text(outp + suf)
            res["cross_skip%d" % int(skip)] = files
            # per-window raw kernel outputs, re-driving the reference functions window by window
            genome = ref_genomes.Genome("athaliana_tair10")
            wins_g = list(genome.get_bins_genome(g.g, 300000))
            wins_s = list(genome.get_bins_arrays(inputs.chrs, inputs.pos, 300000))
            assert len(wins_g) == len(wins_s)
            rows_db, rows_s, off = [], [], [0]
            w_score, w_ninfo = [], []
            for e_g, e_s in zip(wins_g, wins_s):
                gp = g.g.positions[e_g[2]]
                sp = inputs.pos[e_s[2]]
                mdb = np.array(e_g[2], dtype=int)[np.where(np.in1d(gp, sp))[0]]
                msm = np.array(e_s[2], dtype=int)[np.where(np.in1d(sp, gp))[0]]
                rows_db.extend(mdb.tolist())
                rows_s.extend(msm.tolist())
                off.append(len(rows_db))
                if len(mdb) > 0:
                    s, ni = ref_sm.matchGTsAccs(inputs.wei[msm, ], toy["snps"][mdb, :].copy(), skip)
                else:
                    s, ni = np.zeros(toy["snps"].shape[1]), np.zeros(toy["snps"].shape[1], dtype=int)
                w_score.append(np.asarray(s, dtype=float))
                w_ninfo.append(np.asarray(ni, dtype=np.int64))
            arrays["win_chr_skip%d" % int(skip)] = np.array([e[0] for e in wins_g], dtype=np.int64)
            arrays["win_off_skip%d" % int(skip)] = np.array(off, dtype=np.int64)
            arrays["win_rows_db_skip%d" % int(skip)] = np.array(rows_db, dtype=np.int64)
            arrays["win_rows_sample_skip%d" % int(skip)] = np.array(rows_s, dtype=np.int64)
            arrays["win_score_skip%d" % int(skip)] = np.array(w_score)
            arrays["win_ninfo_skip%d" % int(skip)] = np.array(w_ninfo)
        # np_test_identity grid (scipy binom.sf)
        rng = np.random.default_rng(5)
        n = rng.integers(1, 800, size=400)
        x = np.floor(n * (1 - rng.random(400) * 0.1))
        xf = x + rng.random(400) * 0.0    # integers stored as float, as window scores with hard weights
        arrays["ident_n"] = n.astype(np.int64)
        arrays["ident_x"] = xf
        arrays["ident_out"] = ref_sm.np_test_identity(xf, n, error_rate=0.02).astype(np.int64)
        xfrac = np.minimum(x + rng.random(400), n)       # fractional scores (PL weights)
        arrays["ident_xfrac"] = xfrac
        arrays["ident_out_frac"] = ref_sm.np_test_identity(xfrac, n, error_rate=0.02).astype(np.int64)
        from scipy import stats
        arrays["ident_sf_frac"] = stats.binom.sf(n - xfrac - 1, n, 0.02)
        arrays["ident_sf"] = stats.binom.sf(n - xf - 1, n, 0.02)
    np.savez_compressed(os.path.join(OUT, "g5_cross.npz"), **arrays)
    with open(os.path.join(OUT, "g5_cross.json"), "w") as fh:
        json.dump(res, fh, indent=1, sort_keys=True)
    print("g5: cross captured (%d windows)" % (len(arrays["win_off_skip0"]) - 1))


# ----------------------------------------------------------------------------- G1b / G2b / G5b: ONE accession
def g1b_single_accession():
    """Panels of one accession: numpy reduces the reference's [1, n] product along a contiguous axis, i.e. pairwise
    inside 8192-element buffer pieces instead of row after row (core/snpmatch.py:85-87) -- its own fixtures, with their
    own seeds, so that g1 ... g7 keep their bytes."""
    rng = np.random.default_rng(701502)
    out = {}
    names = []
    for n in (8, 9, 64, 129, 500, 1000, 2000, 8193, 20000):
        db = rng.choice(DB_V, size=(n, 1), p=DB_P)
        for kind in (("pl", "hard") if n <= 2000 else ("pl",)):
            wei, _ = make_weights(rng, n, frac_pl=0.8 if kind == "pl" else 0.0)
            key0 = "n%d_a1_%s" % (n, kind)
            out[key0 + "_db"] = db
            out[key0 + "_wei"] = wei
            for skip in (False, True):
                s, ni = ref_sm.matchGTsAccs(wei, db.copy(), skip)
                key = key0 + "_%d" % int(skip)
                names.append(key)
                out[key + "_score"] = np.asarray(s, dtype=np.float64)
                out[key + "_ninfo"] = np.asarray(ni, dtype=np.int64)
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "g1b_single_acc.npz"), **out)
    print("g1b: %d one-accession cases" % len(names))

    def _append(self, other, ignore_index=False):
        return pd.concat([self, other], ignore_index=ignore_index)
    pd.DataFrame.append = _append             # pandas >= 2 (see module docstring)
    toy = build_toy(1004, n_acc=1, planted=0)
    np.savez_compressed(os.path.join(OUT, "toy_db_single.npz"), **toy)
    res = {"note": "pandas2-append-shim"}
    arrays = {}
    with tempfile.TemporaryDirectory() as tmp:
        for skip in (False, True):
            # G2b: Genotyper end to end + the raw accumulators of its chunk loop
            gt, files = run_inbred(toy, skip, False, tmp)
            res["inbred_skip%d" % int(skip)] = files
            score = np.zeros(1)
            ninfo = np.zeros(1, dtype="uint32")
            c0, c1 = gt.commonSNPs
            for j in range(0, len(c0), 1000):
                s, ni = ref_sm.matchGTsAccs(toy["s_wei"][c1[j:j + 1000]], toy["snps"][c0[j:j + 1000], :].copy(), skip)
                score = score + s
                ninfo = ninfo + ni
            arrays["common_db"] = np.asarray(c0, dtype=np.int64)
            arrays["common_sample"] = np.asarray(c1, dtype=np.int64)
            arrays["score_skip%d" % int(skip)] = score
            arrays["ninfo_skip%d" % int(skip)] = np.asarray(ninfo, dtype=np.int64)
            assert np.array_equal(np.array(score, dtype=int), gt.result.scores)
            # G5b: the cross pipeline's files and its per-window kernel outputs
            g = make_genotype(toy["snps"].copy(), toy["accs"], toy["positions"], toy["chrs"], toy["regions"])
            inputs = make_inputs(toy["s_chrs"], toy["s_pos"], toy["s_gt"], toy["s_wei"], toy["s_dp"])
            outp = os.path.join(tmp, "cross%d" % int(skip))
            stderr = sys.stderr
            sys.stderr = io.StringIO()
            try:
                ref_cs.CrossIdentifier(inputs, g, "athaliana_tair10", 300000, outp, run_identifier=True, skip_db_hets=skip)
            finally:
                sys.stderr = stderr
            files = {}
            for suf in (".windowscore.txt", ".scores.txt", ".scores.txt.matches.json", ".matches.json"):
                if os.path.exists(outp + suf):
                    files[suf] = read_text(outp + suf)
            res["cross_skip%d" % int(skip)] = files
            genome = ref_genomes.Genome("athaliana_tair10")
            wins_g = list(genome.get_bins_genome(g.g, 300000))
            wins_s = list(genome.get_bins_arrays(inputs.chrs, inputs.pos, 300000))
            rows_db, rows_s, off, w_score, w_ninfo = [], [], [0], [], []
            for e_g, e_s in zip(wins_g, wins_s):
                gp = g.g.positions[e_g[2]]
                sp = inputs.pos[e_s[2]]
                mdb = np.array(e_g[2], dtype=int)[np.where(np.in1d(gp, sp))[0]]
                msm = np.array(e_s[2], dtype=int)[np.where(np.in1d(sp, gp))[0]]
                rows_db.extend(mdb.tolist())
                rows_s.extend(msm.tolist())
                off.append(len(rows_db))
                if len(mdb) > 0:
                    s, ni = ref_sm.matchGTsAccs(inputs.wei[msm, ], toy["snps"][mdb, :].copy(), skip)
                else:
                    s, ni = np.zeros(1), np.zeros(1, dtype=int)
                w_score.append(np.asarray(s, dtype=float))
                w_ninfo.append(np.asarray(ni, dtype=np.int64))
            arrays["win_off_skip%d" % int(skip)] = np.array(off, dtype=np.int64)
            arrays["win_rows_db_skip%d" % int(skip)] = np.array(rows_db, dtype=np.int64)
            arrays["win_rows_sample_skip%d" % int(skip)] = np.array(rows_s, dtype=np.int64)
            arrays["win_score_skip%d" % int(skip)] = np.array(w_score)
            arrays["win_ninfo_skip%d" % int(skip)] = np.array(w_ninfo)
        # one long window (a cross with a large --binLen): more rows than one 8192-element buffer piece
        rng = np.random.default_rng(701503)
        n = 3 * 8192 + 77
        db = rng.choice(DB_V, size=(n, 1), p=DB_P)
        wei, _ = make_weights(rng, n)
        arrays["long_db"], arrays["long_wei"] = db, wei
        for skip in (False, True):
            s, ni = ref_sm.matchGTsAccs(wei, db.copy(), skip)
            arrays["long_score_skip%d" % int(skip)] = np.asarray(s, dtype=np.float64)
            arrays["long_ninfo_skip%d" % int(skip)] = np.asarray(ni, dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, "g2b_g5b_single_acc.npz"), **arrays)
    with open(os.path.join(OUT, "g2b_g5b_single_acc.json"), "w") as fh:
        json.dump(res, fh, indent=1, sort_keys=True)
    print("g2b/g5b: one-accession inbred + cross captured (%d windows)" % (len(arrays["win_off_skip0"]) - 1))


# ----------------------------------------------------------------------------- G6
def g6_common():
    rng = np.random.default_rng(6)
    out = {}
    names = []

    def case(name, c1, p1, c2, p2):
        a, b = ref_sg.Genotype.get_common_positions(np.array(c1), np.array(p1), np.array(c2), np.array(p2))
        out[name + "_c1"] = np.array(c1)
        out[name + "_p1"] = np.array(p1, dtype=np.int64)
        out[name + "_c2"] = np.array(c2)
        out[name + "_p2"] = np.array(p2, dtype=np.int64)
        out[name + "_i1"] = np.asarray(a, dtype=np.int64)
        out[name + "_i2"] = np.asarray(b, dtype=np.int64)
        names.append(name)

    case("basic", ["1"] * 4 + ["2"] * 3, [5, 9, 12, 40, 3, 9, 77],
         ["Chr1"] * 3 + ["Chr2"] * 2, [9, 40, 41, 3, 78])
    case("prefix_case", ["chr1", "chr1", "CHR2"], [1, 2, 3], ["Chr1", "2", "2"], [2, 3, 4])
    case("db_order", ["2", "2", "1", "1"], [10, 20, 10, 30], ["1", "1", "2"], [10, 30, 20])
    case("no_overlap", ["1", "1"], [1, 2], ["3", "3"], [1, 2])
    case("extra_chr_in_sample", ["1", "1", "2"], [1, 5, 9], ["1", "M", "2", "Pt"], [5, 5, 9, 1])
    # larger random, sorted unique per chromosome
    c1, p1, c2, p2 = [], [], [], []
    for ch in ("1", "2", "3", "4", "5"):
        a = np.sort(rng.choice(np.arange(1, 5000), size=600, replace=False))
        b = np.sort(rng.choice(np.arange(1, 5000), size=300, replace=False))
        c1 += [ch] * len(a)
        p1 += a.tolist()
        c2 += ["Chr" + ch] * len(b)
        p2 += b.tolist()
    case("random", c1, p1, c2, p2)
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "g6_common.npz"), **out)
    print("g6: %d cases" % len(names))


# ----------------------------------------------------------------------------- G7: cross interpreter cases
def cross_sample(rng, toy, kind):
    """samples that drive cross_interpreter (core/csmatch.py:131-186) into its F1 / F2 branches"""
    snps, positions, regions = toy["snps"], toy["positions"], toy["regions"]
    n_db = len(positions)
    hit = np.sort(rng.choice(n_db, size=4000, replace=False))
    chr_of_row = np.zeros(n_db, dtype=int)
    for i, (a, b) in enumerate(regions):
        chr_of_row[a:b] = i
    p1, p2 = 3, 9
    codes = np.zeros(len(hit), dtype=np.int8)
    for i, r in enumerate(hit):
        a, b = snps[r, p1], snps[r, p2]
        if kind == "f1":
            if a >= 0 and b >= 0 and a != b:
                c = 2
            elif a >= 0:
                c = a
            else:
                c = b if b >= 0 else 0
        elif kind == "f2hom":   # homozygous mosaic: 3 Mb blocks alternate parent1 / parent2 (no het blocks)
            blk = (int(positions[r]) // 3000000) % 2
            src = a if blk == 0 else b
            c = src if src >= 0 else 0
        else:   # f2: 3 Mb blocks alternate parent1 / het / parent2
            blk = (int(positions[r]) // 3000000 + chr_of_row[r]) % 3
            if blk == 0:
                c = a if a >= 0 else 0
            elif blk == 1:
                c = b if b >= 0 else 0
            else:
                c = 2 if (a >= 0 and b >= 0 and a != b) else (a if a >= 0 else (b if b >= 0 else 0))
        if c == 2 and rng.random() < 0.0:
            c = 0
        codes[i] = c
    wei, _ = make_weights(rng, len(hit), codes=codes)
    gt_of = {0: "0/0", 1: "1/1", 2: "0/1"}
    chrs = np.array(["Chr%d" % (chr_of_row[r] + 1) for r in hit])
    pos = positions[hit].astype(int)
    gt = np.array([gt_of[int(k)] for k in codes])
    dp = rng.integers(1, 40, size=len(hit))
    return chrs, pos, gt, wei, dp


def g7_cross_cases():
    def _append(self, other, ignore_index=False):
        return pd.concat([self, other], ignore_index=ignore_index)
    pd.DataFrame.append = _append
    toy = dict(np.load(os.path.join(OUT, "toy_db_cross.npz")))
    rng = np.random.default_rng(77)
    res = {"note": "pandas2-append-shim"}
    samples = {}
    with tempfile.TemporaryDirectory() as tmp:
        for kind in ("f1", "f2", "f2hom"):
            chrs, pos, gt, wei, dp = cross_sample(rng, toy, kind)
            samples[kind + "_chrs"], samples[kind + "_pos"], samples[kind + "_gt"] = chrs, pos, gt
            samples[kind + "_wei"], samples[kind + "_dp"] = wei, dp
            g = make_genotype(toy["snps"].copy(), toy["accs"], toy["positions"], toy["chrs"], toy["regions"])
            inputs = make_inputs(chrs, pos, gt, wei, dp)
            outp = os.path.join(tmp, "cross_" + kind)
            stderr = sys.stderr
            sys.stderr = io.StringIO()
            try:
                ref_cs.CrossIdentifier(inputs, g, "athaliana_tair10", 300000, outp, run_identifier=True)
            finally:
                sys.stderr = stderr
            files = {}
            for suf in (".windowscore.txt", ".scores.txt", ".scores.txt.matches.json", ".matches.json"):
                if os.path.exists(outp + suf):
                    files[suf] = read_text(outp + suf)
            res[kind] = files
            if ".matches.json" in files:
                print("g7 %s: case %s" % (kind, json.loads(files[".matches.json"])["interpretation"]))
            else:
                print("g7 %s: no cross interpretation (case < 3)" % kind)
    np.savez_compressed(os.path.join(OUT, "g7_cross_samples.npz"), **samples)
    with open(os.path.join(OUT, "g7_cross_cases.json"), "w") as fh:
        json.dump(res, fh, indent=1, sort_keys=True)


if __name__ == "__main__":
    g1_match()
    g2_g3()
    g4_likelihood()
    g5_cross()
    g6_common()
    g7_cross_cases()
    g1b_single_accession()
    sizes = {f: os.path.getsize(os.path.join(OUT, f)) for f in sorted(os.listdir(OUT)) if f.endswith((".npz", ".json"))}
    print(json.dumps(sizes, indent=1))
