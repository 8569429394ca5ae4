"""
End-to-end drop-in tests (-m gpu): the host-side mirror classes (Genotyper, GenotyperOutput,
CrossIdentifier) and the CLI, driven like the reference, against the files the UNMODIFIED reference
wrote for the same inputs (tests/golden/g2_inbred.json, g3_refine.json, g5_cross.json).
Integer columns, probabilities and JSON must be identical; likelihood columns within 1e-12 relative
(north_star allows 1e-6).
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from snpmatch_amd.core import csmatch, parsers, snp_genotype, snpmatch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RTOL = 1e-12


def make_inputs(toy):
    inp = parsers.ParseInputs("")
    inp.load_snp_info(toy["s_chrs"], toy["s_pos"], toy["s_gt"], toy["s_wei"], toy["s_dp"])
    return inp


def make_g(toy):
    return snp_genotype.Genotype.from_arrays(toy["snps"], toy["accs"], toy["positions"], toy["chrs"], toy["regions"])


def cmp_scores_table(got_text, want_text, float_cols=(4, 5)):
    got = [l.split("\t") for l in got_text.strip().split("\n")]
    want = [l.split("\t") for l in want_text.strip().split("\n")]
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert len(g) == len(w)
        for c in range(len(w)):
            if c in float_cols:
                a, b = float(g[c]), float(w[c])
                assert (np.isnan(a) and np.isnan(b)) or abs(a - b) <= RTOL * abs(b), (g, w)
            else:
                assert g[c] == w[c], (c, g, w)


def test_inbred_end_to_end_matches_reference_files(golden_dir, tmp_path):
    toy = np.load(os.path.join(golden_dir, "toy_db.npz"))
    gold = json.load(open(os.path.join(golden_dir, "g2_inbred.json")))
    for skip in (False, True):
        out = str(tmp_path / ("inbred%d" % skip))
        gt = snpmatch.Genotyper(make_inputs(toy), make_g(toy), out, run_genotyper=True, skip_db_hets=skip)
        want = gold["inbred_skip%d" % int(skip)]
        cmp_scores_table(open(out + ".scores.txt").read(), want["scores.txt"])
        assert json.load(open(out + ".matches.json")) == json.loads(want["matches.json"])
        assert open(out + ".matches.json").read() == want["matches.json"]          # byte-identical JSON
        assert gt.result.scores.dtype.kind == "i" and len(gt.commonSNPs[0]) == 2400


def test_refine_matches_reference_files(golden_dir, tmp_path):
    toy = np.load(os.path.join(golden_dir, "toy_db_refine.npz"))
    gold = json.load(open(os.path.join(golden_dir, "g3_refine.json")))
    out = str(tmp_path / "refine")
    gt = snpmatch.Genotyper(make_inputs(toy), make_g(toy), out, run_genotyper=False)
    gt.filter_tophits()
    assert hasattr(gt, "result_fine") == gold["has_result_fine"]
    cmp_scores_table(open(out + ".scores.txt").read(), gold["scores.txt"])
    cmp_scores_table(open(out + ".refined.scores.txt").read(), gold["refined.scores.txt"])
    assert open(out + ".matches.json").read() == gold["matches.json"]
    if hasattr(gt, "result_fine"):
        # filter_tophits restricts the second pass through a flag per DB row; the reference's interface (an index array through
        # genotyper(filter_pos_ix=...), core/snpmatch.py:202-205) selects the same rows
        top = np.flatnonzero(gt.result.lrts < snpmatch.lr_thres)
        others = np.flatnonzero(gt.result.lrts >= snpmatch.lr_thres)
        seg = gt.g.identify_segregating_snps(top)
        assert np.array_equal(seg, np.flatnonzero(gt.g.segregating_mask(top)))
        again = snpmatch.Genotyper(make_inputs(toy), make_g(toy), out + "_again", run_genotyper=False)
        fine = again.genotyper(filter_pos_ix=seg, mask_acc_ix=others)
        assert np.array_equal(fine.scores, gt.result_fine.scores) and np.array_equal(fine.ninfo, gt.result_fine.ninfo)
        assert fine.num_snps == gt.result_fine.num_snps and np.array_equal(fine.accs, gt.result_fine.accs)


def cmp_window_table(got_text, want_text, strict_scores=None):
    """acc, snps_match, snps_info, identical, num_amb, window_index identical; likelihood within RTOL; the "score" column
    (float window score / informative sites) identical as text in the reference-order mode (the default),
    within RTOL in the certified fast mode (SNPMATCH_CROSS_FAST=1)"""
    if strict_scores is None:
        strict_scores = (os.environ.get("SNPMATCH_CROSS_STRICT", "0") not in ("", "0")) or \
                        (os.environ.get("SNPMATCH_CROSS_FAST", "0") in ("", "0"))
    got = [l.split("\t") for l in got_text.strip().split("\n")]
    want = [l.split("\t") for l in want_text.strip().split("\n")]
    assert got[0] == want[0]
    assert len(got) == len(want)
    for g, w in zip(got[1:], want[1:]):
        for c in range(8):
            if c == 4 or (c == 3 and not strict_scores):
                assert abs(float(g[c]) - float(w[c])) <= RTOL * abs(float(w[c])), (g, w)
            else:
                assert g[c] == w[c], (c, g, w)


@pytest.fixture(autouse=True, params=["auto", "0"], ids=["db-packed-by-default", "db-int8"])
def db_format(request, monkeypatch):
    """every product-path test runs on both residency formats: the default (2-bit packed when the DB's codes allow it) and
    SNPMATCH_PACKED=0 (the int8 panel); tests that set the variable themselves override this"""
    if request.param == "auto":
        monkeypatch.delenv("SNPMATCH_PACKED", raising=False)
    else:
        monkeypatch.setenv("SNPMATCH_PACKED", request.param)
    return request.param


@pytest.fixture(params=["certified", "strict"])
def cross_mode(request, monkeypatch):
    """both window modes of ``cross``: the default (reference order, byte-identical window scores) and SNPMATCH_CROSS_FAST=1
    (segmented fast pass + certificate)"""
    monkeypatch.delenv("SNPMATCH_CROSS_STRICT", raising=False)
    if request.param == "strict":
        monkeypatch.delenv("SNPMATCH_CROSS_FAST", raising=False)
    else:
        monkeypatch.setenv("SNPMATCH_CROSS_FAST", "1")
    return request.param


def test_cross_end_to_end_matches_reference_files(golden_dir, tmp_path, cross_mode):
    toy = np.load(os.path.join(golden_dir, "toy_db_cross.npz"))
    gold = json.load(open(os.path.join(golden_dir, "g5_cross.json")))
    for skip in (False, True):
        out = str(tmp_path / ("cross%d" % skip))
        ci = csmatch.CrossIdentifier(make_inputs(toy), make_g(toy), "athaliana_tair10", 300000, out,
                                     run_identifier=True, skip_db_hets=skip)
        want = gold["cross_skip%d" % int(skip)]
        cmp_window_table(open(out + ".windowscore.txt").read(), want[".windowscore.txt"])
        cmp_scores_table(open(out + ".scores.txt").read(), want[".scores.txt"])
        assert open(out + ".scores.txt.matches.json").read() == want[".scores.txt.matches.json"]
        assert os.path.exists(out + ".matches.json") == (".matches.json" in want)
        if ".matches.json" in want:
            assert json.load(open(out + ".matches.json")) == json.loads(want[".matches.json"])
        assert len(ci.result.accs) == 30 + 45


def test_get_window_data_api(golden_dir):
    """CrossIdentifier.get_window_data (static, one window) keeps the reference's semantics."""
    accs = np.array(["a", "b", "c", "d"])
    f = csmatch.CrossIdentifier.get_window_data(7, accs, np.array([11.0, 10.25, 3.5, 11.0]), np.array([11, 11, 11, 11]))
    assert list(f.columns) == ["acc", "snps_match", "snps_info", "score", "likelihood", "identical", "num_amb", "window_index"]
    # likeliTest(11, 10.25) / 1.0 is far above lr_thres: only the two perfect matches are ambiguous
    assert f["acc"].tolist() == ["a", "d"] and f["snps_match"].tolist() == [11, 11]
    assert f["num_amb"].tolist() == [2, 2] and f["window_index"].tolist() == [7, 7]
    assert f["likelihood"].tolist() == ["1.0", "1.0"] and f["score"].tolist() == ["1.0", "1.0"]
    f = csmatch.CrossIdentifier.get_window_data(3, accs, np.array([10.9, 10.25, 3.5, 10.8]), np.array([11, 11, 11, 11]))
    assert f["acc"].tolist() == ["a", "d"] and f["snps_match"].tolist() == [10, 10] and f["identical"].tolist() == [1.0, 1.0]
    # every accession ambiguous -> empty frame (core/csmatch.py:57-60)
    f = csmatch.CrossIdentifier.get_window_data(1, accs, np.array([5.0, 5.0, 5.0, 5.0]), np.array([5, 5, 5, 5]))
    assert len(f) == 0


def test_scalar_api_known_answers():
    # /root/reference/tests/test_inbred.py:22-24
    assert snpmatch.likeliTest(10, 3) == pytest.approx(122.8361221819443, rel=RTOL)
    assert snpmatch.likeliTest(10, 0) is np.nan
    with pytest.raises(AssertionError):
        snpmatch.likeliTest(0, 10)
    assert snpmatch.likeliTest(11, 11) == 1
    s, n = snpmatch.matchGTsAccs(np.array([[1.0, 0.0, 0.0], [0.0, 0.0, 1.0]]), np.array([[0, 1, -1], [1, 1, 2]], dtype=np.int8))
    assert s.tolist() == [2.0, 1.0, 0.0] and n.tolist() == [2, 2, 1]
    assert snpmatch.np_test_identity(np.array([98.0, 50.0]), np.array([100, 100]), error_rate=0.02).tolist() == [1, 0]


def test_cli_inbred_and_cross(golden_dir, tmp_path):
    """`python -m snpmatch_amd inbred|cross` on a native flat panel + .npz sample."""
    toy = np.load(os.path.join(golden_dir, "toy_db.npz"))
    gold = json.load(open(os.path.join(golden_dir, "g2_inbred.json")))
    db = str(tmp_path / "toy.snpm")
    snp_genotype.save_native(db, toy["snps"], toy["accs"], toy["positions"], toy["chrs"], toy["regions"])
    sample = str(tmp_path / "sample.npz")
    np.savez(sample, chr=toy["s_chrs"], pos=toy["s_pos"], gt=toy["s_gt"], wei=toy["s_wei"], dp=toy["s_dp"])
    env = dict(os.environ, PYTHONPATH=ROOT)
    out = str(tmp_path / "cli_inbred")
    r = subprocess.run([sys.executable, "-m", "snpmatch_amd", "inbred", "-i", sample, "-d", db, "-o", out],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    cmp_scores_table(open(out + ".scores.txt").read(), gold["inbred_skip0"]["scores.txt"])
    assert open(out + ".matches.json").read() == gold["inbred_skip0"]["matches.json"]
    out = str(tmp_path / "cli_cross")
    r = subprocess.run([sys.executable, "-m", "snpmatch_amd", "cross", "-i", sample, "-d", db, "-b", "300000", "-o", out],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert os.path.exists(out + ".windowscore.txt") and os.path.exists(out + ".scores.txt")
    # missing input file -> exit code 1 with the reference's message (snpmatch/__init__.py:114-118)
    r = subprocess.run([sys.executable, "-m", "snpmatch_amd", "inbred", "-i", str(tmp_path / "nope.vcf"), "-d", db],
                       env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "input file does not exist" in r.stderr


def test_cli_inbred_batch_writes_what_inbred_writes(golden_dir, tmp_path):
    """`python -m snpmatch_amd inbred-batch`: three samples (a parsed .npz, the sample VCF, the sample BED) against one resident
    DB, two per device call -- every sample's files equal what a single `inbred` run writes for it"""
    import shutil
    from snpmatch_amd.core import parsers
    vcf = str(tmp_path / "s1.vcf.gz")
    shutil.copy(os.path.join(golden_dir, "701_501.filter.vcf.gz"), vcf)
    bed = str(tmp_path / "s2.bed")
    shutil.copy(os.path.join(golden_dir, "701_502.filter.bed"), bed)
    a, b = parsers.ParseInputs(vcf), parsers.ParseInputs(bed)
    a.wait_for_cache()
    b.wait_for_cache()
    # a DB over the union of both samples' positions, 40 random accessions
    chrs = np.concatenate([a.chrs, np.array(["Chr" + c if not str(c).lower().startswith("chr") else c for c in b.chrs])])
    pos = np.concatenate([a.pos, b.pos])
    key = np.unique(np.array([("%s\t%09d" % (str(c).lower().replace("chr", ""), p)) for c, p in zip(chrs, pos)]))
    k_chr = np.array([k.split("\t")[0] for k in key])
    k_pos = np.array([int(k.split("\t")[1]) for k in key])
    names = sorted(set(k_chr.tolist()))
    regions, start = [], 0
    for c in names:
        n = int(np.sum(k_chr == c))
        regions.append((start, start + n))
        start += n
    rng = np.random.default_rng(11)
    snps = rng.choice(np.array([-1, 0, 1, 2], dtype=np.int8), size=(len(key), 40), p=[0.05, 0.6, 0.33, 0.02])
    db = str(tmp_path / "union.snpm")
    snp_genotype.save_native(db, snps, np.array([str(i) for i in range(40)]), k_pos, np.array(names), np.array(regions))
    third = str(tmp_path / "s3.npz")
    np.savez(third, chr=a.chrs[::2], pos=a.pos[::2], gt=a.gt[::2], wei=a.wei[::2], dp=a.dp[::2])
    env = dict(os.environ, PYTHONPATH=ROOT)
    out = str(tmp_path / "batch")
    r = subprocess.run([sys.executable, "-m", "snpmatch_amd", "inbred-batch", "-i", vcf, bed, third, "-d", db, "-o", out, "--batch_size", "2"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    for f, name in ((vcf, "s1"), (bed, "s2"), (third, "s3")):
        single = str(tmp_path / ("single_" + name))
        r = subprocess.run([sys.executable, "-m", "snpmatch_amd", "inbred", "-i", f, "-d", db, "-o", single],
                           env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        assert open("%s.%s.matches.json" % (out, name)).read() == open(single + ".matches.json").read(), name
        cmp_scores_table(open("%s.%s.scores.txt" % (out, name)).read(), open(single + ".scores.txt").read())
    # two inputs that would write the same files are refused
    r = subprocess.run([sys.executable, "-m", "snpmatch_amd", "inbred-batch", "-i", vcf, vcf, "-d", db, "-o", out],
                       env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "distinct" in r.stderr


@pytest.mark.parametrize("kind", ["f1", "f2", "f2hom"])
def test_cross_interpreter_cases_match_reference(golden_dir, tmp_path, kind, cross_mode):
    """F1-like / F2-like samples: the whole cross pipeline incl. cross_interpreter (cases 5 and 6)."""
    toy = np.load(os.path.join(golden_dir, "toy_db_cross.npz"))
    smp = np.load(os.path.join(golden_dir, "g7_cross_samples.npz"))
    gold = json.load(open(os.path.join(golden_dir, "g7_cross_cases.json")))[kind]
    inp = parsers.ParseInputs("")
    inp.load_snp_info(smp[kind + "_chrs"], smp[kind + "_pos"], smp[kind + "_gt"], smp[kind + "_wei"], smp[kind + "_dp"])
    out = str(tmp_path / ("cross_" + kind))
    csmatch.CrossIdentifier(inp, make_g(toy), "athaliana_tair10", 300000, out, run_identifier=True)
    cmp_window_table(open(out + ".windowscore.txt").read(), gold[".windowscore.txt"])
    cmp_scores_table(open(out + ".scores.txt").read(), gold[".scores.txt"])
    assert open(out + ".scores.txt.matches.json").read() == gold[".scores.txt.matches.json"]
    assert ".matches.json" in gold
    got = json.load(open(out + ".matches.json"))
    want = json.loads(gold[".matches.json"])
    assert got == want
    assert got["interpretation"]["case"] == (6 if kind == "f2hom" else 5)


def test_config1_sample_vcf_through_cli(golden_dir, tmp_path):
    """BASELINE configs[0]: `snpmatch inbred` on the reference's sample VCF (701_501.filter.vcf).  The real
    all_chromosomes_binary.hdf5 is not available, so a TAIR10-shaped DB is synthesised as SURVEY.md 8d
    prescribes: positions = the VCF's and the BED's positions plus random fill to 100k SNPs, 64 accessions,
    seed 701501, accession 40 planted from the VCF's hard calls with 1 % flips.  CLI outputs are checked
    against the oracle run on the same parsed sample."""
    import gzip
    import shutil
    from oracle import c_oracle
    from oracle import snpmatch_oracle as orc
    vcf = str(tmp_path / "701_501.filter.vcf")
    with gzip.open(os.path.join(golden_dir, "701_501.filter.vcf.gz"), "rb") as fi, open(vcf, "wb") as fo:
        shutil.copyfileobj(fi, fo)
    smp = parsers.ParseInputs(vcf, logDebug=False)
    bed = np.loadtxt(os.path.join(golden_dir, "701_502.filter.bed"), dtype=str)
    rng = np.random.default_rng(701501)
    chrlen = [30427671, 19698289, 23459830, 18585056, 26975502]
    s_chr = np.array([int(c.replace("Chr", "")) for c in smp.chrs])
    b_chr, b_pos = bed[:, 0].astype(int), bed[:, 1].astype(int)
    pos_by_chr = []
    for c in range(1, 6):
        have = np.union1d(smp.pos[s_chr == c], b_pos[b_chr == c])
        fill = rng.choice(np.arange(1, chrlen[c - 1] + 1), size=20000, replace=False)
        pos_by_chr.append(np.union1d(have, fill))
    positions = np.concatenate(pos_by_chr)
    regions, start = [], 0
    for p in pos_by_chr:
        regions.append((start, start + len(p)))
        start += len(p)
    n_snp, n_acc, planted = len(positions), 64, 40
    snps = rng.choice(np.array([-1, 0, 1, 2], dtype=np.int8), size=(n_snp, n_acc), p=[0.05, 0.60, 0.33, 0.02])
    codes = parsers.parseGT(smp.gt)
    for c in range(1, 6):
        a, b = regions[c - 1]
        ix = a + np.searchsorted(positions[a:b], smp.pos[s_chr == c])
        col = codes[s_chr == c].copy()
        flip = rng.random(len(col)) < 0.01
        col[flip] = rng.choice(np.array([0, 1], dtype=np.int8), size=int(flip.sum()))
        snps[ix, planted] = col
    accs = np.array([str(9000 + i) for i in range(n_acc)])
    db = str(tmp_path / "tair10_like.snpm")
    snp_genotype.save_native(db, snps, accs, positions, ["1", "2", "3", "4", "5"], regions)
    out = str(tmp_path / "config1")
    r = subprocess.run([sys.executable, "-m", "snpmatch_amd", "inbred", "-v", "-i", vcf, "-d", db, "-o", out],
                       env=dict(os.environ, PYTHONPATH=ROOT), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    # oracle on the same inputs
    chrom = np.repeat(np.array(["1", "2", "3", "4", "5"]), [b - a for a, b in regions])
    c0, c1 = orc.get_common_positions(chrom, positions, smp.chrs, smp.pos)
    assert len(c0) == 7545
    want_s, want_n = c_oracle.genotyper(snps, c0, smp.wei[c1], 1000, False)
    lik, lrt = orc.calculate_likelihoods(np.array(want_s, dtype=int), want_n)
    rows = [l.split("\t") for l in open(out + ".scores.txt").read().strip().split("\n")]
    assert [x[0] for x in rows] == accs.tolist()
    assert [int(x[1]) for x in rows] == np.array(want_s, dtype=int).tolist()
    assert [int(x[2]) for x in rows] == want_n.tolist()
    np.testing.assert_allclose([float(x[4]) for x in rows], lik, rtol=RTOL)
    np.testing.assert_allclose([float(x[5]) for x in rows], lrt, rtol=RTOL)
    assert all(x[6] == "7545" for x in rows)
    js = json.load(open(out + ".matches.json"))
    assert js["interpretation"]["case"] == 0 and js["matches"][0][0] == accs[planted]
    assert js["overlap"] == [1.0, 7545] and js["percent_heterozygosity"] == 110 / 7545.0
    assert os.path.exists(vcf + ".snpmatch.npz") and os.path.exists(vcf + ".snpmatch.stats.json")
    # such a run imports neither pandas nor torch (most of a second of a cold start), and it ran on the library
    probe = ("import sys; from snpmatch_amd import cli; rc = cli.main(['inbred', '-i', %r, '-d', %r, '-o', %r]); "
             "print('RC', rc, 'PANDAS', 'pandas' in sys.modules, 'TORCH', 'torch' in sys.modules)" % (vcf, db, out + "_again"))
    r = subprocess.run([sys.executable, "-c", probe], env=dict(os.environ, PYTHONPATH=ROOT), capture_output=True, text=True, timeout=300)
    assert "RC 0 PANDAS False TORCH False" in r.stdout, (r.stdout, r.stderr[-2000:])
    assert open(out + "_again.scores.txt").read() == open(out + ".scores.txt").read()


def test_inbred_on_packed_panel_matches_reference_files(golden_dir, tmp_path, monkeypatch):
    """SNPMATCH_PACKED=1: the same end-to-end outputs from the 2-bit panel"""
    monkeypatch.setenv("SNPMATCH_PACKED", "1")
    toy = np.load(os.path.join(golden_dir, "toy_db.npz"))
    gold = json.load(open(os.path.join(golden_dir, "g2_inbred.json")))
    out = str(tmp_path / "packed")
    g = make_g(toy)
    snpmatch.Genotyper(make_inputs(toy), g, out, run_genotyper=True)
    assert g._panel.packed
    cmp_scores_table(open(out + ".scores.txt").read(), gold["inbred_skip0"]["scores.txt"])
    assert open(out + ".matches.json").read() == gold["inbred_skip0"]["matches.json"]


def test_cross_on_packed_panel_matches_reference_files(golden_dir, tmp_path, monkeypatch, cross_mode):
    """SNPMATCH_PACKED=1: window scores (strict order on the 2-bit panel), identity test, in-silico crosses and
    the interpretation reproduce the reference's cross outputs"""
    monkeypatch.setenv("SNPMATCH_PACKED", "1")
    toy = np.load(os.path.join(golden_dir, "toy_db_cross.npz"))
    gold = json.load(open(os.path.join(golden_dir, "g5_cross.json")))
    for skip in (False, True):
        out = str(tmp_path / ("pcross%d" % skip))
        g = make_g(toy)
        ci = csmatch.CrossIdentifier(make_inputs(toy), g, "athaliana_tair10", 300000, out, run_identifier=True, skip_db_hets=skip)
        assert g._panel.packed
        want = gold["cross_skip%d" % int(skip)]
        cmp_window_table(open(out + ".windowscore.txt").read(), want[".windowscore.txt"])
        cmp_scores_table(open(out + ".scores.txt").read(), want[".scores.txt"])
        assert open(out + ".scores.txt.matches.json").read() == want[".scores.txt.matches.json"]
        assert len(ci.result.accs) == 30 + 45
