"""
Pins the CPU oracle (oracle/snpmatch_oracle.py and oracle/snpmatch_oracle.c) to
 (a) the reference's own known-answer tests, and
 (b) golden vectors produced by running the unmodified reference (tests/golden/make_golden.py).
fp64 scores are compared BIT FOR BIT.
"""
import json
import os

import numpy as np
import pytest

from oracle import c_oracle
from oracle import snpmatch_oracle as orc


def bits(a):
    return np.asarray(a, dtype=np.float64).view(np.uint64)


# ------------------------------------------------------------------ reference KATs
def test_reference_known_answers():
    # /root/reference/tests/test_inbred.py:22-24
    assert orc.likeli_test(10, 3) == 122.8361221819443
    assert np.isnan(orc.likeli_test(10, 0))
    # v5.0.1 asserts y <= n (core/snpmatch.py:43); the stale test expected nan
    with pytest.raises(AssertionError):
        orc.likeli_test(0, 10)
    # README.md:90-92 example rows: (matches, ninfo) -> likelihood, and LRT against the top hit
    l1 = orc.likeli_test(4987, 4946)
    l2 = orc.likeli_test(5194, 4861)
    l3 = orc.likeli_test(4933, 4368)
    assert abs(l1 - 517.57517) < 1e-3
    assert abs(l2 - 4897.207) < 1e-2
    assert abs(l2 / l1 - 9.4618) < 1e-3
    assert abs(l3 / l1 - 16.7165) < 1e-3
    assert orc.likeli_test(11, 11) == 1                       # README.md:120-122
    lik, bad = c_oracle.likelihood([3.0, 0.0, 4946.0, 11.0, 0.0], [10, 10, 4987, 11, 0])
    assert lik[0] == 122.8361221819443 and np.isnan(lik[1]) and lik[3] == 1.0 and np.isnan(lik[4])
    assert abs(lik[2] - l1) <= 1e-12 * l1 and not bad


# ------------------------------------------------------------------ G1: matchGTsAccs
def test_match_golden_bitexact(golden_dir):
    g = np.load(os.path.join(golden_dir, "g1_match.npz"))
    for name in g["names"]:
        db, wei = g[name + "_db"], g[name + "_wei"]
        skip = name.endswith("_1")
        want_s, want_n = g[name + "_score"], g[name + "_ninfo"]
        s, n = orc.match_gts_accs(wei, db, skip)
        assert np.array_equal(bits(s), bits(want_s)), name
        assert np.array_equal(n, want_n), name
        s, n = c_oracle.match(wei, db, skip)
        assert np.array_equal(bits(s), bits(want_s)), "C " + name
        assert np.array_equal(n, want_n), "C " + name
        s, n = orc.match_gts_accs_graph(wei, db.copy(), skip)
        assert np.array_equal(bits(s), bits(want_s)), "graph " + name
        assert np.array_equal(np.asarray(n), want_n), "graph " + name


# ------------------------------------------------------------------ G1b / G2b / G5b: panels of ONE accession
def test_single_accession_goldens_bitexact(golden_dir):
    """numpy sums the reference's [1, n] product along a contiguous axis: pairwise inside 8192-element pieces, not row
    after row (core/snpmatch.py:85-87 with N_acc == 1) -- both oracles follow that rule for one-column panels"""
    g = np.load(os.path.join(golden_dir, "g1b_single_acc.npz"))
    differs_from_sequential = 0
    for name in g["names"]:
        key0, skip = name[:-2], name.endswith("_1")
        db, wei = g[key0 + "_db"], g[key0 + "_wei"]
        assert db.shape[1] == 1
        want_s, want_n = g[name + "_score"], g[name + "_ninfo"]
        for tag, (s, n) in (("numpy", orc.match_gts_accs(wei, db, skip)), ("C", c_oracle.match(wei, db, skip)),
                            ("graph", orc.match_gts_accs_graph(wei, db.copy(), skip))):
            assert np.array_equal(bits(s), bits(want_s)), (tag, name)
            assert np.array_equal(np.asarray(n), want_n), (tag, name)
        # the row-after-row order (what every wider panel gets) really is another number for some of them
        wide = np.concatenate([db, db], axis=1)
        s2, _ = c_oracle.match(wei, wide, skip)
        differs_from_sequential += int(bits(s2)[0] != bits(want_s)[0])
        assert bits(s2)[0] == bits(s2)[1]
    assert differs_from_sequential >= 8


def test_single_accession_genotyper_and_windows_golden(golden_dir):
    toy = np.load(os.path.join(golden_dir, "toy_db_single.npz"))
    g = np.load(os.path.join(golden_dir, "g2b_g5b_single_acc.npz"))
    assert toy["snps"].shape[1] == 1
    c0, c1 = orc.get_common_positions(
        np.repeat(toy["chrs"], [b - a for a, b in toy["regions"]]), toy["positions"], toy["s_chrs"], toy["s_pos"])
    assert np.array_equal(c0, g["common_db"]) and np.array_equal(c1, g["common_sample"])
    wei = toy["s_wei"][c1]
    for skip in (0, 1):
        s, n = orc.genotyper_scores(wei, toy["snps"][c0], 1000, bool(skip))
        assert np.array_equal(bits(s), bits(g["score_skip%d" % skip])) and np.array_equal(n, g["ninfo_skip%d" % skip])
        s, n = c_oracle.genotyper(toy["snps"], c0, wei, 1000, bool(skip))
        assert np.array_equal(bits(s), bits(g["score_skip%d" % skip])) and np.array_equal(n, g["ninfo_skip%d" % skip])
        off = g["win_off_skip%d" % skip]
        rows_db, rows_s = g["win_rows_db_skip%d" % skip], g["win_rows_sample_skip%d" % skip]
        ws, wn, _, _ = c_oracle.windows(toy["snps"], rows_db, toy["s_wei"][rows_s], off, bool(skip))
        assert np.array_equal(bits(ws), bits(g["win_score_skip%d" % skip]))
        assert np.array_equal(wn, g["win_ninfo_skip%d" % skip])
        ws, wn = orc.window_scores(toy["s_wei"][rows_s], toy["snps"][rows_db], off, bool(skip))[:2]
        assert np.array_equal(bits(ws), bits(g["win_score_skip%d" % skip]))
        assert np.array_equal(wn, g["win_ninfo_skip%d" % skip])
        # one window longer than numpy's 8192-element buffer
        s, n = c_oracle.match(g["long_wei"], g["long_db"], bool(skip))
        assert np.array_equal(bits(s), bits(g["long_score_skip%d" % skip])) and np.array_equal(n, g["long_ninfo_skip%d" % skip])
        s, n = orc.match_gts_accs(g["long_wei"], g["long_db"], bool(skip))
        assert np.array_equal(bits(s), bits(g["long_score_skip%d" % skip]))


def test_match_asserts():
    with pytest.raises(AssertionError, match="same number of positions"):
        orc.match_gts_accs(np.ones((3, 3)), np.zeros((4, 2), dtype=np.int8))
    with pytest.raises(AssertionError, match="shape == n,3"):
        orc.match_gts_accs(np.ones((4, 2)), np.zeros((4, 2), dtype=np.int8))


# ------------------------------------------------------------------ G2: chunk loop
def test_genotyper_golden(golden_dir):
    toy = np.load(os.path.join(golden_dir, "toy_db.npz"))
    g = np.load(os.path.join(golden_dir, "g2_inbred.npz"))
    c0, c1 = orc.get_common_positions(
        np.repeat(toy["chrs"], [b - a for a, b in toy["regions"]]), toy["positions"], toy["s_chrs"], toy["s_pos"])
    assert np.array_equal(c0, g["common_db"]) and np.array_equal(c1, g["common_sample"])
    wei = toy["s_wei"][c1]
    for skip in (0, 1):
        s, n = orc.genotyper_scores(wei, toy["snps"][c0], 1000, bool(skip))
        assert np.array_equal(bits(s), bits(g["score_skip%d" % skip]))
        assert np.array_equal(n, g["ninfo_skip%d" % skip])
        s, n = c_oracle.genotyper(toy["snps"], c0, wei, 1000, bool(skip))
        assert np.array_equal(bits(s), bits(g["score_skip%d" % skip]))
        assert np.array_equal(n, g["ninfo_skip%d" % skip])


def test_scores_table_from_oracle(golden_dir):
    """likelihood / lrt columns of scores.txt recomputed by the oracle match the reference's file."""
    toy = np.load(os.path.join(golden_dir, "toy_db.npz"))
    g = np.load(os.path.join(golden_dir, "g2_inbred.npz"))
    files = json.load(open(os.path.join(golden_dir, "g2_inbred.json")))
    rows = [l.split("\t") for l in files["inbred_skip0"]["scores.txt"].strip().split("\n")]
    matches = np.array(g["score_skip0"], dtype=int)
    ninfo = g["ninfo_skip0"]
    lik, lrt = orc.calculate_likelihoods(matches, ninfo)
    for i, r in enumerate(rows):
        assert r[0] == str(toy["accs"][i])
        assert int(r[1]) == matches[i] and int(r[2]) == ninfo[i]
        assert abs(float(r[4]) - lik[i]) <= 1e-12 * abs(lik[i])
        assert abs(float(r[5]) - lrt[i]) <= 1e-12 * abs(lrt[i])


# ------------------------------------------------------------------ G4: likelihoods
def test_likelihood_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "g4_likelihood.npz"))
    got = np.array([orc.likeli_test(int(n), y if y != int(y) else int(y)) for n, y in zip(g["n"], g["y"])], dtype=float)
    assert np.array_equal(np.isnan(got), np.isnan(g["lik"]))
    ok = ~np.isnan(got)
    assert np.array_equal(bits(got[ok]), bits(g["lik"][ok]))
    lik_c, bad = c_oracle.likelihood(g["y"], g["n"])
    assert not bad
    assert np.array_equal(np.isnan(lik_c), np.isnan(g["lik"]))
    np.testing.assert_allclose(lik_c[ok], g["lik"][ok], rtol=1e-13, atol=0)
    for s, n, l, r, kw in (("sc_i", "ni_i", "l_i", "r_i", {}), ("sc_f", "ni_f", "l_f", "r_f", {}),
                           ("sc_i", "ni_i", "l_a", "r_a", {"amin": 517.0})):
        lik, lrt = orc.calculate_likelihoods(g[s], g[n], **kw)
        np.testing.assert_array_equal(np.isnan(lik), np.isnan(g[l]))
        np.testing.assert_allclose(lik, g[l], rtol=1e-15, equal_nan=True)
        np.testing.assert_allclose(lrt, g[r], rtol=1e-15, equal_nan=True)


# ------------------------------------------------------------------ G5: windows
def test_windows_golden(golden_dir):
    toy = np.load(os.path.join(golden_dir, "toy_db_cross.npz"))
    g = np.load(os.path.join(golden_dir, "g5_cross.npz"))
    genome_ids = np.array(["1", "2", "3", "4", "5"])
    chrlen = [30427671, 19698289, 23459830, 18585056, 26975502]
    wg = orc.genome_windows_db(genome_ids, chrlen, toy["chrs"], toy["regions"], toy["positions"], 300000)
    ws = orc.genome_windows_sample(genome_ids, chrlen, toy["s_chrs"], toy["s_pos"], 300000)
    assert len(wg) == len(ws) == 399
    assert np.array_equal([w[0] for w in wg], g["win_chr_skip0"])
    rows_db, rows_s, off = [], [], [0]
    for e_g, e_s in zip(wg, ws):
        gp = toy["positions"][e_g[2]]
        sp = toy["s_pos"][e_s[2]]
        rows_db += np.array(e_g[2], dtype=int)[np.isin(gp, sp)].tolist()
        rows_s += np.array(e_s[2], dtype=int)[np.isin(sp, gp)].tolist()
        off.append(len(rows_db))
    assert np.array_equal(off, g["win_off_skip0"])
    assert np.array_equal(rows_db, g["win_rows_db_skip0"])
    assert np.array_equal(rows_s, g["win_rows_sample_skip0"])
    rows_db = np.array(rows_db)
    wei = toy["s_wei"][np.array(rows_s)]
    for skip in (0, 1):
        s, n, ts, tn = orc.window_scores(wei, toy["snps"][rows_db], off, bool(skip))
        assert np.array_equal(bits(s), bits(g["win_score_skip%d" % skip]))
        assert np.array_equal(n, g["win_ninfo_skip%d" % skip])
        s2, n2, ts2, tn2 = c_oracle.windows(toy["snps"], rows_db, wei, off, bool(skip))
        assert np.array_equal(bits(s2), bits(g["win_score_skip%d" % skip]))
        assert np.array_equal(n2, g["win_ninfo_skip%d" % skip])
        assert np.array_equal(bits(ts), bits(ts2)) and np.array_equal(tn, tn2)


def test_identity_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "g5_cross.npz"))
    assert np.array_equal(orc.test_identity(g["ident_x"], g["ident_n"]), g["ident_out"])
    assert np.array_equal(orc.test_identity(g["ident_xfrac"], g["ident_n"]), g["ident_out_frac"])


# ------------------------------------------------------------------ G6: common positions
def test_common_positions_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "g6_common.npz"))
    for name in g["names"]:
        i1, i2 = orc.get_common_positions(g[name + "_c1"], g[name + "_p1"], g[name + "_c2"], g[name + "_p2"])
        assert np.array_equal(i1, g[name + "_i1"]), name
        assert np.array_equal(i2, g[name + "_i2"]), name


# ------------------------------------------------------------------ G7: in-silico F1 rows of the cross table
def test_np_sum_restatement_bitexact():
    """np.sum == pairwise sums of 8192-element pieces added in order: the order the k_f1_* kernels follow"""
    rng = np.random.default_rng(5)
    sizes = list(range(0, 140)) + [255, 256, 257, 1000, 4095, 4096, 4097, 8191, 8192, 8193, 8199, 8200,
                                   9000, 16384, 16385, 20000, 40000, 80000, 100003]
    for n in sizes:
        for _ in range(3):
            a = np.exp(-rng.integers(0, 256, n) / 10.0)
            assert bits(np.sum(a)) == bits(orc.np_sum_restated(a)), n
    w = np.exp(-rng.integers(0, 256, (60000, 3)) / 10.0)
    idx = np.flatnonzero(rng.random(60000) < 0.4)
    assert bits(np.sum(w[idx, 2])) == bits(orc.np_sum_restated(w[idx, 2]))     # the reference's operand shape


@pytest.mark.parametrize("kind", ["f1", "f2", "f2hom"])
def test_insilico_f1_rows_golden(golden_dir, kind):
    """the 45 'AxB' rows the reference appends to <out>.scores.txt (core/csmatch.py:106-129)"""
    toy = np.load(os.path.join(golden_dir, "toy_db_cross.npz"))
    smp = np.load(os.path.join(golden_dir, "g7_cross_samples.npz"))
    with open(os.path.join(golden_dir, "g7_cross_cases.json")) as fh:
        table = json.load(fh)[kind][".scores.txt"]
    rows = [ln.split("\t") for ln in table.splitlines()]
    crosses = [r for r in rows if "x" in r[0]]
    assert len(crosses) == 45
    accs = [str(a) for a in toy["accs"]]
    order = []
    for r in crosses:                                   # combination order -> the ten selected accessions
        for name in r[0].split("x"):
            if accs.index(name) not in order:
                order.append(accs.index(name))
    db_chr = np.repeat(toy["chrs"], [b - a for a, b in toy["regions"]])
    db_rows, smp_rows = orc.get_common_positions(db_chr, toy["positions"], smp[kind + "_chrs"], smp[kind + "_pos"])
    score, ninfo = orc.insilico_f1_pairs(toy["snps"][db_rows][:, order], smp[kind + "_wei"][smp_rows])
    for k, r in enumerate(crosses):
        assert float(r[1]) == score[k] and int(float(r[2])) == ninfo[k], r[0]
