"""
Parity tests proper (-m gpu): the HIP path, called through the C ABI (ctypes), against
 (a) the golden vectors generated from the reference, and
 (b) the CPU oracle on the same seeded inputs,
bit-exact for counts / informative sites / strict-order fp64 scores, <= 1e-6 relative for
likelihoods (tolerance from BASELINE.json north_star; we assert 1e-12).
"""
import os

import numpy as np
import pytest

from oracle import c_oracle
from oracle import snpmatch_oracle as orc
from snpmatch_amd import engine, synth

pytestmark = pytest.mark.gpu

LIK_RTOL = 1e-12      # north_star allows 1e-6


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


@pytest.fixture(scope="module")
def ctx():
    c = engine.Context(0)
    yield c
    c.close()


def rand_db(rng, n, n_acc):
    return rng.choice(np.array([-1, 0, 1, 2], dtype=np.int8), size=(n, n_acc), p=[0.05, 0.60, 0.33, 0.02])


def rand_wei(rng, n, frac_pl=0.8):
    codes = rng.choice(np.array([0, 1, 2], dtype=np.int8), size=n, p=[0.6, 0.35, 0.05])
    return synth.sample_weights(rng, codes, frac_pl)


# ------------------------------------------------------------------ matchGTsAccs (a1)
def test_match_golden_bitexact(ctx, golden_dir):
    g = np.load(os.path.join(golden_dir, "g1_match.npz"))
    for name in g["names"]:
        skip = name.endswith("_1")
        s, n = ctx.score_dense(g[name + "_wei"], g[name + "_db"], skip)
        assert np.array_equal(bits(s), bits(g[name + "_score"])), name
        assert np.array_equal(n, g[name + "_ninfo"]), name


def test_match_asserts_like_reference(ctx):
    with pytest.raises(AssertionError, match="same number of positions"):
        ctx.score_dense(np.ones((3, 3)), np.zeros((4, 2), dtype=np.int8))
    with pytest.raises(AssertionError, match="shape == n,3"):
        ctx.score_dense(np.ones((4, 2)), np.zeros((4, 2), dtype=np.int8))


def test_match_does_not_modify_inputs(ctx):
    rng = np.random.default_rng(3)
    db = rand_db(rng, 100, 33)
    wei = rand_wei(rng, 100)
    db0, wei0 = db.copy(), wei.copy()
    ctx.score_dense(wei, db, True)
    assert np.array_equal(db, db0) and np.array_equal(wei, wei0)


# ------------------------------------------------------------------ Genotyper chunk loop (a2)
CASES = [
    # n_snp_panel, n_acc, n_matched (None = dense all rows), chunk
    (4000, 1, 1000, 1000),
    (4000, 3, 2500, 1000),
    (6000, 64, 3001, 1000),
    (20000, 257, 5000, 1000),
    (20000, 1135, 7545, 1000),
    (3000, 1250, None, 1000),
    (1500, 4097, None, 500),
    (2100, 10000, None, 1000),
    (1300, 8192, None, 1000),          # a row pitch that would be a multiple of 8 KiB: padded by 256 B (panel_row_pitch)
    (130, 17, 129, 7),
]


@pytest.mark.parametrize("n_snp,n_acc,n_match,chunk", CASES)
@pytest.mark.parametrize("skip", [False, True])
def test_query_modes_vs_oracle(ctx, n_snp, n_acc, n_match, chunk, skip):
    rng = np.random.default_rng(n_snp * 31 + n_acc)
    db = rand_db(rng, n_snp, n_acc)
    if n_acc > 5:
        db[:, 2] = -1                      # an accession with no informative site at all
        db[:, 3] = 3                       # out-of-range codes: informative, never matching
    panel = engine.Panel.from_host(ctx, db)
    if n_match is None:
        row_idx, n = None, n_snp
    else:
        row_idx = np.sort(rng.choice(n_snp, size=n_match, replace=False)).astype(np.int64)
        n = n_match
    wei = rand_wei(rng, n)
    want_s, want_n = c_oracle.genotyper(db, row_idx, wei, chunk, skip)
    q = engine.Query(panel, row_idx, wei)
    bound = q.error_bound(chunk)
    assert 0 < bound < 1e-6
    # strict: the reference's fp64 bits
    s, ni = q.run(chunk, skip, engine.MODE_STRICT)
    assert np.array_equal(bits(s), bits(want_s))
    assert np.array_equal(ni, want_n)
    # exact (default): counts bit-exact, fp64 within the certified bound
    s, ni, info = q.run(chunk, skip, engine.MODE_EXACT, return_info=True)
    assert np.array_equal(ni, want_n)
    assert np.array_equal(np.array(s, dtype=int), np.array(want_s, dtype=int))
    assert np.max(np.abs(s - want_s)) <= bound
    # fast pass alone stays inside the bound too
    s, ni = q.run(chunk, skip, engine.MODE_FAST)
    assert np.array_equal(ni, want_n)
    assert np.max(np.abs(s - want_s)) <= bound
    q.free()
    panel.free()


def test_hard_weights_are_exact_in_every_mode(ctx):
    """0/1 weights (BED input, VCF without PL): every order is exact -> all modes bit-identical."""
    rng = np.random.default_rng(77)
    db = rand_db(rng, 9000, 300)
    wei = rand_wei(rng, 9000, frac_pl=0.0)
    want_s, want_n = c_oracle.genotyper(db, None, wei, 1000, False)
    panel = engine.Panel.from_host(ctx, db)
    q = engine.Query(panel, None, wei)
    assert q.error_bound(1000) == 0.0
    for mode in (engine.MODE_EXACT, engine.MODE_STRICT, engine.MODE_FAST):
        s, ni, info = q.run(1000, False, mode, return_info=True)
        assert np.array_equal(bits(s), bits(want_s)) and np.array_equal(ni, want_n)
        assert info["n_strict_reeval"] == 0


def test_exact_mode_reevaluates_boundary_cases(ctx):
    """Scores that sit on an integer boundary must be re-evaluated in strict order: a perfect-match
    accession under PL weights sums exact 1.0s, an integer the fast pass cannot certify."""
    rng = np.random.default_rng(5)
    n, n_acc = 5000, 40
    db = rand_db(rng, n, n_acc)
    codes = db[:, 7].copy()
    codes[codes < 0] = 0
    wei = synth.sample_weights(rng, codes, frac_pl=1.0)
    want_s, want_n = c_oracle.genotyper(db, None, wei, 1000, False)
    panel = engine.Panel.from_host(ctx, db)
    q = engine.Query(panel, None, wei)
    s, ni, info = q.run(1000, False, engine.MODE_EXACT, return_info=True)
    assert info["n_strict_reeval"] >= 1
    assert np.array_equal(np.array(s, dtype=int), np.array(want_s, dtype=int))
    assert bits(s)[7] == bits(want_s)[7]


def test_golden_inbred_accumulators(ctx, golden_dir):
    toy = np.load(os.path.join(golden_dir, "toy_db.npz"))
    g = np.load(os.path.join(golden_dir, "g2_inbred.npz"))
    panel = engine.Panel.from_host(ctx, toy["snps"])
    q = engine.Query(panel, g["common_db"], toy["s_wei"][g["common_sample"]])
    for skip in (0, 1):
        s, ni = q.run(1000, bool(skip), engine.MODE_STRICT)
        assert np.array_equal(bits(s), bits(g["score_skip%d" % skip]))
        assert np.array_equal(ni, g["ninfo_skip%d" % skip])
        s, ni = q.run(1000, bool(skip), engine.MODE_EXACT)
        assert np.array_equal(np.array(s, dtype=int), np.array(g["score_skip%d" % skip], dtype=int))
        assert np.array_equal(ni, g["ninfo_skip%d" % skip])


def test_empty_and_ragged(ctx):
    rng = np.random.default_rng(9)
    db = rand_db(rng, 50, 5)
    panel = engine.Panel.from_host(ctx, db)
    q = engine.Query(panel, np.zeros(0, dtype=np.int64), np.zeros((0, 3)))
    for mode in (engine.MODE_EXACT, engine.MODE_STRICT, engine.MODE_FAST):
        s, ni = q.run(1000, False, mode)
        assert np.array_equal(s, np.zeros(5)) and np.array_equal(ni, np.zeros(5, dtype=np.int64))
    with pytest.raises(AssertionError):
        engine.Query(panel, np.array([50], dtype=np.int64), np.ones((1, 3)))       # row outside the panel
    with pytest.raises(AssertionError, match="n_acc too large"):
        engine.Panel(ctx, 1, (1 << 27) + 1)                                        # beyond the 32-bit row-group offsets
    with pytest.raises(AssertionError, match="shape == n,3"):
        engine.Query(panel, np.array([1], dtype=np.int64), np.ones((1, 2)))
    # repeated rows are allowed (the reference would gather them twice as well)
    idx = np.array([3, 3, 4, 10, 10, 10], dtype=np.int64)
    wei = rand_wei(rng, 6)
    want_s, want_n = c_oracle.genotyper(db, idx, wei, 1000, False)
    s, ni = engine.Query(panel, idx, wei).run(1000, False, engine.MODE_STRICT)
    assert np.array_equal(bits(s), bits(want_s)) and np.array_equal(ni, want_n)


# ------------------------------------------------------------------ windows (a7)
def test_windows_golden(ctx, golden_dir):
    toy = np.load(os.path.join(golden_dir, "toy_db_cross.npz"))
    g = np.load(os.path.join(golden_dir, "g5_cross.npz"))
    panel = engine.Panel.from_host(ctx, toy["snps"])
    for skip in (0, 1):
        rows_db = g["win_rows_db_skip%d" % skip]
        wei = toy["s_wei"][g["win_rows_sample_skip%d" % skip]]
        off = g["win_off_skip%d" % skip]
        q = engine.Query(panel, rows_db, wei)
        s, ni, ts, tn = q.run_windows(off, bool(skip))
        assert np.array_equal(bits(s), bits(g["win_score_skip%d" % skip]))
        assert np.array_equal(ni, g["win_ninfo_skip%d" % skip])
        _, _, ots, otn = c_oracle.windows(toy["snps"], rows_db, wei, off, bool(skip))
        assert np.array_equal(bits(ts), bits(ots)) and np.array_equal(tn, otn)


def test_windows_random_vs_oracle(ctx):
    rng = np.random.default_rng(21)
    db = rand_db(rng, 30000, 1135)
    panel = engine.Panel.from_host(ctx, db)
    rows = np.sort(rng.choice(30000, size=12000, replace=False)).astype(np.int64)
    wei = rand_wei(rng, 12000)
    cuts = np.sort(rng.choice(12001, size=398, replace=True))
    off = np.concatenate([[0], cuts, [12000]]).astype(np.int64)        # includes empty windows
    want = c_oracle.windows(db, rows, wei, off, False)
    got = engine.Query(panel, rows, wei).run_windows(off, False)
    assert np.array_equal(bits(got[0]), bits(want[0])) and np.array_equal(got[1], want[1])
    assert np.array_equal(bits(got[2]), bits(want[2])) and np.array_equal(got[3], want[3])


# ------------------------------------------------------------------ likelihood (a3/a4) and identity (a8)
def test_likelihood_golden(ctx, golden_dir):
    g = np.load(os.path.join(golden_dir, "g4_likelihood.npz"))
    # pointwise: one row per point so that the ratio column is irrelevant
    lik, _ = ctx.likelihood(g["y"].reshape(-1, 1), g["n"].reshape(-1, 1))
    lik = lik.ravel()
    assert np.array_equal(np.isnan(lik), np.isnan(g["lik"]))
    ok = ~np.isnan(lik)
    np.testing.assert_allclose(lik[ok], g["lik"][ok], rtol=LIK_RTOL, atol=0)
    assert lik[(g["n"] == 10) & (g["y"] == 3)][0] == pytest.approx(122.8361221819443, rel=LIK_RTOL)
    for s, n, l, r, amin in (("sc_i", "ni_i", "l_i", "r_i", None), ("sc_f", "ni_f", "l_f", "r_f", None),
                             ("sc_i", "ni_i", "l_a", "r_a", 517.0)):
        lik, lrt = ctx.likelihood(g[s], g[n], amin=amin)
        np.testing.assert_allclose(lik, g[l], rtol=LIK_RTOL, equal_nan=True)
        np.testing.assert_allclose(lrt, g[r], rtol=LIK_RTOL, equal_nan=True)


def test_likelihood_truncates_and_asserts(ctx):
    lik, lrt = ctx.likelihood(np.array([4946.9, 4861.2]), np.array([4987, 5194]), truncate=True)
    assert lik[0] == pytest.approx(orc.likeli_test(4987, 4946), rel=LIK_RTOL)
    assert lrt[1] == pytest.approx(orc.likeli_test(5194, 4861) / orc.likeli_test(4987, 4946), rel=LIK_RTOL)
    with pytest.raises(AssertionError, match="greater than n"):
        ctx.likelihood(np.array([11.0]), np.array([10]))
    lik, lrt = ctx.likelihood(np.array([0.0, 0.0]), np.array([5, 0]))      # all-NaN row
    assert np.isnan(lik).all() and np.isnan(lrt).all()


def test_identity_golden(ctx, golden_dir):
    g = np.load(os.path.join(golden_dir, "g5_cross.npz"))
    out, sf = ctx.binom_identity(g["ident_x"], g["ident_n"], 0.02, return_sf=True)
    np.testing.assert_allclose(sf, g["ident_sf"], rtol=1e-9, atol=1e-300)
    assert np.array_equal(out, g["ident_out"])
    out, sf = ctx.binom_identity(g["ident_xfrac"], g["ident_n"], 0.02, return_sf=True)
    np.testing.assert_allclose(sf, g["ident_sf_frac"], rtol=1e-9, atol=1e-300)
    assert np.array_equal(out, g["ident_out_frac"])


# ------------------------------------------------------------------ panel plumbing
def test_synthetic_fill_matches_numpy_twin(ctx):
    panel = engine.Panel(ctx, 777, 1135)
    panel.fill_synthetic(1001, snp0=5000, acc0=8)
    got = panel.download_rows(0, 777)
    want = synth.panel_values(1001, 5000, 777, 8, 1135)
    assert np.array_equal(got, want)


def test_upload_roundtrip_and_canonical_codes(ctx):
    rng = np.random.default_rng(4)
    db = rng.integers(-128, 128, size=(300, 77)).astype(np.int8)
    panel = engine.Panel.from_host(ctx, db)
    back = panel.download_rows(0, 300)
    want = np.where(db < 0, -1, np.where(db > 2, 3, db)).astype(np.int8)
    assert np.array_equal(back, want)
    wei = rand_wei(rng, 300)
    s, ni = engine.Query(panel, None, wei).run(1000, False, engine.MODE_STRICT)
    ws, wn = c_oracle.genotyper(db, None, wei, 1000, False)
    assert np.array_equal(bits(s), bits(ws)) and np.array_equal(ni, wn)


@pytest.mark.parametrize("skip", [False, True])
def test_reference_order_kernel_takes_both_forms(ctx, skip):
    """k_strict4 scores panels whose calls are all in {-1, 0, 1, 2} with two compares per call and falls back to three
    when an upload stored a call code > 2 (informative, matches no class: core/snpmatch.py:78-88 compares with 0, 1, 2
    only).  The same panel first without, then with such calls -- the flag is raised by the second upload and stays."""
    rng = np.random.default_rng(11)
    n, n_acc = 5003, 1300
    db = rng.choice(np.array([-1, 0, 1, 2], dtype=np.int8), size=(n, n_acc), p=[0.05, 0.6, 0.3, 0.05])
    wei = rand_wei(rng, n)
    panel = engine.Panel.from_host(ctx, db)
    q = engine.Query(panel, None, wei)
    s, ni = q.run(1000, skip, engine.MODE_STRICT)
    ws, wn = c_oracle.genotyper(db, None, wei, 1000, skip)
    assert np.array_equal(bits(s), bits(ws)) and np.array_equal(ni, wn)
    db[17:4000:7, 3::5] = 3
    db[4001, :] = 127
    panel.upload_rows(0, db)
    s, ni = q.run(1000, skip, engine.MODE_STRICT)
    ws, wn = c_oracle.genotyper(db, None, wei, 1000, skip)
    assert np.array_equal(bits(s), bits(ws)) and np.array_equal(ni, wn)
    se, ne = q.run(1000, skip, engine.MODE_EXACT)
    assert np.array_equal(ne, wn) and np.array_equal(np.array(se, dtype=np.int64), np.array(ws, dtype=np.int64))
    # overwriting the rows with clean calls keeps the (conservative) flag: results stay the reference's
    db[db > 2] = 0
    panel.upload_rows(0, db)
    s, ni = q.run(1000, skip, engine.MODE_STRICT)
    ws, wn = c_oracle.genotyper(db, None, wei, 1000, skip)
    assert np.array_equal(bits(s), bits(ws)) and np.array_equal(ni, wn)


# ------------------------------------------------------------------ size-independent properties at scale
def test_large_panel_properties(ctx):
    """10k accessions x 1M SNPs (10 GB) generated on the device: every informative element matches
    exactly one category under all-ones weights (score == ninfo), ninfo agrees with the numpy twin on
    sampled columns, a planted sample matches its accession perfectly, and the score is additive over
    row ranges for integer weights."""
    n_snp, n_acc = 1_000_000, 10_000
    panel = engine.Panel(ctx, n_snp, n_acc)
    panel.fill_synthetic(10050)
    ones = np.ones((n_snp, 3))
    q = engine.Query(panel, None, ones)
    s, ni, info = q.run(1000, False, engine.MODE_EXACT, return_info=True)
    assert info["all_integer_weights"] and info["n_strict_reeval"] == 0
    assert np.array_equal(s, ni.astype(np.float64))
    cols = [0, 1, 4999, 9996]                       # whole quads around these columns from the twin
    for c in cols:
        c4 = (c // 4) * 4
        tw = synth.panel_values(10050, 0, n_snp, c4, 4)
        assert np.array_equal(ni[c4:c4 + 4], n_snp - (tw < 0).sum(axis=0))
    s_skip, ni_skip = q.run(1000, True, engine.MODE_EXACT)
    tw = synth.panel_values(10050, 0, n_snp, 416, 4)
    assert np.array_equal(ni_skip[416:420], n_snp - ((tw < 0) | (tw == 2)).sum(axis=0))
    q.free()
    # planted accession 417, hard weights: perfect match, nobody else above it
    col = tw[:, 1].copy()
    codes = col.copy()
    codes[codes < 0] = 0
    wei = orc.weights_from_gt_codes(codes)
    q = engine.Query(panel, None, wei)
    s, ni = q.run(1000, False, engine.MODE_EXACT)
    assert s[417] == ni[417] and np.argmax(s / np.maximum(ni, 1)) == 417
    # additivity over row ranges (integer weights: exact)
    h = n_snp // 2 + 77
    s1, n1 = engine.Query(panel, None, wei[:h], row0=0).run(1000, False, engine.MODE_EXACT)
    s2, n2 = engine.Query(panel, None, wei[h:], row0=h).run(1000, False, engine.MODE_EXACT)
    assert np.array_equal(s1 + s2, s) and np.array_equal(n1 + n2, ni)


def test_medium_dense_vs_oracle_counts(ctx):
    """100k SNPs x 10k accessions against the C oracle: counts bit-exact in the default mode."""
    n_snp, n_acc = 100_000, 10_000
    panel = engine.Panel(ctx, n_snp, n_acc)
    panel.fill_synthetic(424242)
    db = panel.download_rows(0, n_snp)
    assert np.array_equal(db[:1000], synth.panel_values(424242, 0, 1000, 0, n_acc))
    rng = np.random.default_rng(8)
    codes, wei = synth.planted_sample(rng, db[:, 4321], err=0.02)
    want_s, want_n = c_oracle.genotyper(db, None, wei, 1000, False)
    q = engine.Query(panel, None, wei)
    s, ni, info = q.run(1000, False, engine.MODE_EXACT, return_info=True)
    assert np.array_equal(ni, want_n)
    assert np.array_equal(np.array(s, dtype=int), np.array(want_s, dtype=int))
    assert np.max(np.abs(s - want_s)) <= q.error_bound(1000)
    lik, lrt = ctx.likelihood(s, ni, truncate=True)
    wl, wr = orc.calculate_likelihoods(np.array(want_s, dtype=int), want_n)
    np.testing.assert_allclose(lik, wl, rtol=LIK_RTOL, equal_nan=True)
    np.testing.assert_allclose(lrt, wr, rtol=LIK_RTOL, equal_nan=True)
    assert np.nanargmin(lik) == 4321


# ------------------------------------------------------------------ BASELINE.json configs[1] / configs[2] at full shape
def test_config2_and_config3_full_shape(ctx):
    """1001-Genomes-shaped panel (1135 accessions x 11M SNPs, 12.7 GB resident), 200k-SNP sample:
    inbred (row-gather mode, config 2) and 300 kb windows (399 windows, config 3) against the C oracle
    run on the same 200k rows regenerated by the numpy twin of the device generator."""
    from snpmatch_amd.core import genomes
    n_snp, n_acc, n_match = 11_000_000, 1135, 200_000
    panel = engine.Panel(ctx, n_snp, n_acc)
    panel.fill_synthetic(1001)
    rng = np.random.default_rng(1001)
    rows = np.sort(rng.choice(n_snp, size=n_match, replace=False)).astype(np.int64)
    db = synth.panel_rows(1001, rows, 0, n_acc)
    assert np.array_equal(db[:500], np.vstack([panel.download_rows(int(r), 1) for r in rows[:500]]))
    codes, wei = synth.planted_sample(rng, db[:, 417], err=0.02)
    want_s, want_n = c_oracle.genotyper(db, None, wei, 1000, False)
    q = engine.Query(panel, rows, wei)
    s, ni, info = q.run(1000, False, engine.MODE_EXACT, return_info=True)
    assert np.array_equal(ni, want_n)
    assert np.array_equal(np.array(s, dtype=int), np.array(want_s, dtype=int))
    s2, ni2 = q.run(1000, False, engine.MODE_STRICT)
    assert np.array_equal(bits(s2), bits(want_s)) and np.array_equal(ni2, want_n)
    lik, lrt = ctx.likelihood(s, ni, truncate=True)
    wl, wr = orc.calculate_likelihoods(np.array(want_s, dtype=int), want_n)
    np.testing.assert_allclose(lik, wl, rtol=LIK_RTOL, equal_nan=True)
    np.testing.assert_allclose(lrt, wr, rtol=LIK_RTOL, equal_nan=True)
    assert int(np.nanargmin(lik)) == 417 and (lrt < 3.841).sum() == 1
    # config 3: TAIR10 chromosomes proportional to their lengths, 300 kb windows over the matched rows
    g = genomes.Genome("athaliana_tair10")
    frac = np.cumsum(g.chrlen) / g.chrlen.sum()
    bounds = np.concatenate([[0], np.round(frac * n_snp).astype(np.int64)])
    off = [0]
    for c in range(5):
        in_chr = rows[(rows >= bounds[c]) & (rows < bounds[c + 1])]
        pos = 1 + (in_chr - bounds[c]) * int(g.chrlen[c] - 1) // int(bounds[c + 1] - bounds[c])   # pseudo-positions
        for t in range(1, int(g.chrlen[c]), 300000):
            off.append(off[-1] + int(((pos >= t) & (pos <= t + 299999)).sum()))
    off = np.array(off, dtype=np.int64)
    assert len(off) - 1 == 399 and off[-1] == n_match
    got = q.run_windows(off, False)
    want = c_oracle.windows(db, None, wei, off, False)
    assert np.array_equal(bits(got[0]), bits(want[0])) and np.array_equal(got[1], want[1])
    assert np.array_equal(bits(got[2]), bits(want[2])) and np.array_equal(got[3], want[3])
    wlik, wlrt = ctx.likelihood(got[0], got[1])
    k = 200
    ol, orr = orc.calculate_likelihoods(want[0][k], want[1][k])
    np.testing.assert_allclose(wlik[k], ol, rtol=LIK_RTOL, equal_nan=True)
    np.testing.assert_allclose(wlrt[k], orr, rtol=LIK_RTOL, equal_nan=True)


def test_upload_staging_throughput(ctx):
    """pinned-host staging path: a 2 GiB host panel goes through the double-buffered slabs intact."""
    import time
    n_snp, n_acc = 1 << 21, 1000
    rng = np.random.default_rng(12)
    host = rng.integers(-1, 3, size=(n_snp, n_acc), dtype=np.int8)
    t0 = time.perf_counter()
    panel = engine.Panel(ctx, n_snp, n_acc)
    panel.upload_rows(0, host)
    panel.upload_wait()
    dt = time.perf_counter() - t0
    print("staging upload: %.2f GB in %.2f s = %.2f GB/s" % (host.nbytes / 1e9, dt, host.nbytes / 1e9 / dt))
    for r0 in (0, 12345, n_snp - 777):
        assert np.array_equal(panel.download_rows(r0, 777), host[r0:r0 + 777])


def test_segregating_rows_on_device(ctx):
    from snpmatch_amd.core import snp_genotype
    rng = np.random.default_rng(31)
    db = rand_db(rng, 70000, 40)
    db[:, 5] = db[:, 4]                       # identical pair: segregating only where one of them is missing? no: never
    g = snp_genotype.Genotype.from_arrays(db, ["a%d" % i for i in range(40)], np.arange(1, 70001), ["1"], [(0, 70000)])
    assert len(orc.segregating_rows(db, np.array([4, 5]))) == 0
    panel = g.panel(ctx)
    assert len(g.identify_segregating_snps(np.array([4, 5]))) == 0
    want = orc.segregating_rows(db, np.array([1, 4, 9, 17]))
    got = np.where(panel.segregating_rows(np.array([1, 4, 9, 17])))[0]
    assert np.array_equal(got, want) and len(got) > 1000
    assert np.array_equal(g.identify_segregating_snps(np.array([1, 4, 9, 17])), want)
    assert g.identify_segregating_snps(np.arange(21)) is None and orc.segregating_rows(db, np.arange(21)) is None


def test_exact_mode_on_accession_major_copy():
    """The strict re-evaluation of flagged accessions through the accession-major packed copy (forced for
    short queries here) gives the reference's fp64 bits, dense and gathered, with and without skip_hets;
    a panel holding the unencodable code 3 falls back to the SNP-major path and is still exact."""
    os.environ["SNPM_ACC_MAJOR_MIN_ROWS"] = "0"
    try:
        c = engine.Context(0)
    finally:
        del os.environ["SNPM_ACC_MAJOR_MIN_ROWS"]
    rng = np.random.default_rng(55)
    n, n_acc = 9000, 300
    for has3 in (False, True):
        db = rand_db(rng, n, n_acc)
        if has3:
            db[::7, 11] = 3
        panel = engine.Panel.from_host(c, db)
        for rows in (None, np.sort(rng.choice(n, size=5003, replace=False)).astype(np.int64)):
            sub = db if rows is None else db[rows]
            codes = sub[:, 7].copy()
            codes[codes < 0] = 0
            codes[codes > 2] = 0
            wei = synth.sample_weights(rng, codes, frac_pl=1.0)       # column 7 sums exact 1.0s: must be re-evaluated
            q = engine.Query(panel, rows, wei)
            for skip in (False, True):
                want_s, want_n = c_oracle.genotyper(db, rows, wei, 1000, skip)
                s, ni, info = q.run(1000, skip, engine.MODE_EXACT, return_info=True)
                assert info["n_strict_reeval"] >= 1
                assert np.array_equal(ni, want_n)
                assert np.array_equal(np.array(s, dtype=int), np.array(want_s, dtype=int))
                if not skip:
                    assert bits(s)[7] == bits(want_s)[7]
        # new rows invalidate the copy: the next run rebuilds it
        db2 = db.copy()
        db2[:100] = rand_db(rng, 100, n_acc)
        panel.upload_rows(0, db2[:100])
        codes = db2[:, 9].copy()
        codes[codes < 0] = 0
        codes[codes > 2] = 0
        wei = synth.sample_weights(rng, codes, frac_pl=1.0)
        want_s, want_n = c_oracle.genotyper(db2, None, wei, 1000, False)
        s, ni, info = engine.Query(panel, None, wei).run(1000, False, engine.MODE_EXACT, return_info=True)
        assert info["n_strict_reeval"] >= 1 and bits(s)[9] == bits(want_s)[9]
        assert np.array_equal(np.array(s, dtype=int), np.array(want_s, dtype=int)) and np.array_equal(ni, want_n)
    c.close()


@pytest.mark.parametrize("forced", [1, 3, 17, 64])
def test_reevaluation_on_the_accession_major_copy_many_shapes(forced):
    """k_strict_sparse_T (flagged accessions re-scored on the accession-major packed copy): the first `forced` accessions are
    re-evaluated (SNPM_DEBUG_REEVAL) and must carry the reference's fp64 bits -- chunk sizes that do not divide by 8, queries
    shorter than one unrolled step, a ragged last chunk, int8 and packed panels, skip_hets, 1 to 64 columns per segment.
    (An LDS-staged variant of this kernel was tried in round 3 and was slower: the pass is bound by the 24 B of weights per
    row it has to read, 1.2 GB for 50M rows, not by how the lanes fetch them -- profiles/r03h_ab_sparse_T_dense.txt.)"""
    for _ in (0,):
        os.environ.update(SNPM_ACC_MAJOR_MIN_ROWS="0", SNPM_DEBUG_REEVAL=str(forced))
        try:
            c = engine.Context(0)
        finally:
            for k in ("SNPM_ACC_MAJOR_MIN_ROWS", "SNPM_DEBUG_REEVAL"):
                del os.environ[k]
        rng = np.random.default_rng(900 + forced)
        sparse_runs = 0
        for n, chunk in ((1, 1000), (13, 7), (9000, 1000), (20011, 1001), (4097, 7)):
            n_acc = int(rng.choice([64, 300, 1135]))
            db = rand_db(rng, n, n_acc)
            for packed in (False, True):
                panel = engine.Panel.from_host(c, db, packed=packed)
                wei = rand_wei(rng, n)
                q = engine.Query(panel, None, wei)
                for skip in (False, True):
                    want_s, want_n = c_oracle.genotyper(db, None, wei, chunk, skip)
                    s, ni, info = q.run(chunk, skip, engine.MODE_EXACT, return_info=True)
                    tag = (n, chunk, n_acc, packed, skip)
                    assert info["all_integer_weights"] or info["n_strict_reeval"] >= min(forced, n_acc), (tag, info)
                    if 0 < info["n_strict_reeval"] <= 64:                  # more: the dense tier re-scores every accession
                        assert info["reeval_path"] == 1, (tag, info)
                        sparse_runs += 1
                    k = min(forced, n_acc)
                    assert np.array_equal(bits(s)[:k], bits(want_s)[:k]), tag
                    assert np.array_equal(ni, want_n) and np.array_equal(np.array(s, dtype=int), np.array(want_s, dtype=int)), tag
                q.free()
                panel.free()
        assert sparse_runs >= 6 or forced == 64
        c.close()


# ------------------------------------------------------------------ packed (2 bits per call) panel format
@pytest.mark.parametrize("n_snp,n_acc,n_match,chunk", [(4000, 1, 1000, 1000), (6000, 64, 3001, 1000), (20000, 1135, 7545, 1000),
                                                       (3000, 1250, None, 1000), (2100, 10000, None, 1000), (130, 17, 129, 7),
                                                       (700, 32768, 301, 1000)])
@pytest.mark.parametrize("skip", [False, True])
def test_packed_panel_modes_vs_oracle(ctx, n_snp, n_acc, n_match, chunk, skip):
    rng = np.random.default_rng(n_snp * 17 + n_acc)
    db = rand_db(rng, n_snp, n_acc)
    if n_acc > 5:
        db[:, 2] = -1
        db[::5, 3] = -7                    # any negative value is "missing"
    panel = engine.Panel.from_host(ctx, db, packed=True)
    # bytes per row: whole 256-B column blocks + the ragged tail at a power-of-two pitch (split layout) where that saves >= 5 % of
    # the row, else rows padded to 256 B (+ 256 B at multiples of 8 KiB)
    row = (n_acc + 3) // 4
    tail = 4
    while tail < row % 256:
        tail *= 2
    split = 0 < row % 256 <= 128 and (256 - tail) * 20 >= row // 256 * 256 + 256 and os.environ.get("SNPM_PACKED_SPLIT", "1") != "0"
    want_pitch = (row // 256 * 256 + tail) if split else (row + 255) // 256 * 256
    assert panel.packed and panel.pitch == want_pitch + (256 if (not split and want_pitch % 8192 == 0) else 0)
    assert panel.pitch == ctx.row_pitch(n_acc, True)
    back = panel.download_rows(0, n_snp)
    assert np.array_equal(back, np.where(db < 0, -1, db))
    if n_match is None:
        row_idx, n = None, n_snp
    else:
        row_idx = np.sort(rng.choice(n_snp, size=n_match, replace=False)).astype(np.int64)
        n = n_match
    wei = rand_wei(rng, n)
    want_s, want_n = c_oracle.genotyper(db, row_idx, wei, chunk, skip)
    q = engine.Query(panel, row_idx, wei)
    bound = q.error_bound(chunk)
    s, ni = q.run(chunk, skip, engine.MODE_STRICT)
    assert np.array_equal(bits(s), bits(want_s)) and np.array_equal(ni, want_n)
    s, ni = q.run(chunk, skip, engine.MODE_EXACT)
    assert np.array_equal(ni, want_n)
    assert np.array_equal(np.array(s, dtype=int), np.array(want_s, dtype=int))
    assert np.max(np.abs(s - want_s)) <= bound
    s, ni = q.run(chunk, skip, engine.MODE_FAST)
    assert np.array_equal(ni, want_n) and np.max(np.abs(s - want_s)) <= bound


@pytest.mark.parametrize("max_parts", [0, 2])
def test_packed_phased_waves(max_parts):
    """k_fast_packed_q4 and k_fast_bits give the lanes of a last wave that covers <= 32 dwords of the row to row groups (phases): every
    phase count (2, 3, 4, 5, 6, 8) and the widths just outside the rule, dense and gathered rows, partial last tiles and
    groups, one long multi-epoch part per column block (SNPM_DEBUG_MAX_PARTS) as well as many short ones, a batch of samples
    (the segmented launch) -- informative counts exact, scores inside the certified bound, integer parts equal."""
    if max_parts:
        os.environ["SNPM_DEBUG_MAX_PARTS"] = str(max_parts)
    try:
        c = engine.Context(0)
    finally:
        os.environ.pop("SNPM_DEBUG_MAX_PARTS", None)
    rng = np.random.default_rng(4711 + max_parts)
    # accessions -> dwords in the last wave -> phases: 512 -> 32 -> 2, 300 -> 19 -> 3, 250 -> 16 -> 4, 170 -> 11 -> 5,
    # 1184 -> 10 -> 6, 1135 -> 7 -> 8, 16 -> 1 -> 8, 1530 -> 32 -> 2, 2400 -> 22 -> 2; 513 / 1025 + 528: 33 dwords, not phased
    # 7300 -> 7 full waves + 9 dwords (one 8-wave block whose last wave has 7 phases), 9300 -> 9 full waves + 6 dwords (4-wave
    # blocks: the third one holds a full wave, a phased wave and two idle ones)
    for n_acc in (512, 300, 250, 170, 1184, 1135, 16, 1530, 2400, 513, 1553, 7300, 9300):
        n_snp = int(rng.choice([1, 5, 63, 64, 71, 1000, 16389, 40003] if not max_parts else [16389, 40003, 70001]))
        if n_acc > 4096:
            n_snp = min(n_snp, 16389)
        db = rand_db(rng, n_snp, n_acc)
        db[:, n_acc - 1] = -1                       # the last accession: no informative site
        panel = engine.Panel.from_host(c, db, packed=True)
        for gathered, frac_pl in ((False, 0.8), (True, 0.8), (False, 0.0), (True, 0.0)):     # 0.0: hard calls (k_fast_bits)
            rows = None
            n = n_snp
            if gathered:
                n = max(1, n_snp - int(rng.integers(0, 50)))
                rows = np.sort(rng.choice(n_snp, size=n, replace=False)).astype(np.int64)
            wei = rand_wei(rng, n, frac_pl)
            skip = bool(rng.integers(0, 2))
            want_s, want_n = c_oracle.genotyper(db, rows, wei, 1000, skip)
            q = engine.Query(panel, rows, wei)
            for mode in (engine.MODE_FAST, engine.MODE_EXACT):
                s, ni = q.run(1000, skip, mode)
                tag = "%d x %d gathered=%s pl=%s skip=%s mode=%d" % (n_snp, n_acc, gathered, frac_pl, skip, mode)
                assert np.array_equal(ni, want_n), tag
                assert np.max(np.abs(s - want_s)) <= q.error_bound(1000), tag
            assert np.array_equal(np.array(s, dtype=int), np.array(want_s, dtype=int)), tag
            q.free()
        if n_snp >= 1000 and not max_parts:
            samples = []
            for b in range(5):
                nb = int(rng.integers(1, n_snp))
                r = np.sort(rng.choice(n_snp, size=nb, replace=False)).astype(np.int64)
                samples.append((r, rand_wei(rng, nb)))
            got = engine.score_batch(panel, samples, 1000, False, engine.MODE_EXACT, likelihoods=False)
            for b, (r, w) in enumerate(samples):
                want_s, want_n = c_oracle.genotyper(db, r, w, 1000, False)
                assert np.array_equal(got["ninfo"][b], want_n), (n_acc, b)
                assert np.array_equal(np.array(got["score"][b], dtype=np.int64), np.array(want_s, dtype=np.int64)), (n_acc, b)
        panel.free()
    c.close()


def test_packed_panel_misc(ctx, golden_dir):
    # codes > 2 cannot be stored
    with pytest.raises(AssertionError, match="packed panel"):
        engine.Panel.from_host(ctx, np.array([[0, 1, 3]], dtype=np.int8), packed=True)
    # device generator: packed and int8 panels hold the same values
    a = engine.Panel(ctx, 3000, 1135, packed=True)
    a.fill_synthetic(1001, snp0=77, acc0=12)
    assert np.array_equal(a.download_rows(0, 3000), synth.panel_values(1001, 77, 3000, 12, 1135))
    # windows (cross) on a packed panel: the reference's per-window fp64 bits
    toy = np.load(os.path.join(golden_dir, "toy_db_cross.npz"))
    g = np.load(os.path.join(golden_dir, "g5_cross.npz"))
    panel = engine.Panel.from_host(ctx, toy["snps"], packed=True)
    for skip in (0, 1):
        q = engine.Query(panel, g["win_rows_db_skip%d" % skip], toy["s_wei"][g["win_rows_sample_skip%d" % skip]])
        s, ni, ts, tn = q.run_windows(g["win_off_skip%d" % skip], bool(skip))
        assert np.array_equal(bits(s), bits(g["win_score_skip%d" % skip])) and np.array_equal(ni, g["win_ninfo_skip%d" % skip])
    # forced re-evaluation of integer-sum accessions goes through the strided packed reader
    rng = np.random.default_rng(8)
    db = rand_db(rng, 6000, 50)
    codes = db[:, 7].copy()
    codes[codes < 0] = 0
    wei = synth.sample_weights(rng, codes, frac_pl=1.0)
    want_s, want_n = c_oracle.genotyper(db, None, wei, 1000, False)
    s, ni, info = engine.Query(engine.Panel.from_host(ctx, db, packed=True), None, wei).run(1000, False, engine.MODE_EXACT, return_info=True)
    assert info["n_strict_reeval"] >= 1 and bits(s)[7] == bits(want_s)[7]
    assert np.array_equal(np.array(s, dtype=int), np.array(want_s, dtype=int)) and np.array_equal(ni, want_n)
    seg = np.where(engine.Panel.from_host(ctx, db, packed=True).segregating_rows(np.array([1, 4, 9])))[0]
    ref = np.where(engine.Panel.from_host(ctx, db).segregating_rows(np.array([1, 4, 9])))[0]
    assert np.array_equal(seg, ref) and len(seg) > 100


def test_packed_large_panel_equals_int8(ctx):
    """10k accessions x 2M SNPs generated on the device in both formats: identical counts, scores within the bound."""
    n_snp, n_acc = 2_000_000, 10_000
    rng = np.random.default_rng(3)
    col = synth.panel_values(777, 0, n_snp, 416, 4)[:, 1]
    codes, wei = synth.planted_sample(rng, col, 0.02)
    res = []
    for packed in (False, True):
        panel = engine.Panel(ctx, n_snp, n_acc, packed=packed)
        panel.fill_synthetic(777)
        q = engine.Query(panel, None, wei)
        res.append(q.run(1000, False, engine.MODE_EXACT) + (q.error_bound(1000),))
        q.free()
        panel.free()
    assert np.array_equal(res[0][1], res[1][1])
    assert np.array_equal(np.array(res[0][0], dtype=int), np.array(res[1][0], dtype=int))
    assert np.max(np.abs(res[0][0] - res[1][0])) <= 2 * res[0][2]
    assert int(np.argmax(res[1][0] / np.maximum(res[1][1], 1))) == 417


def test_fuzz_random_configurations(ctx):
    _fuzz_random_configurations(ctx, 300, 20260101)


def test_fuzz_random_configurations_on_long_tiles():
    """the same differential check when every int8 query walks 248-row LUT tiles (what scans of >= 2M rows use)"""
    os.environ["SNPM_LONG_SCAN_ROWS"] = "1"
    try:
        c = engine.Context(0)
    finally:
        del os.environ["SNPM_LONG_SCAN_ROWS"]
    _fuzz_random_configurations(c, 120, 20261004)
    c.close()


def _fuzz_random_configurations(ctx, n_cases, seed):
    """300 seeded random configurations (shape, chunk, skip_hets, dense / gathered / repeated rows, int8 / packed,
    weight mix incl. zero rows and weights > 1 capped by ninfo) against the C oracle:
    strict mode = fp64 bits, default mode = counts + certified bound, windows = fp64 bits."""
    rng = np.random.default_rng(seed)
    for case in range(n_cases):
        n_snp = int(rng.integers(1, 7000))
        n_acc = int(rng.choice([1, 2, 3, 5, 63, 64, 65, 255, 256, 257, 700, 1135, 2049]))
        packed = bool(rng.integers(0, 2))
        skip = bool(rng.integers(0, 2))
        chunk = int(rng.choice([1, 7, 128, 1000, 1001, 5000]))
        db = rand_db(rng, n_snp, n_acc)
        if not packed and rng.random() < 0.3:
            db[rng.integers(0, n_snp), :] = 3
        kind = rng.integers(0, 3)
        if kind == 0:
            rows, n = None, n_snp
        elif kind == 1:
            n = int(rng.integers(0, n_snp + 1))
            rows = np.sort(rng.choice(n_snp, size=n, replace=False)).astype(np.int64)
        else:
            n = int(rng.integers(1, 2 * n_snp + 2))
            rows = rng.integers(0, n_snp, size=n).astype(np.int64)          # unsorted, with repeats
        wei = rand_wei(rng, n, frac_pl=float(rng.choice([0.0, 0.5, 1.0])))
        if n > 3:
            wei[rng.integers(0, n)] = 0.0
        tag = "case %d: %dx%d packed=%s skip=%s chunk=%d kind=%d n=%d" % (case, n_snp, n_acc, packed, skip, chunk, kind, n)
        panel = engine.Panel.from_host(ctx, db, packed=packed)
        q = engine.Query(panel, rows, wei)
        want_s, want_n = c_oracle.genotyper(db, rows, wei, chunk, skip)
        s, ni = q.run(chunk, skip, engine.MODE_STRICT)
        assert np.array_equal(bits(s), bits(want_s)) and np.array_equal(ni, want_n), tag
        s, ni = q.run(chunk, skip, engine.MODE_EXACT)
        assert np.array_equal(ni, want_n), tag
        assert np.array_equal(np.array(s, dtype=int), np.array(want_s, dtype=int)), tag
        assert np.max(np.abs(s - want_s), initial=0.0) <= q.error_bound(chunk), tag
        if n > 0:
            cuts = np.sort(rng.integers(0, n + 1, size=int(rng.integers(0, 12))))
            off = np.concatenate([[0], cuts, [n]]).astype(np.int64)
            got = q.run_windows(off, skip)
            want = c_oracle.windows(db, rows, wei, off, skip)
            assert np.array_equal(bits(got[0]), bits(want[0])) and np.array_equal(got[1], want[1]), tag
            assert np.array_equal(bits(got[2]), bits(want[2])) and np.array_equal(got[3], want[3]), tag
        if rows is not None and type(panel) is engine.Panel and hasattr(panel, "genotype_once") and np.all(wei <= 1.0):
            # the one-call form of the same sample (snpm_genotype_once: k_once_prep / k_once_finish), plain and dictionary-coded
            # (weights above 1 can push a count above its informative sites: that call asserts like the reference's likeliTest)
            table, inv = np.unique(wei.ravel(), return_inverse=True)
            for coded in (False, True):
                if coded and not 1 <= len(table) <= 65536:      # (an empty match list has no weights to make a table of)
                    continue
                args = (rows, inv.reshape(wei.shape).astype(np.uint16) if coded else wei, None, chunk, skip)
                kw = {"table": table} if coded else {}
                once = panel.genotype_once(*args, engine.MODE_STRICT, **kw)
                assert np.array_equal(bits(once["score"]), bits(want_s)) and np.array_equal(once["ninfo"], want_n), tag
                once = panel.genotype_once(*args, engine.MODE_EXACT, **kw)
                assert np.array_equal(once["ninfo"], want_n) and np.array_equal(once["score"].astype(int), want_s.astype(int)), tag
                assert np.array_equal(bits(once["score"]), bits(s)), tag             # the three-call path's kernels in its geometry
                lik_o, _ = orc.calculate_likelihoods(want_s.astype(int), want_n)
                ok = ~np.isnan(lik_o)
                assert np.array_equal(np.isnan(once["lik"]), ~ok) and np.allclose(once["lik"][ok], lik_o[ok], rtol=1e-12, atol=0), tag
        q.free()
        panel.free()


@pytest.mark.parametrize("long_tiles", [False, True])
def test_fuzz_long_parts_multi_epoch(long_tiles):
    """Same differential check with the fast pass forced into few, long parts (several accumulation epochs
    per part, partial last tiles, odd tile counts), int8 and packed; ``long_tiles``: every int8 query walks the 248-row tiles
    that long scans (>= 2M rows) use (SNPM_LONG_SCAN_ROWS=1)."""
    os.environ["SNPM_DEBUG_MAX_PARTS"] = "3"
    if long_tiles:
        os.environ["SNPM_LONG_SCAN_ROWS"] = "1"
    try:
        c = engine.Context(0)
    finally:
        del os.environ["SNPM_DEBUG_MAX_PARTS"]
        os.environ.pop("SNPM_LONG_SCAN_ROWS", None)
    rng = np.random.default_rng(77)
    for case in range(30):
        n_snp = int(rng.choice([8191, 8192, 8193, 24576, 24577, 30001, 49153, 70000, 15872, 15873, 47617]))
        n_acc = int(rng.choice([3, 64, 257, 1135, 4200]))      # 4200: blocks of >= 256 threads (register-staged LUT tiles)
        packed = bool(rng.integers(0, 2))
        skip = bool(rng.integers(0, 2))
        db = rand_db(rng, n_snp, n_acc)
        wei = rand_wei(rng, n_snp, frac_pl=float(rng.choice([0.0, 0.8])))
        rows = None
        if rng.random() < 0.4:
            rows = np.sort(rng.choice(n_snp, size=n_snp - int(rng.integers(0, 200)), replace=False)).astype(np.int64)
            wei = wei[:len(rows)]
        panel = engine.Panel.from_host(c, db, packed=packed)
        q = engine.Query(panel, rows, wei)
        want_s, want_n = c_oracle.genotyper(db, rows, wei, 1000, skip)
        for mode in (engine.MODE_FAST, engine.MODE_EXACT):
            s, ni = q.run(1000, skip, mode)
            tag = "case %d: %dx%d packed=%s skip=%s mode=%d" % (case, n_snp, n_acc, packed, skip, mode)
            assert np.array_equal(ni, want_n), tag
            assert np.max(np.abs(s - want_s)) <= q.error_bound(1000), tag
        assert np.array_equal(np.array(s, dtype=int), np.array(want_s, dtype=int)), tag
        q.free()
        panel.free()
    c.close()


def test_fuzz_likelihood_and_identity(ctx):
    from scipy import stats
    rng = np.random.default_rng(99)
    for case in range(40):
        m = int(rng.integers(1, 6))
        ln = int(rng.choice([1, 2, 63, 64, 65, 1000, 1025, 3000]))
        n = rng.integers(0, 5000, size=(m, ln))
        frac = rng.random((m, ln))
        y = np.floor(frac * (n + 1)).clip(0, n).astype(float)
        if rng.random() < 0.5:
            y = np.minimum(y + rng.random((m, ln)) * 0.999, n)            # fractional scores (cross windows)
        y[rng.random((m, ln)) < 0.05] = 0.0
        perfect = rng.random((m, ln)) < 0.05
        y[perfect] = n[perfect]
        if rng.random() < 0.2:
            y[0, :] = 0.0                                                  # an all-NaN row
        lik, lrt = ctx.likelihood(y, n)
        for r in range(m):
            wl, wr = orc.calculate_likelihoods(y[r], n[r])
            np.testing.assert_allclose(lik[r], wl, rtol=LIK_RTOL, atol=0, equal_nan=True)
            np.testing.assert_allclose(lrt[r], wr, rtol=LIK_RTOL, atol=0, equal_nan=True)
        # truncation as GenotyperOutput does
        lik_t, _ = ctx.likelihood(y[0], n[0], truncate=True)
        wl, _ = orc.calculate_likelihoods(np.array(y[0], dtype=int), n[0])
        np.testing.assert_allclose(lik_t, wl, rtol=LIK_RTOL, equal_nan=True)
        # identity test against scipy
        x = y[0]
        nn = n[0]
        er = float(rng.choice([0.0005, 0.02, 0.1]))
        out, sf = ctx.binom_identity(x, nn, er, 0.05, return_sf=True)
        want_sf = stats.binom.sf(nn - x - 1, nn, er)
        np.testing.assert_allclose(sf, want_sf, rtol=1e-9, atol=1e-100)      # far tails (1e-280) carry ~1e-8 from lgamma/exp
        safe = np.abs(want_sf - 0.05) > 1e-9
        assert np.array_equal(out[safe], (want_sf >= 0.05).astype(int)[safe])


def test_full_bench_size_properties(ctx):
    """The N=1 bench shard itself (10 000 accessions x 6.25M SNPs, 64 GB resident), checked through invariants
    that do not need an oracle run: (1) all-ones weights: every informative call matches exactly one category,
    so score == ninfo, in fast and default mode; (2) ninfo of sampled accession quads equals the count from the
    numpy twin of the generator; (3) a hard-weight sample planted on accession 417 matches it perfectly and is
    the unique top hit; (4) scores and ninfo are additive over a split of the SNP axis (integer weights: exact);
    (5) mixed PL weights: default-mode counts equal strict-mode counts on the first 300 accessions' worth of
    columns re-scored in reference order on a 200k-SNP slab; (6) mixed PL weights over all 6.25M SNPs: default-mode
    counts equal reference-order counts for every accession and 16 accessions equal the C oracle bit for bit."""
    n_snp, n_acc = 6_250_000, 10_000
    panel = engine.Panel(ctx, n_snp, n_acc)
    panel.fill_synthetic(10050)
    q = engine.Query(panel, None, np.ones((n_snp, 3)))
    s, ni = q.run(1000, False, engine.MODE_EXACT)
    assert np.array_equal(s, ni.astype(np.float64)) and ni.min() > 0.94 * n_snp
    s2, ni2 = q.run(1000, False, engine.MODE_FAST)
    assert np.array_equal(s2, s) and np.array_equal(ni2, ni)
    q.free()
    for c4 in (0, 4996, 9996):
        miss = np.zeros(4, dtype=np.int64)
        for r0 in range(0, n_snp, 1_250_000):
            miss += (synth.panel_values(10050, r0, 1_250_000, c4, 4) < 0).sum(axis=0)
        assert np.array_equal(ni[c4:c4 + 4], n_snp - miss)
    col = np.concatenate([synth.panel_values(10050, r0, 1_250_000, 416, 4)[:, 1] for r0 in range(0, n_snp, 1_250_000)])
    codes = col.copy()
    codes[codes < 0] = 0
    wei = orc.weights_from_gt_codes(codes)
    s, ni = engine.Query(panel, None, wei).run(1000, False, engine.MODE_EXACT)
    assert s[417] == ni[417] and int(np.argmax(s / ni)) == 417
    lik, lrt = ctx.likelihood(s, ni, truncate=True)
    assert lik[417] == 1.0 and (lrt < 3.841).sum() == 1
    h = 3_000_077
    s1, n1 = engine.Query(panel, None, wei[:h], row0=0).run(1000, False, engine.MODE_EXACT)
    s2, n2 = engine.Query(panel, None, wei[h:], row0=h).run(1000, False, engine.MODE_EXACT)
    assert np.array_equal(s1 + s2, s) and np.array_equal(n1 + n2, ni)
    rng = np.random.default_rng(1)
    slab = 200_000
    r0 = 4_000_000
    _, wpl = synth.planted_sample(rng, col[r0:r0 + slab], 0.02)
    qs = engine.Query(panel, None, wpl, row0=r0)
    se, ne = qs.run(1000, False, engine.MODE_EXACT)
    ss, ns = qs.run(1000, False, engine.MODE_STRICT)
    assert np.array_equal(ne, ns) and np.array_equal(np.array(se, dtype=int), np.array(ss, dtype=int))
    assert np.max(np.abs(se - ss)) <= qs.error_bound(1000)
    db = panel.download_rows(r0, slab)[:, :300]
    ws, wn = c_oracle.genotyper(db, None, wpl, 1000, False)
    assert np.array_equal(bits(ss[:300]), bits(ws)) and np.array_equal(ns[:300], wn)
    qs.free()
    # (6) the whole SNP axis with mixed PL weights (SURVEY 8d spot check): default-mode counts == reference-order
    # counts for all 10 000 accessions, and 16 accessions (4 quads regenerated by the numpy twin) agree with the
    # C oracle over all 6.25M rows -- strict scores bit for bit
    _, wfull = synth.planted_sample(rng, col, 0.02)
    qf = engine.Query(panel, None, wfull)
    se, ne, info = qf.run(1000, False, engine.MODE_EXACT, return_info=True)
    ss, ns = qf.run(1000, False, engine.MODE_STRICT)
    assert np.array_equal(ne, ns) and np.array_equal(np.array(se, dtype=np.int64), np.array(ss, dtype=np.int64))
    assert np.max(np.abs(se - ss)) <= qf.error_bound(1000) and info["n_strict_reeval"] < 50
    for c4 in (416, 0, 5000, 9996):
        quad = np.concatenate([synth.panel_values(10050, r, 1_250_000, c4, 4) for r in range(0, n_snp, 1_250_000)])
        ws, wn = c_oracle.genotyper(quad, None, wfull, 1000, False)
        assert np.array_equal(bits(ss[c4:c4 + 4]), bits(ws)) and np.array_equal(ns[c4:c4 + 4], wn)
        assert np.array_equal(np.array(se[c4:c4 + 4], dtype=np.int64), np.array(ws, dtype=np.int64))
    assert int(np.argmax(se / ne)) == 417
    qf.free()


def test_native_file_loader(ctx, tmp_path):
    """snpm_panel_load_file: a flat panel streamed from disk through the pinned staging slabs equals the same
    matrix uploaded from memory (int8 and packed), row windows included; strided host rows upload correctly."""
    import time
    from snpmatch_amd.core import snp_genotype
    rng = np.random.default_rng(17)
    n_snp, n_acc = 700_001, 1135
    db = rand_db(rng, n_snp, n_acc)
    path = str(tmp_path / "db.snpm")
    snp_genotype.save_native(path, db, ["a%d" % i for i in range(n_acc)], np.arange(1, n_snp + 1), ["1"], [(0, n_snp)])
    t0 = time.perf_counter()
    p1 = engine.Panel.from_npy(ctx, os.path.join(path, "snps.npy"))
    p1.upload_wait()
    dt = time.perf_counter() - t0
    print("file loader: %.2f GB in %.3f s = %.2f GB/s" % (db.nbytes / 1e9, dt, db.nbytes / 1e9 / dt))
    for r0 in (0, 65536 - 3, n_snp - 500):
        assert np.array_equal(p1.download_rows(r0, 500), db[r0:r0 + 500])
    p2 = engine.Panel.from_npy(ctx, os.path.join(path, "snps.npy"), packed=True)
    assert np.array_equal(p2.download_rows(1234, 4000), db[1234:5234])
    # the host-side Genotype picks the native loader for flat panels
    g = snp_genotype.Genotype(path, None)
    assert np.array_equal(g.panel(ctx).download_rows(7, 100), db[7:107])
    # partial load into the middle of a panel + strided host rows
    p3 = engine.Panel(ctx, 1000, n_acc)
    p3.fill_synthetic(5)
    with open(os.path.join(path, "snps.npy"), "rb") as fh:
        np.lib.format.read_magic(fh)
        np.lib.format.read_array_header_1_0(fh)
        off = fh.tell()
    p3.load_file(os.path.join(path, "snps.npy"), off + 50 * n_acc, row0=100, nrows=200)
    got = p3.download_rows(0, 1000)
    assert np.array_equal(got[100:300], db[50:250]) and np.array_equal(got[:100], synth.panel_values(5, 0, 100, 0, n_acc))
    wide = np.full((600, n_acc + 37), 9, dtype=np.int8)
    wide[:, :n_acc] = db[:600]
    p4 = engine.Panel(ctx, 600, n_acc)
    check_rc = ctx.lib.snpm_panel_upload_rows(p4.h, 0, 600, wide.ctypes.data, wide.shape[1])
    assert check_rc == 0
    assert np.array_equal(p4.download_rows(0, 600), db[:600])
    with pytest.raises(AssertionError, match="cannot open"):
        p4.load_file(str(tmp_path / "missing.npy"), 0)


# ------------------------------------------------------------------ in-silico F1s (core/csmatch.py:106-129)
@pytest.mark.parametrize("n_snp,n_match,n_acc,k,packed", [
    (50, None, 12, 2, False), (7, None, 5, 3, False), (2048, None, 40, 10, False), (2049, 2047, 40, 10, True),
    (30000, 9000, 64, 10, False), (70000, None, 33, 10, True), (300000, 250000, 16, 5, False),
    (120000, None, 12, 10, False), (20000, 17000, 1135, 32, False)])
def test_insilico_f1_pairs_bitexact(ctx, n_snp, n_match, n_acc, k, packed):
    """k_f1_*: scores carry the bits of the reference's np.sum expression, counts are exact"""
    rng = np.random.default_rng(n_snp + 7 * k)
    db = rand_db(rng, n_snp, n_acc)
    db[:, 1] = db[:, 0]                                    # identical parents: no het class at all
    if n_acc > 4:
        db[rng.random(n_snp) < 0.9, 3] = -1                # mostly missing parent: short lists
    panel = engine.Panel.from_host(ctx, db, packed=packed)
    row_idx = None if n_match is None else np.sort(rng.choice(n_snp, size=n_match, replace=False)).astype(np.int64)
    n = n_snp if n_match is None else n_match
    wei = rand_wei(rng, n)
    sel = np.concatenate(([0, 1], rng.permutation(np.arange(2, n_acc))[:k - 2]))[:k]
    q = engine.Query(panel, row_idx, wei)
    s, ni = q.f1_pairs(sel)
    rows = slice(None) if row_idx is None else row_idx
    want_s, want_n = orc.insilico_f1_pairs(db[rows][:, sel], wei)
    assert np.array_equal(ni, want_n)
    assert np.array_equal(bits(s), bits(want_s))
    assert ni[0] == np.count_nonzero((db[rows, 0] == 0) | (db[rows, 0] == 1))    # pair (0, 1): clones
    q.free()


def test_insilico_f1_pairs_in_several_batches():
    """the 45 pairs processed a few at a time (scratch slab limited through SNPM_F1_SLAB_BYTES): same bits"""
    os.environ["SNPM_F1_SLAB_BYTES"] = str(70000 * 8 * 7)          # 7 pairs per batch
    try:
        c = engine.Context(0)
    finally:
        del os.environ["SNPM_F1_SLAB_BYTES"]
    rng = np.random.default_rng(12)
    db = rand_db(rng, 70000, 24)
    wei = rand_wei(rng, 70000)
    sel = rng.permutation(24)[:10]
    q = engine.Query(engine.Panel.from_host(c, db), None, wei)
    s, ni = q.f1_pairs(sel)
    want_s, want_n = orc.insilico_f1_pairs(db[:, sel], wei)
    assert np.array_equal(ni, want_n) and np.array_equal(bits(s), bits(want_s))
    q.free()
    c.close()


def test_insilico_f1_pairs_edges(ctx):
    rng = np.random.default_rng(8)
    db = rand_db(rng, 100, 9)
    panel = engine.Panel.from_host(ctx, db)
    q = engine.Query(panel, np.zeros(0, dtype=np.int64), np.zeros((0, 3)))
    s, ni = q.f1_pairs([0, 1, 2])
    assert s.tolist() == [0.0, 0.0, 0.0] and ni.tolist() == [0, 0, 0]            # no matched SNPs
    q.free()
    q = engine.Query(panel, None, rand_wei(rng, 100))
    assert len(q.f1_pairs([4])[0]) == 0 and len(q.f1_pairs([])[0]) == 0           # fewer than two accessions
    with pytest.raises(AssertionError, match="outside the panel"):
        q.f1_pairs([0, 9])
    with pytest.raises(AssertionError, match="between 0 and 32"):
        q.f1_pairs(list(range(9)) * 4)
    q.free()


def test_no_device_memory_leak_over_many_objects():
    """panels, queries (all modes, windows, in-silico crosses) created and freed in a loop: the free device
    memory reported by the driver comes back to where it started"""
    import torch
    c = engine.Context(0)
    rng = np.random.default_rng(21)
    db = rand_db(rng, 20000, 300)
    wei = rand_wei(rng, 20000)
    off = np.array([0, 5000, 5000, 20000], dtype=np.int64)

    def cycle(packed):
        panel = engine.Panel.from_host(c, db, packed=packed)
        q = engine.Query(panel, None, wei)
        for mode in (engine.MODE_EXACT, engine.MODE_STRICT, engine.MODE_FAST):
            q.run(1000, False, mode)
        q.run_windows(off)
        q.f1_pairs(np.arange(10))
        q.free()
        rows = np.sort(rng.choice(20000, size=3000, replace=False)).astype(np.int64)
        q = engine.Query(panel, rows, wei[:3000])
        q.run(1000, True, engine.MODE_EXACT)
        q.free()
        panel.free()

    cycle(False)
    cycle(True)                      # grow-only workspaces of the context are now at their final size
    c.synchronize()
    free0 = torch.cuda.mem_get_info(0)[0]
    for i in range(40):
        cycle(bool(i & 1))
    c.synchronize()
    free1 = torch.cuda.mem_get_info(0)[0]
    assert free0 - free1 < (8 << 20), "device memory leaked: %d bytes" % (free0 - free1)
    c.close()


# ------------------------------------------------------------------ hard-call samples on packed panels (k_fast_bits)
def _hard_weights(rng, n, one_hot=True):
    if one_hot:
        codes = rng.choice(np.array([0, 1, 2], dtype=np.int8), size=n, p=[0.6, 0.3, 0.1])
        return orc.weights_from_gt_codes(codes)
    return rng.integers(0, 2, size=(n, 3)).astype(np.float64)          # any 0/1 pattern, also all-zero rows


@pytest.mark.parametrize("max_parts", [0, 3])
def test_packed_hard_calls_bit_parallel_path(max_parts):
    """all weights 0 or 1 on a packed panel: scores are counts (k_fast_bits) and equal the oracle exactly in every
    mode; short / long parts (several epochs), dense and gathered rows, skip_hets, non-one-hot 0/1 patterns"""
    old = os.environ.get("SNPM_DEBUG_MAX_PARTS")
    if max_parts:
        os.environ["SNPM_DEBUG_MAX_PARTS"] = str(max_parts)
    try:
        c = engine.Context(0)
    finally:
        if max_parts:
            if old is None:
                del os.environ["SNPM_DEBUG_MAX_PARTS"]
            else:
                os.environ["SNPM_DEBUG_MAX_PARTS"] = old
    rng = np.random.default_rng(50 + max_parts)
    shapes = [(1, 1), (7, 5), (63, 64), (64, 17), (65, 1135), (1000, 4200), (8191, 257), (30001, 1135), (70000, 64), (24577, 4200)]
    for n_snp, n_acc in shapes:
        db = rand_db(rng, n_snp, n_acc)
        panel = engine.Panel.from_host(c, db, packed=True)
        for gather in (False, True):
            rows = None
            n = n_snp
            if gather:
                n = int(rng.integers(0, n_snp + 1))
                rows = np.sort(rng.choice(n_snp, size=n, replace=False)).astype(np.int64)
            for one_hot in (True, False):
                wei = _hard_weights(rng, n, one_hot)
                q = engine.Query(panel, rows, wei)
                assert q.error_bound(1000) == 0.0
                for skip in (False, True):
                    want_s, want_n = c_oracle.genotyper(db, rows, wei, 1000, skip)
                    for mode in (engine.MODE_FAST, engine.MODE_EXACT, engine.MODE_STRICT):
                        s, ni = q.run(1000, skip, mode)
                        tag = "%d x %d gather=%s one_hot=%s skip=%s mode=%d" % (n_snp, n_acc, gather, one_hot, skip, mode)
                        assert np.array_equal(ni, want_n), tag
                        assert np.array_equal(s, want_s), tag
                q.free()
        panel.free()
    c.close()
