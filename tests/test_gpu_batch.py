"""
Many samples per call and certified fast windows (-m gpu): the segmented fast pass (k_fast<SEG>) with the
per-(segment, accession) certificate against single runs and the C oracle.  Reference: one sample per process
(core/snpmatch.py:256-268), one matchGTsAccs call per window (core/csmatch.py:80-90).
"""
import os

import numpy as np
import pytest

from oracle import c_oracle
from oracle import snpmatch_oracle as orc
from snpmatch_amd import engine, synth

pytestmark = pytest.mark.gpu
LIK_RTOL = 1e-12


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def rand_db(rng, n, n_acc):
    return rng.choice(np.array([-1, 0, 1, 2], dtype=np.int8), size=(n, n_acc), p=[0.05, 0.60, 0.33, 0.02])


def make_ctx(**env):
    for k, v in env.items():
        os.environ[k] = str(v)
    try:
        return engine.Context(0)
    finally:
        for k in env:
            del os.environ[k]


def mixed_samples(rng, db, count):
    """samples of very different kinds: planted with PL weights, hard calls, empty, tiny, perfect PL matches"""
    n_snp, n_acc = db.shape
    out = []
    for b in range(count):
        kind = b % 6
        n = int(rng.choice([0, 1, 127, 128, 129, 1000, 4097, 12000])) if kind == 5 else int(rng.integers(2000, 15000))
        n = min(n, n_snp)
        rows = np.sort(rng.choice(n_snp, size=n, replace=False)).astype(np.int64)
        acc = int(rng.integers(0, n_acc))
        codes = db[rows, acc].copy()
        codes[codes < 0] = 0
        codes[codes > 2] = 0
        if kind == 1:
            wei = orc.weights_from_gt_codes(codes)                       # hard 0/1 calls
        elif kind == 2:
            wei = synth.sample_weights(rng, codes, frac_pl=1.0)          # perfect PL match: an exact-integer score
        else:
            flip = rng.random(n) < 0.03
            codes[flip] = rng.integers(0, 3, size=int(flip.sum()))
            wei = synth.sample_weights(rng, codes, frac_pl=0.8)
        out.append((rows, wei))
    return out


@pytest.mark.parametrize("packed", [False, True])
@pytest.mark.parametrize("skip", [False, True])
def test_batch_equals_single_runs_and_oracle(packed, skip):
    ctx = make_ctx()
    rng = np.random.default_rng(21 + packed)
    n_snp, n_acc = 60_000, 1135
    db = rand_db(rng, n_snp, n_acc)
    if not packed:
        db[:, 3] = 3                           # out-of-range codes: informative, never matching
    db[:, 2] = -1
    panel = engine.Panel.from_host(ctx, db, packed=packed)
    samples = mixed_samples(rng, db, 32)
    got = engine.score_batch(panel, samples, 1000, skip, engine.MODE_EXACT)
    assert got["pairs_reeval"] >= 1 and not got["strict_fallback"]       # the perfect PL matches
    strict = engine.score_batch(panel, samples, 1000, skip, engine.MODE_STRICT, likelihoods=False)
    fast = engine.score_batch(panel, samples, 1000, skip, engine.MODE_FAST, likelihoods=False)
    for b, (rows, wei) in enumerate(samples):
        want_s, want_n = c_oracle.genotyper(db, rows, wei, 1000, skip)
        assert np.array_equal(got["ninfo"][b], want_n), b
        assert np.array_equal(np.array(got["score"][b], dtype=np.int64), np.array(want_s, dtype=np.int64)), b
        assert np.array_equal(bits(strict["score"][b]), bits(want_s)) and np.array_equal(strict["ninfo"][b], want_n), b
        assert np.array_equal(fast["ninfo"][b], want_n) and np.max(np.abs(fast["score"][b] - want_s), initial=0) < 1e-7
        q = engine.Query(panel, rows, wei)
        s1, n1 = q.run(1000, skip, engine.MODE_EXACT)
        q.free()
        assert np.array_equal(n1, got["ninfo"][b])
        assert np.array_equal(np.array(s1, dtype=np.int64), np.array(got["score"][b], dtype=np.int64))
        wl, wr = orc.calculate_likelihoods(np.array(want_s, dtype=np.int64), want_n)
        np.testing.assert_allclose(got["lik"][b], wl, rtol=LIK_RTOL, equal_nan=True)
        np.testing.assert_allclose(got["lrt"][b], wr, rtol=LIK_RTOL, equal_nan=True)
    ctx.close()


@pytest.mark.parametrize("n_acc", [257, 1135, 1300, 2048, 2400, 3072])
def test_dense_windows_on_narrow_packed_panels(n_acc):
    """The segmented pass on packed panels whose blocks have one to three waves (16- / 32-row tiles, a phased last wave), DENSE
    rows: certified fast windows against the reference-order windows -- informative counts, integer parts and totals.  (A
    build of exactly this variant once lost per-lane values to a spill placed inside a divergent region: every column of
    the full waves came back with garbage counts.)"""
    ctx = make_ctx()
    rng = np.random.default_rng(n_acc)
    for n_snp in (32, 64, 257, 1000, 9000):
        db = rand_db(rng, n_snp, n_acc)
        panel = engine.Panel.from_host(ctx, db, packed=True)
        codes = rng.choice(np.array([0, 1, 2], dtype=np.int8), size=n_snp, p=[0.6, 0.35, 0.05])
        wei = synth.sample_weights(rng, codes, 0.8)
        q = engine.Query(panel, None, wei)
        for off in ([0, n_snp], [0, 1, n_snp], [0, n_snp // 3, 2 * n_snp // 3, n_snp],
                    np.concatenate([[0], np.sort(rng.integers(0, n_snp + 1, size=6)), [n_snp]]).tolist()):
            off = np.array(off, dtype=np.int64)
            ws, wn, ts, tn = q.run_windows(off, False)
            fs, fn, fts, ftn = q.run_windows(off, False, fast=True)
            assert np.array_equal(fn, wn) and np.array_equal(ftn, tn), (n_snp, off.tolist())
            assert np.array_equal(np.array(fs, dtype=np.int64), np.array(ws, dtype=np.int64)), (n_snp, off.tolist())
            assert np.array_equal(np.array(fts, dtype=np.int64), np.array(ts, dtype=np.int64)), (n_snp, off.tolist())
            for w in range(len(off) - 1):                         # and the oracle, window by window
                if off[w + 1] == off[w]:
                    continue
                want_s, want_n = c_oracle.genotyper(db[off[w]:off[w + 1]], None, wei[off[w]:off[w + 1]], 1 << 30, False)
                assert np.array_equal(wn[w], want_n) and np.array_equal(bits(ws[w]), bits(want_s)), (n_snp, w)
        q.free()
        panel.free()
    ctx.close()


def test_batch_device_inputs_forced_pairs_and_fallback():
    import torch
    ctx = make_ctx(SNPM_DEBUG_REEVAL=3)          # accessions 0..2 of every sample go through the pair re-evaluation
    rng = np.random.default_rng(5)
    n_snp, n_acc = 30_000, 300
    db = rand_db(rng, n_snp, n_acc)
    db[:, 100:250] = db[:, [100]]               # 150 identical accessions: one sample matches all of them perfectly
    panel = engine.Panel.from_host(ctx, db)
    samples = mixed_samples(rng, db, 8)
    off = np.concatenate([[0], np.cumsum([len(r) for r, _ in samples])]).astype(np.int64)
    d_rows = torch.as_tensor(np.concatenate([r for r, _ in samples]), device="cuda:0")
    d_wei = torch.as_tensor(np.concatenate([w for _, w in samples]), device="cuda:0")
    got = engine.score_batch(panel, None, 1000, False, engine.MODE_EXACT, device=(d_rows.data_ptr(), d_wei.data_ptr(), off))
    assert got["pairs_reeval"] >= 3 * 8 and not got["strict_fallback"]
    for b, (rows, wei) in enumerate(samples):
        want_s, want_n = c_oracle.genotyper(db, rows, wei, 1000, False)
        assert np.array_equal(got["ninfo"][b], want_n)
        assert np.array_equal(np.array(got["score"][b], dtype=np.int64), np.array(want_s, dtype=np.int64))
        assert np.array_equal(bits(got["score"][b][:3]), bits(want_s[:3]))          # re-evaluated pairs carry the reference's bits
    # too many uncertain pairs for the sparse tier (cap shrinks with the chunk count): strict pass for every sample
    rows = np.arange(n_snp, dtype=np.int64)
    codes = db[:, 100].copy()
    codes[codes < 0] = 0
    big = [(rows, synth.sample_weights(rng, codes, frac_pl=1.0))] * 3
    got = engine.score_batch(panel, big, 7, False, engine.MODE_EXACT)           # 4286 chunks -> cap 1957 pairs; 3 x 153 flagged
    want_s, want_n = c_oracle.genotyper(db, rows, big[0][1], 7, False)
    assert got["pairs_reeval"] >= 450
    for b in range(3):
        assert np.array_equal(got["ninfo"][b], want_n)
        assert np.array_equal(np.array(got["score"][b], dtype=np.int64), np.array(want_s, dtype=np.int64))
    ctx.close()


@pytest.mark.parametrize("packed", [False, True])
def test_fast_windows_certified_against_strict(packed):
    ctx = make_ctx()
    rng = np.random.default_rng(8 + packed)
    n_snp, n_acc, n_match = 80_000, 1135, 30_000
    db = rand_db(rng, n_snp, n_acc)
    panel = engine.Panel.from_host(ctx, db, packed=packed)
    rows = np.sort(rng.choice(n_snp, size=n_match, replace=False)).astype(np.int64)
    # an F1-like sample: left half of the genome from accession 10, right half from accession 20 -> many windows
    # are perfect matches of PL-weighted calls (exact-integer window scores: the certificate must flag them)
    codes = np.where(np.arange(n_match) < n_match // 2, db[rows, 10], db[rows, 20]).astype(np.int8)
    codes[codes < 0] = 0
    wei = synth.sample_weights(rng, codes, frac_pl=0.9)
    cuts = np.sort(rng.choice(np.arange(1, n_match), size=398, replace=False))
    off = np.concatenate([[0, 0], cuts, [n_match, n_match]]).astype(np.int64)      # empty first and last windows
    q = engine.Query(panel, rows, wei)
    for skip in (False, True):
        want = q.run_windows(off, skip)                                      # reference order: the oracle's bits
        ws = c_oracle.windows(db, rows, wei, off, skip)
        assert np.array_equal(bits(want[0]), bits(ws[0])) and np.array_equal(want[1], ws[1])
        got = q.run_windows(off, skip, fast=True)
        info = q.last_windows_info
        assert info["pairs_reeval"] >= 100 and not info["strict_fallback"]
        assert np.array_equal(got[1], want[1]) and np.array_equal(got[3], want[3])
        assert np.array_equal(got[0].astype(np.int64), want[0].astype(np.int64))
        assert np.array_equal(got[2].astype(np.int64), want[2].astype(np.int64))
        np.testing.assert_allclose(got[0], want[0], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(got[2], want[2], rtol=1e-12)
        lik_f, lrt_f = ctx.likelihood(got[0], got[1])
        lik_s, lrt_s = ctx.likelihood(want[0], want[1])
        np.testing.assert_allclose(lik_f, lik_s, rtol=1e-9, equal_nan=True)       # north_star: 1e-6
    # hard calls: every order is exact -> bit-identical, nothing to re-evaluate
    hard = orc.weights_from_gt_codes(codes)
    qh = engine.Query(panel, rows, hard)
    want = qh.run_windows(off, False)
    got = qh.run_windows(off, False, fast=True)
    assert qh.last_windows_info == {"pairs_reeval": 0, "totals_reeval": 0, "strict_fallback": False}
    for a, b in zip(got, want):
        assert np.array_equal(a, b)
    ctx.close()


def test_coded_weights_batch_equals_fp64_batch():
    """weights sent as uint16 dictionary codes (exp(-PL/10) table from numpy) give the fp64 path's bits"""
    ctx = make_ctx()
    rng = np.random.default_rng(31)
    n_snp, n_acc = 50_000, 257
    db = rand_db(rng, n_snp, n_acc)
    panel = engine.Panel.from_host(ctx, db)
    samples = mixed_samples(rng, db, 12)
    tab = engine.pl_table()
    coded = []
    for rows, wei in samples:
        c = engine.weight_codes(wei, tab)
        assert c is not None and c.dtype == np.uint16
        coded.append((rows, c))
    want = engine.score_batch(panel, samples, 1000, False, engine.MODE_EXACT)
    got = engine.score_batch(panel, coded, 1000, False, engine.MODE_EXACT, table=tab)
    for k in ("score", "ninfo", "lik", "lrt"):
        assert np.array_equal(got[k].view(np.uint64), want[k].view(np.uint64)), k
    assert got["pairs_reeval"] == want["pairs_reeval"]
    assert engine.weight_codes(np.array([[0.3, 1.0, 0.0]]), tab) is None          # 0.3 is no exp(-k/10)
    # a row index outside the panel is reported, not read
    bad = [(np.array([1, 2, n_snp], dtype=np.int64), np.ones((3, 3)))]
    with pytest.raises(AssertionError, match="outside the panel"):
        engine.score_batch(panel, bad)
    ctx.close()


def test_batch_genotyper_writes_the_single_sample_files(tmp_path, golden_dir):
    """snpmatch.genotype_batch: the files of Genotyper for every sample, from one batched call"""
    from snpmatch_amd.core import parsers, snp_genotype, snpmatch
    toy = np.load(os.path.join(golden_dir, "toy_db.npz"))
    g = snp_genotype.Genotype.from_arrays(toy["snps"], toy["accs"], toy["positions"], toy["chrs"], toy["regions"])
    rng = np.random.default_rng(2)
    inputs, outs = [], []
    for b in range(5):
        keep = np.sort(rng.choice(len(toy["s_pos"]), size=len(toy["s_pos"]) - 100 * b, replace=False))
        inp = parsers.ParseInputs("")
        inp.load_snp_info(toy["s_chrs"][keep], toy["s_pos"][keep], toy["s_gt"][keep], toy["s_wei"][keep], toy["s_dp"])
        inputs.append(inp)
        outs.append(str(tmp_path / ("batch%d" % b)))
    snpmatch.genotype_batch(inputs, g, outs)
    for b, inp in enumerate(inputs):
        single = str(tmp_path / ("single%d" % b))
        snpmatch.Genotyper(inp, g, single, run_genotyper=True)
        for ext in (".scores.txt", ".matches.json"):
            assert open(outs[b] + ext).read() == open(single + ext).read(), (b, ext)
