"""
Shared-row scan of a batch of samples (-m gpu): every DB row read once and scored against all samples as an int8 MFMA
contraction of fixed-point weight digits with the one-hot panel (snpmatch_amd/csrc/snpm_k_shared.hpp), same certificate and
reference-order re-evaluation as the per-sample pass.  Reference: one `Genotyper.genotyper` run per sample over the same
panel (core/snpmatch.py:207-233, the chunk loop :218-225); counts and informative sites must be bit-equal to the C oracle
and to the per-sample pass, for overlapping and disjoint marker sets, int8 and packed panels.
"""
import os

import numpy as np
import pytest

from oracle import c_oracle
from oracle import snpmatch_oracle as orc
from snpmatch_amd import engine, synth

pytestmark = pytest.mark.gpu
LIK_RTOL = 1e-12


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


def sample_on(rng, db, rows, kind):
    """weights of a sample planted on a random accession at the given rows: kind 0 PL + 3 % wrong calls, 1 hard 0/1 calls,
    2 perfect PL match (an exact-integer score for the planted accession)"""
    n_acc = db.shape[1]
    acc = int(rng.integers(0, n_acc))
    codes = db[rows, acc].copy()
    codes[codes < 0] = 0
    codes[codes > 2] = 0
    if kind == 1:
        return orc.weights_from_gt_codes(codes)
    if kind == 2:
        return synth.sample_weights(rng, codes, frac_pl=1.0)
    flip = rng.random(len(rows)) < 0.03
    codes[flip] = rng.integers(0, 3, size=int(flip.sum()))
    return synth.sample_weights(rng, codes, frac_pl=0.8)


def chip_samples(rng, db, count, n_markers, drop=0.05, extra=0.02):
    """samples genotyped on one marker set: each lacks a few markers and has a few of its own"""
    n_snp = db.shape[0]
    base = np.sort(rng.choice(n_snp, size=n_markers, replace=False))
    out = []
    for b in range(count):
        keep = base[rng.random(n_markers) >= drop]
        own = rng.choice(n_snp, size=int(extra * n_markers), replace=False)
        rows = np.unique(np.concatenate([keep, own])).astype(np.int64)
        out.append((rows, sample_on(rng, db, rows, b % 3)))
    return out


def check_against_oracle(db, samples, got, skip, lik=True, digits=5):
    for b, (rows, wei) in enumerate(samples):
        want_s, want_n = c_oracle.genotyper(db, rows, wei, 1000, skip)
        assert np.array_equal(got["ninfo"][b], want_n), b
        assert np.array_equal(np.array(got["score"][b], dtype=np.int64), np.array(want_s, dtype=np.int64)), b
        # an unflagged score is the fixed-point sum: below the reference's by at most 2^-F per matched SNP
        assert np.max(np.abs(got["score"][b] - want_s), initial=0) < 1e-7 + len(rows) * 2.0 ** -(8 * (digits - 1) + 6), b
        if lik:
            wl, wr = orc.calculate_likelihoods(np.array(want_s, dtype=np.int64), want_n)
            np.testing.assert_allclose(got["lik"][b], wl, rtol=LIK_RTOL, equal_nan=True)
            np.testing.assert_allclose(got["lrt"][b], wr, rtol=LIK_RTOL, equal_nan=True)


@pytest.mark.parametrize("packed", [False, True])
@pytest.mark.parametrize("skip", [False, True])
def test_shared_rows_equal_oracle_and_per_sample_pass(packed, skip):
    ctx = make_ctx()
    rng = np.random.default_rng(77 + packed + 2 * skip)
    n_snp, n_acc = 50_000, 1135
    db = rand_db(rng, n_snp, n_acc)
    db[:, 2] = -1
    panel = engine.Panel.from_host(ctx, db, packed=packed)
    samples = chip_samples(rng, db, 37, 6000)
    samples[5] = (np.zeros(0, dtype=np.int64), np.zeros((0, 3)))           # an empty sample
    samples[6] = (samples[6][0][:1], samples[6][1][:1])                     # one SNP
    engine.batch_configure(ctx, shared_rows=1)
    got = engine.score_batch(panel, samples, 1000, skip, engine.MODE_EXACT)
    st = engine.batch_last_stats(ctx)
    assert got["shared_rows"] and st["taken"] and st["digits"] == 5, st       # chosen by the longest sample: 2^-38 per matched SNP
    union = np.unique(np.concatenate([r for r, _ in samples]))
    assert got["union_rows"] == len(union) == st["union_rows"]
    assert got["pairs_reeval"] >= 1 and not got["strict_fallback"]          # the perfect PL matches
    check_against_oracle(db, samples, got, skip)
    engine.batch_configure(ctx, shared_rows=0)
    seg = engine.score_batch(panel, samples, 1000, skip, engine.MODE_EXACT)
    assert not seg["shared_rows"] and engine.batch_last_stats(ctx)["why_not"] == "policy"
    assert np.array_equal(seg["ninfo"], got["ninfo"])
    assert np.array_equal(seg["score"].astype(np.int64), got["score"].astype(np.int64))
    np.testing.assert_allclose(seg["lik"], got["lik"], rtol=LIK_RTOL, equal_nan=True)
    # the fast pass alone (no certificate): the quantisation is far below what MODE_FAST promises
    engine.batch_configure(ctx, shared_rows=1)
    fast = engine.score_batch(panel, samples, 1000, skip, engine.MODE_FAST, likelihoods=False)
    assert fast["shared_rows"] and np.array_equal(fast["ninfo"], got["ninfo"])
    assert np.max(np.abs(fast["score"] - seg["score"])) < 1e-7 + 6400 * 2.0 ** -38
    ctx.close()


@pytest.mark.parametrize("digits", [3, 5, 6, 7])
def test_fewer_digits_flag_more_pairs_and_stay_exact(digits):
    """the certificate carries the quantisation: with coarse fixed-point weights more (sample, accession) pairs go through
    the reference-order re-evaluation, the counts do not change"""
    ctx = make_ctx(SNPM_SHARED_TILES=5)                                     # several row tiles on a small job
    rng = np.random.default_rng(100 + digits)
    n_snp, n_acc = 30_000, 700
    db = rand_db(rng, n_snp, n_acc)
    panel = engine.Panel.from_host(ctx, db)
    samples = chip_samples(rng, db, 24, 3000)
    engine.batch_configure(ctx, shared_rows=1, digits=digits)
    got = engine.score_batch(panel, samples, 1000, False, engine.MODE_EXACT)
    st = engine.batch_last_stats(ctx)
    assert st["taken"] and st["digits"] == digits and st["row_tiles"] >= 2, st
    check_against_oracle(db, samples, got, False, digits=digits)
    if digits == 3:
        assert got["pairs_reeval"] > 100 or got["strict_fallback"]          # 2^-22 per SNP: hundreds of unproven pairs
    ctx.close()


@pytest.mark.parametrize("packed", [False, True])
@pytest.mark.parametrize("n_acc", [1, 5, 130, 257, 1135, 2100])
def test_widths_and_disjoint_marker_sets(packed, n_acc):
    """panel widths around the 128-accession wave tiles (split packed layouts included); samples that share no row at all"""
    ctx = make_ctx()
    rng = np.random.default_rng(n_acc * 2 + packed)
    n_snp = 20_000
    db = rand_db(rng, n_snp, n_acc)
    panel = engine.Panel.from_host(ctx, db, packed=packed)
    perm = rng.permutation(n_snp)
    samples = []
    for b in range(9):                                                      # disjoint: sample b owns its own slice of the rows
        rows = np.sort(perm[b * 2000:b * 2000 + int(rng.integers(1, 2000))]).astype(np.int64)
        samples.append((rows, sample_on(rng, db, rows, b % 3)))
    engine.batch_configure(ctx, shared_rows=1)
    got = engine.score_batch(panel, samples, 1000, False, engine.MODE_EXACT)
    assert got["shared_rows"] and got["union_rows"] == sum(len(r) for r, _ in samples)
    check_against_oracle(db, samples, got, False)
    samples = chip_samples(rng, db, 11, 2500, drop=0.3, extra=0.3)
    got = engine.score_batch(panel, samples, 777, True, engine.MODE_EXACT)   # another chunk length, het calls skipped
    assert got["shared_rows"]
    for b, (rows, wei) in enumerate(samples):
        want_s, want_n = c_oracle.genotyper(db, rows, wei, 777, True)
        assert np.array_equal(got["ninfo"][b], want_n) and np.array_equal(got["score"][b].astype(np.int64), want_s.astype(np.int64)), b
    ctx.close()


def test_batches_the_contraction_cannot_take_fall_back():
    ctx = make_ctx()
    rng = np.random.default_rng(9)
    n_snp, n_acc = 20_000, 300
    db = rand_db(rng, n_snp, n_acc)
    panel = engine.Panel.from_host(ctx, db)
    engine.batch_configure(ctx, shared_rows=1)
    samples = chip_samples(rng, db, 10, 2000)
    # (a) a row list that is not increasing
    rows, wei = samples[3]
    order = rng.permutation(len(rows))
    shuffled = list(samples)
    shuffled[3] = (rows[order], wei[order])
    got = engine.score_batch(panel, shuffled, 1000, False, engine.MODE_EXACT)
    assert not got["shared_rows"] and engine.batch_last_stats(ctx)["why_not"] == "a row list is not strictly increasing"
    check_against_oracle(db, shuffled, got, False)
    # (b) weights outside [0, 1]
    heavy = list(samples)
    heavy[7] = (samples[7][0], samples[7][1] * 3.0)
    got = engine.score_batch(panel, heavy, 1000, False, engine.MODE_EXACT, likelihoods=False)     # (scores above ninfo: no likelihoods, the reference asserts y <= n)
    assert not got["shared_rows"] and engine.batch_last_stats(ctx)["why_not"] == "a weight outside [0, 1]"
    check_against_oracle(db, heavy, got, False, lik=False)
    # (c) a panel with call codes > 2 (informative, matching nothing): the one-hot expansion has no class for them
    db2 = db.copy()
    db2[::7, 11] = 3
    panel2 = engine.Panel.from_host(ctx, db2)
    got = engine.score_batch(panel2, samples, 1000, False, engine.MODE_EXACT)
    assert not got["shared_rows"] and engine.batch_last_stats(ctx)["why_not"] == "call codes > 2 in the panel"
    check_against_oracle(db2, samples, got, False)
    # (d) strict mode never takes it; (e) the automatic policy keeps host batches and sparse overlaps on the per-sample pass
    got = engine.score_batch(panel, samples, 1000, False, engine.MODE_STRICT, likelihoods=False)
    assert not got["shared_rows"]
    engine.batch_configure(ctx, shared_rows=-1)
    got = engine.score_batch(panel, samples, 1000, False, engine.MODE_EXACT)
    assert not got["shared_rows"] and engine.batch_last_stats(ctx)["why_not"] == "policy"
    ctx.close()


def test_automatic_policy_on_device_inputs_and_passes_over_groups():
    import torch
    ctx = make_ctx(SNPM_SHARED_WS_MB=1)                                     # a tiny digit-matrix budget: several passes over groups of samples
    rng = np.random.default_rng(31)
    n_snp, n_acc = 40_000, 513
    db = rand_db(rng, n_snp, n_acc)
    panel = engine.Panel.from_host(ctx, db, packed=True)
    samples = chip_samples(rng, db, 70, 1500)
    off = np.concatenate([[0], np.cumsum([len(r) for r, _ in samples])]).astype(np.int64)
    d_rows = torch.as_tensor(np.concatenate([r for r, _ in samples]), device="cuda:0")
    d_wei = torch.as_tensor(np.concatenate([w for _, w in samples]), device="cuda:0")
    got = engine.score_batch(panel, None, 1000, False, engine.MODE_EXACT, device=(d_rows.data_ptr(), d_wei.data_ptr(), off))
    st = engine.batch_last_stats(ctx)
    assert got["shared_rows"] and st["passes"] >= 2 and st["density"] > 0.3, st
    check_against_oracle(db, samples, got, False)
    # sparse overlap: random markers out of the whole panel -> the automatic choice declines, the forced one still agrees
    sparse = []
    for b in range(16):
        rows = np.sort(rng.choice(n_snp, size=1200, replace=False)).astype(np.int64)
        sparse.append((rows, sample_on(rng, db, rows, 0)))
    off = np.concatenate([[0], np.cumsum([len(r) for r, _ in sparse])]).astype(np.int64)
    d_rows = torch.as_tensor(np.concatenate([r for r, _ in sparse]), device="cuda:0")
    d_wei = torch.as_tensor(np.concatenate([w for _, w in sparse]), device="cuda:0")
    got = engine.score_batch(panel, None, 1000, False, engine.MODE_EXACT, device=(d_rows.data_ptr(), d_wei.data_ptr(), off))
    assert not got["shared_rows"] and engine.batch_last_stats(ctx)["why_not"] == "overlap below the threshold"
    engine.batch_configure(ctx, shared_rows=1)
    forced = engine.score_batch(panel, None, 1000, False, engine.MODE_EXACT, device=(d_rows.data_ptr(), d_wei.data_ptr(), off))
    assert forced["shared_rows"]
    assert np.array_equal(forced["ninfo"], got["ninfo"]) and np.array_equal(forced["score"].astype(np.int64), got["score"].astype(np.int64))
    check_against_oracle(db, sparse, forced, False)
    ctx.close()


def test_coded_batch_through_the_shared_rows():
    """dictionary-coded weights (exp(-PL/10) tables, snpm_score_batch_coded): the digits come from the table's fp64 values"""
    ctx = make_ctx()
    rng = np.random.default_rng(4)
    n_snp, n_acc = 30_000, 1135
    db = rand_db(rng, n_snp, n_acc)
    panel = engine.Panel.from_host(ctx, db, packed=True)
    table = engine.pl_table(7460)
    base = np.sort(rng.choice(n_snp, size=4000, replace=False)).astype(np.int64)
    samples, plain = [], []
    for b in range(20):
        rows = base[rng.random(len(base)) > 0.1]
        pl = rng.integers(0, 256, size=(len(rows), 3)).astype(np.uint16)
        pl[np.arange(len(rows)), rng.integers(0, 3, size=len(rows))] = 0
        hard = rng.random(len(rows)) < 0.2
        pl[hard] = np.where(pl[hard] == 0, 0, 7459).astype(np.uint16)     # hard calls: weights 1 / 0 (exp underflows to 0.0)
        samples.append((rows, pl))
        plain.append((rows, table[pl]))
    engine.batch_configure(ctx, shared_rows=1)
    got = engine.score_batch(panel, samples, 1000, False, engine.MODE_EXACT, table=table)
    assert got["shared_rows"]
    check_against_oracle(db, plain, got, False)
    ctx.close()


def test_parts_on_two_streams_and_the_probe_agree_with_one_stream():
    """the pass cut into parts (digit layout on the auxiliary stream beside the previous part's contraction) returns the bits of
    the one-stream pass, over several passes over groups too; the automatic policy's probe declines scattered marker sets before
    the full pass and lets a batch on one marker set through"""
    import torch
    rng = np.random.default_rng(12)
    n_snp, n_acc = 60_000, 900
    db = rand_db(rng, n_snp, n_acc)
    samples = chip_samples(rng, db, 40, 9000)
    off = np.concatenate([[0], np.cumsum([len(r) for r, _ in samples])]).astype(np.int64)
    res = {}
    for name, env in (("one stream", {"SNPM_SHARED_PARTS": 1, "SNPM_SHARED_TILES": 32}), ("four parts", {"SNPM_SHARED_PARTS": 4, "SNPM_SHARED_TILES": 32}),
                      ("eight parts, several passes", {"SNPM_SHARED_PARTS": 8, "SNPM_SHARED_TILES": 64, "SNPM_SHARED_WS_MB": 4})):
        ctx = make_ctx(**env)
        panel = engine.Panel.from_host(ctx, db, packed=True)
        d_rows = torch.as_tensor(np.concatenate([r for r, _ in samples]), device="cuda:0")
        d_wei = torch.as_tensor(np.concatenate([w for _, w in samples]), device="cuda:0")
        for rep in range(3):                                                 # back to back: a pass must not overwrite digits still being read
            got = engine.score_batch(panel, None, 1000, False, engine.MODE_EXACT, device=(d_rows.data_ptr(), d_wei.data_ptr(), off))
        st = engine.batch_last_stats(ctx)
        assert got["shared_rows"] and st["taken"], (name, st)                # automatic policy: probe, then the full pass
        if "several passes" in name:
            assert st["passes"] >= 2, st
        res[name] = got
        ctx.close()
    check_against_oracle(db, samples, res["one stream"], False)
    for name, got in res.items():
        for k in ("score", "ninfo", "lik", "lrt"):
            assert np.array_equal(np.ascontiguousarray(got[k]).view(np.uint64), np.ascontiguousarray(res["one stream"][k]).view(np.uint64)), (name, k)
    # scattered markers: declined by the probe (the estimate is reported), same results from the per-sample pass
    ctx = make_ctx()
    panel = engine.Panel.from_host(ctx, db, packed=True)
    sparse = []
    for b in range(16):
        rows = np.sort(rng.choice(n_snp, size=3000, replace=False)).astype(np.int64)
        sparse.append((rows, sample_on(rng, db, rows, 0)))
    off = np.concatenate([[0], np.cumsum([len(r) for r, _ in sparse])]).astype(np.int64)
    d_rows = torch.as_tensor(np.concatenate([r for r, _ in sparse]), device="cuda:0")
    d_wei = torch.as_tensor(np.concatenate([w for _, w in sparse]), device="cuda:0")
    got = engine.score_batch(panel, None, 1000, False, engine.MODE_EXACT, device=(d_rows.data_ptr(), d_wei.data_ptr(), off))
    st = engine.batch_last_stats(ctx)
    assert not got["shared_rows"] and st["why_not"] == "overlap below the threshold" and 0.0 < st["density"] < 0.2, st
    check_against_oracle(db, sparse, got, False)
    ctx.close()


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_random_configurations_against_the_oracle(seed):
    """random widths (around the 128-accession wave tiles and the split packed layouts), batch sizes, sample lengths, overlaps,
    digit counts, row-tile counts, digit-matrix budgets, chunk lengths; both formats, both skip_hets settings"""
    rng = np.random.default_rng(9000 + seed)
    for case in range(6):
        n_acc = int(rng.choice([1, 3, 64, 127, 128, 129, 255, 300, 511, 513, 1023, 1135, 1500, 2049, 2600]))
        n_snp = int(rng.integers(2_000, 40_000))
        packed = bool(rng.integers(0, 2))
        skip = bool(rng.integers(0, 2))
        digits = int(rng.choice([-1, 3, 4, 5, 6, 7]))
        # row tiles: fewer than 8 = filler tiles only, 16 = aligned only, 9 / 20 = aligned tiles + fillers spread over the XCDs
        env = {"SNPM_SHARED_TILES": int(rng.choice([1, 2, 3, 7, 9, 16, 20])), "SNPM_SHARED_WS_MB": int(rng.choice([1, 2, 64]))}
        ctx = make_ctx(**env)
        db = rand_db(rng, n_snp, n_acc)
        panel = engine.Panel.from_host(ctx, db, packed=packed)
        n_markers = int(rng.integers(8, max(9, n_snp // 2)))
        B = int(rng.integers(1, 45))
        samples = chip_samples(rng, db, B, n_markers, drop=float(rng.choice([0.0, 0.05, 0.5])), extra=float(rng.choice([0.0, 0.02, 0.3])))
        samples = [(r, w) for r, w in samples]
        if B > 2 and rng.integers(0, 2):
            samples[1] = (np.zeros(0, dtype=np.int64), np.zeros((0, 3)))
        chunk = int(rng.choice([1, 7, 100, 1000, 9000, 20000]))
        engine.batch_configure(ctx, shared_rows=1, digits=digits)
        got = engine.score_batch(panel, samples, chunk, skip, engine.MODE_EXACT)
        st = engine.batch_last_stats(ctx)
        assert got["shared_rows"] and st["taken"], (case, st)
        for b, (rows, wei) in enumerate(samples):
            want_s, want_n = c_oracle.genotyper(db, rows, wei, chunk, skip)
            assert np.array_equal(got["ninfo"][b], want_n), (seed, case, b, n_acc, packed, skip, st)
            assert np.array_equal(got["score"][b].astype(np.int64), want_s.astype(np.int64)), (seed, case, b, n_acc, packed, skip, st)
            wl, wr = orc.calculate_likelihoods(np.array(want_s, dtype=np.int64), want_n)
            np.testing.assert_allclose(got["lik"][b], wl, rtol=LIK_RTOL, equal_nan=True)
        ctx.close()


def test_forced_pairs_and_the_reference_order_fallback_behind_the_contraction():
    """SNPM_DEBUG_REEVAL=k flags accessions 0..k-1 of every sample: the pair re-evaluation behind the shared-row pass returns the
    reference's fp64 bits for them -- in both forms of k_strict_pairs: a wave per (pair, chunk) chain while they are few (3 x 60
    pairs x 3 chunks), a lane per chain from 2048 chains on (chunks of 50 rows; 40 flagged accessions per sample; packed panel and
    skip_hets included); with more flagged pairs than the sparse tier takes (700 x 60 > 32768) every sample goes through the
    reference-order chain -- from the caller's device arrays, which the pass read in place"""
    import torch
    rng = np.random.default_rng(77)
    n_snp, n_acc = 30_000, 800
    db = rand_db(rng, n_snp, n_acc)
    samples = chip_samples(rng, db, 60, 2500)
    off = np.concatenate([[0], np.cumsum([len(r) for r, _ in samples])]).astype(np.int64)
    for reeval, chunk, packed, skip in ((3, 1000, False, False), (3, 50, False, False), (3, 37, True, True), (40, 1000, True, False),
                                        (700, 1000, False, False)):
        ctx = make_ctx(SNPM_DEBUG_REEVAL=reeval)
        panel = engine.Panel.from_host(ctx, db, packed=packed)
        d_rows = torch.as_tensor(np.concatenate([r for r, _ in samples]), device="cuda:0")
        d_wei = torch.as_tensor(np.concatenate([w for _, w in samples]), device="cuda:0")
        got = engine.score_batch(panel, None, chunk, skip, engine.MODE_EXACT, device=(d_rows.data_ptr(), d_wei.data_ptr(), off))
        assert got["shared_rows"] and got["pairs_reeval"] >= reeval * 60
        assert got["strict_fallback"] == (reeval == 700)
        for b, (rows, wei) in enumerate(samples):
            want_s, want_n = c_oracle.genotyper(db, rows, wei, chunk, skip)
            assert np.array_equal(got["ninfo"][b], want_n) and np.array_equal(got["score"][b].astype(np.int64), want_s.astype(np.int64)), b
            k = n_acc if reeval == 700 else reeval                     # these went through the reference-order kernels: its bits
            assert np.array_equal(got["score"][b][:k].view(np.uint64), want_s[:k].view(np.uint64)), (b, reeval, chunk, packed, skip)
        ctx.close()


def test_samples_on_stretches_of_the_union_and_results_above_the_slab_limit():
    """k_sh_pos writes every element of a sample's position row itself (zeros included): samples that cover only the head, the
    middle or the tail of the union leave long zero stretches, a sample of 70k calls takes more than one round of the kernel's 64
    blocks; 150 samples x 1135 accessions are 1.3 MB per result array: above the limit of the one-copy return, each array
    travels by itself."""
    ctx = make_ctx()
    rng = np.random.default_rng(4711)
    n_snp, n_acc = 160_000, 1135
    db = rand_db(rng, n_snp, n_acc)
    panel = engine.Panel.from_host(ctx, db)
    base = np.sort(rng.choice(n_snp, size=70_000, replace=False)).astype(np.int64)
    stretches = [(0, 70_000), (0, 9_000), (30_000, 41_000), (61_000, 70_000), (69_999, 70_000), (0, 1), (255, 257), (16_383, 16_390)]
    samples = []
    for b in range(150):
        lo, hi = stretches[b % len(stretches)]
        rows = base[lo:hi]
        if b % 5 == 4:
            rows = rows[rng.random(len(rows)) >= 0.4]
        if b == 17:
            rows = rows[:0]
        samples.append((rows, sample_on(rng, db, rows, b % 3)))
    engine.batch_configure(ctx, shared_rows=1)
    got = engine.score_batch(panel, samples, 1000, False, engine.MODE_EXACT)
    st = engine.batch_last_stats(ctx)
    assert got["shared_rows"] and st["taken"] and got["union_rows"] == 70_000, st
    assert got["score"].nbytes > (1 << 20)
    check_against_oracle(db, samples, got, False)
    # 64 of them: every array below the limit, one copy behind the status line
    got64 = engine.score_batch(panel, samples[:64], 1000, False, engine.MODE_EXACT)
    assert got64["shared_rows"] and got64["score"].nbytes < (1 << 20)
    for k in ("score", "ninfo", "lik", "lrt"):
        assert np.array_equal(got64[k], got[k][:64], equal_nan=True), k
    # the same return path behind the per-sample pass
    engine.batch_configure(ctx, shared_rows=0)
    seg = engine.score_batch(panel, samples[:64], 1000, False, engine.MODE_EXACT)
    assert not seg["shared_rows"]
    assert np.array_equal(seg["ninfo"], got64["ninfo"]) and np.array_equal(seg["score"].astype(np.int64), got64["score"].astype(np.int64))
    np.testing.assert_allclose(seg["lik"], got64["lik"], rtol=LIK_RTOL, equal_nan=True)
    only_scores = engine.score_batch(panel, samples[:64], 1000, False, engine.MODE_EXACT, likelihoods=False)
    assert np.array_equal(only_scores["score"], seg["score"]) and np.array_equal(only_scores["ninfo"], seg["ninfo"])
    ctx.close()


def test_result_arrays_of_an_earlier_call_are_written_again():
    """engine.score_batch(out=previous): the same arrays come back with the new batch's results (a service reuses its buffers: fresh
    arrays cost their page faults per call); a dict of another shape is ignored"""
    ctx = make_ctx()
    rng = np.random.default_rng(31)
    db = rand_db(rng, 40_000, 600)
    panel = engine.Panel.from_host(ctx, db)
    first = chip_samples(rng, db, 12, 3000)
    second = chip_samples(rng, db, 12, 3500)
    engine.batch_configure(ctx, shared_rows=1)
    a = engine.score_batch(panel, first, 1000, False, engine.MODE_EXACT)
    keep = {k: a[k].copy() for k in ("score", "ninfo", "lik", "lrt")}
    b = engine.score_batch(panel, second, 1000, False, engine.MODE_EXACT, out=a)
    assert all(b[k] is a[k] for k in ("score", "ninfo", "lik", "lrt"))
    check_against_oracle(db, second, b, False)
    assert not np.array_equal(b["score"], keep["score"])
    c = engine.score_batch(panel, first[:5], 1000, False, engine.MODE_EXACT, out=b)          # another shape: new arrays
    assert c["score"].shape == (5, 600) and c["score"] is not b["score"]
    check_against_oracle(db, first[:5], c, False)
    d = engine.score_batch(panel, first, 1000, False, engine.MODE_EXACT, likelihoods=False, out=b)
    assert d["score"] is b["score"] and "lik" not in d and np.array_equal(d["score"], keep["score"]) and np.array_equal(d["ninfo"], keep["ninfo"])
    ctx.close()

