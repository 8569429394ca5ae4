"""
Slab-streamed scoring (-m gpu): a job whose SNP axis is scored slab after slab with a carry must give what one
query over the whole axis gives -- the reference's fp64 bits in strict mode, exact counts in certified mode
(core/snpmatch.py:218-225 accumulates chunk after chunk over the whole axis) -- plus the device-side certificate
in its three regimes (nothing flagged, a few accessions, more than the sparse tier takes), the device-resident
query constructor and the device sample generator against their numpy twins.
"""
import os

import numpy as np
import pytest

from oracle import c_oracle
from snpmatch_amd import engine, synth

pytestmark = pytest.mark.gpu


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


def slab_scorer(ctx, db, wei, slabs, chunk, skip, packed=False):
    """a panel that holds one slab at a time; load(k) uploads slab k into its first rows"""
    starts = np.concatenate([[0], np.cumsum(slabs)])
    panel = engine.Panel(ctx, max(slabs), db.shape[1], packed=packed)

    def load(k, p):
        p.upload_rows(0, db[starts[k]:starts[k + 1]])

    return engine.SlabScorer(panel, slabs, load, lambda k: wei[starts[k]:starts[k + 1]], chunk=chunk, skip_hets=skip), panel


@pytest.mark.parametrize("n,n_acc,slabs,chunk,packed", [
    (24000, 300, [12000, 12000], 1000, False),
    (25003, 1135, [8000, 10000, 7003], 1000, False),
    (9001, 70, [4200, 4200, 601], 7, False),
    (25003, 1135, [16000, 9003], 1000, True),
])
@pytest.mark.parametrize("skip", [False, True])
def test_slabs_equal_one_pass(n, n_acc, slabs, chunk, packed, skip):
    ctx = make_ctx()
    rng = np.random.default_rng(n + n_acc)
    db = rand_db(rng, n, n_acc)
    codes = db[:, 5].copy()
    codes[codes < 0] = 0
    wei = synth.sample_weights(rng, codes, 0.8)
    want_s, want_n = c_oracle.genotyper(db, None, wei, chunk, skip)
    sc, panel = slab_scorer(ctx, db, wei, slabs, chunk, skip, packed)
    # strict: the chain of additions continues across slabs -> the reference's bits over the whole axis
    s, ni, _ = sc.run(engine.MODE_STRICT)
    assert np.array_equal(bits(s), bits(want_s)) and np.array_equal(ni, want_n)
    # certified: exact counts, fp64 within the summed bound
    s, ni, info = sc.run(engine.MODE_EXACT)
    assert np.array_equal(ni, want_n)
    assert np.array_equal(np.array(s, dtype=np.int64), np.array(want_s, dtype=np.int64))
    assert np.max(np.abs(s - want_s)) < 1e-6
    # one query over the whole axis on a resident panel gives the same counts
    full = engine.Panel.from_host(ctx, db, packed=packed)
    s1, n1 = engine.Query(full, None, wei).run(chunk, skip, engine.MODE_EXACT)
    assert np.array_equal(n1, ni) and np.array_equal(np.array(s1, dtype=np.int64), np.array(s, dtype=np.int64))
    sc.free()
    ctx.close()


def test_slab_second_pass_for_flagged_accessions():
    """accession 7 is a perfect match of PL-weighted calls (its score is an exact integer: the certificate must
    flag it) and accessions 0..2 are forced: the second pass re-scores them in reference order across the slabs"""
    ctx = make_ctx(SNPM_DEBUG_REEVAL=3)
    rng = np.random.default_rng(5)
    n, n_acc = 30000, 200
    db = rand_db(rng, n, n_acc)
    codes = db[:, 7].copy()
    codes[codes < 0] = 0
    wei = synth.sample_weights(rng, codes, frac_pl=1.0)
    want_s, want_n = c_oracle.genotyper(db, None, wei, 1000, False)
    sc, panel = slab_scorer(ctx, db, wei, [11000, 11000, 8000], 1000, False)
    s, ni, info = sc.run(engine.MODE_EXACT)
    assert info["second_pass"] and 4 <= info["n_strict_reeval"] <= 64
    assert np.array_equal(ni, want_n) and np.array_equal(np.array(s, dtype=np.int64), np.array(want_s, dtype=np.int64))
    for a in (0, 1, 2, 7):
        assert bits(s)[a] == bits(want_s)[a]
    sc.free()
    ctx.close()


def test_more_flagged_than_the_sparse_tier_takes():
    """100 identical accessions matching the sample perfectly: 100 exact-integer scores -> the dense tier
    (every accession in reference order) runs, on the device's own decision, in a single query and in a slab job"""
    ctx = make_ctx()
    rng = np.random.default_rng(6)
    n, n_acc = 20000, 300
    db = rand_db(rng, n, n_acc)
    db[:, 100:200] = db[:, [100]]
    codes = db[:, 100].copy()
    codes[codes < 0] = 0
    wei = synth.sample_weights(rng, codes, frac_pl=1.0)
    want_s, want_n = c_oracle.genotyper(db, None, wei, 1000, False)
    panel = engine.Panel.from_host(ctx, db)
    s, ni, info = engine.Query(panel, None, wei).run(1000, False, engine.MODE_EXACT, return_info=True)
    assert info["n_strict_reeval"] >= 100 and info["reeval_path"] == 3
    assert np.array_equal(bits(s), bits(want_s)) and np.array_equal(ni, want_n)
    sc, _ = slab_scorer(ctx, db, wei, [10000, 10000], 1000, False)
    s, ni, info = sc.run(engine.MODE_EXACT)
    assert info["second_pass"] and info["n_strict_reeval"] >= 100
    assert np.array_equal(bits(s), bits(want_s)) and np.array_equal(ni, want_n)
    sc.free()
    ctx.close()


def test_device_sample_and_device_query_match_their_twins():
    import torch
    ctx = make_ctx()
    n_snp, n_acc, planted = 300_000, 64, 17
    panel = engine.Panel(ctx, n_snp, n_acc)
    panel.fill_synthetic(99)
    # refill of a row range: rows 1000.. of the buffer <- SNPs 5000.. of the synthetic panel, the rest untouched
    panel.fill_synthetic(99, snp0=5000, row0=1000, nrows=2000)
    assert np.array_equal(panel.download_rows(1000, 2000), synth.panel_values(99, 5000, 2000, 0, n_acc))
    assert np.array_equal(panel.download_rows(3000, 100), synth.panel_values(99, 3000, 100, 0, n_acc))
    panel.fill_synthetic(99)
    wei_dev = torch.empty((n_snp, 3), dtype=torch.float64, device="cuda:0")
    for frac_pl in (0.8, 0.0):
        ctx.sample_synthetic(99, 0, n_snp, planted, wei_dev.data_ptr(), err=0.02, frac_pl=frac_pl)
        ctx.synchronize()
        twin = synth.sample_weights_twin(99, 0, n_snp, planted, 0.02, frac_pl)
        assert np.array_equal(bits(wei_dev.cpu().numpy()), bits(twin))
        # a slab of it, generated on its own, is that slab of the whole
        part = torch.empty((1000, 3), dtype=torch.float64, device="cuda:0")
        ctx.sample_synthetic(99, 123_000, 1000, planted, part.data_ptr(), err=0.02, frac_pl=frac_pl)
        ctx.synchronize()
        assert np.array_equal(bits(part.cpu().numpy()), bits(twin[123_000:124_000]))
        qd = engine.Query.from_device(panel, None, wei_dev.data_ptr(), n_snp)
        qh = engine.Query(panel, None, twin)
        for mode in (engine.MODE_EXACT, engine.MODE_STRICT):
            a, b = qd.run(1000, False, mode), qh.run(1000, False, mode)
            assert np.array_equal(bits(a[0]), bits(b[0])) and np.array_equal(a[1], b[1])
        assert qd.error_bound(1000) == qh.error_bound(1000)
        s, ni = qd.run(1000, False, engine.MODE_EXACT)
        assert int(np.argmax(s / ni)) == planted
    # gathered rows from device memory
    rows = np.sort(np.random.default_rng(1).choice(n_snp, size=40_000, replace=False)).astype(np.int64)
    rows_dev = torch.as_tensor(rows, device="cuda:0")
    w_sub = torch.as_tensor(twin[rows], device="cuda:0")
    a = engine.Query.from_device(panel, rows_dev.data_ptr(), w_sub.data_ptr(), len(rows)).run(1000, False, engine.MODE_STRICT)
    b = engine.Query(panel, rows, twin[rows]).run(1000, False, engine.MODE_STRICT)
    assert np.array_equal(bits(a[0]), bits(b[0])) and np.array_equal(a[1], b[1])
    ctx.close()


def test_error_bound_on_device_covers_the_host_formula():
    """the device-side bound (k_eref) is the host formula of round 1 evaluated in fp64 and rounded up"""
    ctx = make_ctx()
    rng = np.random.default_rng(3)
    for n, chunk in ((5000, 1000), (12345, 1000), (999, 7), (1, 1000)):
        db = rand_db(rng, n, 40)
        wei = synth.sample_weights(rng, rng.choice(np.array([0, 1, 2], dtype=np.int8), size=n), 0.8)
        q = engine.Query(engine.Panel.from_host(ctx, db), None, wei)
        wmax = np.abs(wei).max(axis=1)
        K = -(-n // chunk)
        u = 2.0 ** -53
        acc = sum(wmax[k * chunk:(k + 1) * chunk].sum() * (min(chunk, n - k * chunk) + 3 + K - k) for k in range(K))
        eref = acc * u / (1 - (chunk + 3 + K) * u)
        got = q.error_bound(chunk)
        assert eref <= got <= eref * 1.001 + wmax.sum() * 1e-11, (n, chunk, eref, got)
    ctx.close()


@pytest.mark.parametrize("order", ["hard_then_pl", "pl_then_hard"])
def test_mixed_job_hard_call_slab_and_pl_slab(order):
    """A job with one slab of hard 0/1 calls and one of PL weights: the hard-call slab is exact on its own, but the
    reference adds its chunk sums onto (or under) a non-integer running total, so its terms belong in the job's bound
    with the chunks still to come (the bound used to leave them out).  The job's bound must cover the
    host formula over BOTH slabs; counts equal the oracle's; forced accessions carry the reference's bits."""
    ctx = make_ctx(SNPM_DEBUG_REEVAL=3)
    rng = np.random.default_rng(11)
    n, n_acc, chunk = 20000, 257, 1000
    db = rand_db(rng, n, n_acc)
    codes = db[:, 9].copy()
    codes[codes < 0] = 0
    hard, pl = synth.sample_weights(rng, codes, frac_pl=0.0), synth.sample_weights(rng, codes, frac_pl=1.0)
    half = n // 2
    wei = np.concatenate([hard[:half], pl[half:]] if order == "hard_then_pl" else [pl[:half], hard[half:]])
    want_s, want_n = c_oracle.genotyper(db, None, wei, chunk, False)
    sc, panel = slab_scorer(ctx, db, wei, [half, n - half], chunk, False)
    s, ni, info = sc.run(engine.MODE_EXACT)
    assert info["second_pass"] and 3 <= info["n_strict_reeval"] <= 64
    assert np.array_equal(ni, want_n) and np.array_equal(s.astype(np.int64), want_s.astype(np.int64))
    assert np.array_equal(bits(s[:3]), bits(want_s[:3]))
    # the bound of the job against the reference-order formula over all 20 chunks (terms of the hard-call slab included)
    wmax = np.abs(wei).max(axis=1)
    K, u = n // chunk, 2.0 ** -53
    eref = sum(wmax[k * chunk:(k + 1) * chunk].sum() * (chunk + 3 + K - k) for k in range(K)) * u
    bound = sc.carry.error_bound()
    assert eref <= bound <= eref + wmax.sum() * 20000 * u, (eref, bound)      # + the fast pass's own term (<= 8192 + 64 + groups additions)
    assert np.max(np.abs(s - want_s)) <= bound
    sc.free()
    # a job of hard calls only is exact in any order: nothing flagged (not even the forced accessions), bound 0
    sc, panel = slab_scorer(ctx, db, hard, [half, n - half], chunk, False)
    s, ni, info = sc.run(engine.MODE_EXACT)
    want_s, want_n = c_oracle.genotyper(db, None, hard, chunk, False)
    assert not info["second_pass"] and sc.carry.error_bound() == 0.0
    assert np.array_equal(bits(s), bits(want_s)) and np.array_equal(ni, want_n)
    sc.free()
    ctx.close()
