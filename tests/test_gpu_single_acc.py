"""
Panels of ONE accession (-m gpu).  numpy reduces the reference's [1, n] product along a contiguous axis -- pairwise inside
8192-element buffer pieces -- where every wider panel is summed row after row (core/snpmatch.py:85-87).  The reference-order
kernels switch to k_strict_single for such panels; these tests pin that path to the goldens the unmodified reference produced
(tests/golden/g1b_single_acc.npz, g2b_g5b_single_acc.npz) and to the C oracle on shapes the goldens do not hold.
"""
import json
import os

import numpy as np
import pytest

from oracle import c_oracle
from snpmatch_amd import engine, synth

pytestmark = pytest.mark.gpu


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


def test_matchGTsAccs_single_accession_goldens(ctx, golden_dir):
    """the literal matchGTsAccs entry point (snpm_score_dense_host) on the reference's one-accession vectors: fp64 bits"""
    g = np.load(os.path.join(golden_dir, "g1b_single_acc.npz"))
    for name in g["names"]:
        key0, skip = name[:-2], name.endswith("_1")
        s, n = ctx.score_dense(g[key0 + "_wei"], g[key0 + "_db"], skip)
        assert np.array_equal(bits(s), bits(g[name + "_score"])), name
        assert np.array_equal(n, g[name + "_ninfo"]), name
    g = np.load(os.path.join(golden_dir, "g2b_g5b_single_acc.npz"))
    for skip in (0, 1):                    # one call longer than numpy's 8192-element buffer
        s, n = ctx.score_dense(g["long_wei"], g["long_db"], bool(skip))
        assert np.array_equal(bits(s), bits(g["long_score_skip%d" % skip]))
        assert np.array_equal(n, g["long_ninfo_skip%d" % skip])


@pytest.mark.parametrize("packed", [False, True])
def test_genotyper_and_windows_single_accession_goldens(ctx, golden_dir, packed):
    """Genotyper's chunk loop and the cross windows on the one-accession toy DB: the reference's accumulators bit for bit in
    reference-order mode, its counts in the default mode, with the re-evaluation tiers forced through the same kernel"""
    toy = np.load(os.path.join(golden_dir, "toy_db_single.npz"))
    g = np.load(os.path.join(golden_dir, "g2b_g5b_single_acc.npz"))
    panel = engine.Panel.from_host(ctx, toy["snps"], packed=packed)
    c0, c1 = g["common_db"], g["common_sample"]
    q = engine.Query(panel, c0, toy["s_wei"][c1])
    for skip in (0, 1):
        want_s, want_n = g["score_skip%d" % skip], g["ninfo_skip%d" % skip]
        s, n = q.run(1000, bool(skip), engine.MODE_STRICT)
        assert np.array_equal(bits(s), bits(want_s)) and np.array_equal(n, want_n)
        s, n = q.run(1000, bool(skip), engine.MODE_EXACT)
        assert np.array_equal(n, want_n) and np.array_equal(s.astype(int), want_s.astype(int))
    q.free()
    for skip in (0, 1):
        off = g["win_off_skip%d" % skip]
        qw = engine.Query(panel, g["win_rows_db_skip%d" % skip], toy["s_wei"][g["win_rows_sample_skip%d" % skip]])
        for fast in (False, True):
            out = qw.run_windows(off, bool(skip), totals=True, fast=fast)
            ws, wn = out[0], out[1]
            assert np.array_equal(wn, g["win_ninfo_skip%d" % skip])
            if not fast:
                assert np.array_equal(bits(ws), bits(g["win_score_skip%d" % skip]))
            else:
                assert np.array_equal(ws.astype(int), g["win_score_skip%d" % skip].astype(int))
                assert np.allclose(ws, g["win_score_skip%d" % skip], rtol=0, atol=1e-9)
        qw.free()
    panel.free()


def same_table(got_text, want_text, float_cols=(4, 5), rtol=1e-12):
    """scores.txt: every column identical as text except likelihood / lrt (1e-12 relative; north_star allows 1e-6)"""
    got = [l.split("\t") for l in got_text.strip().split("\n")]
    want = [l.split("\t") for l in want_text.strip().split("\n")]
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert len(g) == len(w)
        for c in range(len(w)):
            if c in float_cols:
                a, b = float(g[c]), float(w[c])
                assert (np.isnan(a) and np.isnan(b)) or abs(a - b) <= rtol * abs(b), (g, w)
            else:
                assert g[c] == w[c], (c, g, w)


def test_product_files_single_accession(ctx, golden_dir, tmp_path):
    """`snpmatch inbred` / `cross` on a one-accession DB write the reference's files"""
    from snpmatch_amd.core import csmatch, parsers, snp_genotype, snpmatch
    toy = np.load(os.path.join(golden_dir, "toy_db_single.npz"))
    files = json.load(open(os.path.join(golden_dir, "g2b_g5b_single_acc.json")))
    g = snp_genotype.Genotype.from_arrays(toy["snps"], toy["accs"], toy["positions"], toy["chrs"], toy["regions"])
    for skip in (0, 1):
        inputs = parsers.ParseInputs("")
        inputs.load_snp_info(toy["s_chrs"], toy["s_pos"], toy["s_gt"], toy["s_wei"], toy["s_dp"])
        outp = str(tmp_path / ("inbred%d" % skip))
        snpmatch.Genotyper(inputs, g, outp, run_genotyper=True, skip_db_hets=bool(skip))
        want = files["inbred_skip%d" % skip]
        same_table(open(outp + ".scores.txt").read(), want["scores.txt"])
        assert open(outp + ".matches.json").read() == want["matches.json"]
        outc = str(tmp_path / ("cross%d" % skip))
        csmatch.CrossIdentifier(inputs, g, "athaliana_tair10", 300000, outc, run_identifier=True, skip_db_hets=bool(skip))
        for suf, text in files["cross_skip%d" % skip].items():
            if suf.endswith(".json") or suf == ".windowscore.txt":      # (the window table of a one-accession DB is its header)
                assert open(outc + suf).read() == text, suf
            else:
                same_table(open(outc + suf).read(), text)


def test_single_accession_reevaluation_tiers_slabs_and_batches():
    """every route into the reference-order kernels on a one-accession panel, against the C oracle (itself pinned to the
    reference's one-accession goldens): forced sparse re-evaluation, the slab carry, batched samples with a forced pair
    re-evaluation, a shard of a WIDER panel (which must keep the row-after-row order)"""
    os.environ["SNPM_DEBUG_REEVAL"] = "1"
    try:
        c = engine.Context(0)
    finally:
        del os.environ["SNPM_DEBUG_REEVAL"]
    rng = np.random.default_rng(4242)
    for n, chunk in ((1, 1000), (13, 7), (9000, 1000), (20011, 1001), (30000, 30000)):
        db = rand_db(rng, n, 1)
        wei = rand_wei(rng, n)
        for packed in (False, True):
            panel = engine.Panel.from_host(c, db, packed=packed)
            q = engine.Query(panel, None, wei)
            for skip in (False, True):
                want_s, want_n = c_oracle.genotyper(db, None, wei, chunk, skip)
                s, ni, info = q.run(chunk, skip, engine.MODE_EXACT, return_info=True)
                assert info["n_strict_reeval"] >= 1                       # the forced accession went through the sparse tier
                assert np.array_equal(bits(s), bits(want_s)) and np.array_equal(ni, want_n), (n, chunk, packed, skip)
                s, ni = q.run(chunk, skip, engine.MODE_STRICT)
                assert np.array_equal(bits(s), bits(want_s)) and np.array_equal(ni, want_n)
            q.free()
            if n >= 9000 and chunk == 1000:
                # two SNP slabs with a carry: the chain of chunk totals continues across the slab boundary
                half = 4000
                for mode in (engine.MODE_STRICT, engine.MODE_EXACT):
                    carry = engine.Carry(c, 1)
                    q1, q2 = engine.Query(panel, None, wei[:half]), engine.Query(panel, None, wei[half:], row0=half)
                    q1.run_carry(carry, 1000, False, mode, -(-(n - half) // 1000))
                    q2.run_carry(carry, 1000, False, mode, 0)
                    s3, n3, flagged = carry.finish()
                    want_s, want_n = c_oracle.genotyper(db, None, wei, 1000, False)
                    assert np.array_equal(n3, want_n) and np.array_equal(s3.astype(int), want_s.astype(int))
                    if mode == engine.MODE_STRICT:
                        assert np.array_equal(bits(s3), bits(want_s))
                    q1.free(), q2.free(), carry.free()
                # batched samples: sample = segment, chunks of 1000 inside it, the forced (sample, accession) pairs re-scored
                samples = []
                for k in range(3):
                    rows = np.sort(rng.choice(n, size=2500 + 17 * k, replace=False)).astype(np.int64)
                    samples.append((rows, rand_wei(rng, len(rows))))
                out = engine.score_batch(panel, samples)
                for k, (rows, w) in enumerate(samples):
                    want_s, want_n = c_oracle.genotyper(db, rows, w, 1000, False)
                    assert np.array_equal(out["ninfo"][k], want_n)
                    assert np.array_equal(bits(out["score"][k]), bits(want_s)), k
            panel.free()
    # a one-column SHARD of a five-accession panel is summed row after row, like the panel it belongs to
    db5 = rand_db(rng, 5000, 5)
    wei = rand_wei(rng, 5000)
    want_s, want_n = c_oracle.genotyper(db5, None, wei, 1000, False)
    shard = engine.Panel.from_host(c, db5, cols=(4, 5))
    s, ni = engine.Query(shard, None, wei).run(1000, False, engine.MODE_STRICT)
    assert np.array_equal(bits(s), bits(want_s[4:5])) and np.array_equal(ni, want_n[4:5])
    alone_s, _ = c_oracle.genotyper(db5[:, 4:5], None, wei, 1000, False)
    lone = engine.Panel.from_host(c, np.ascontiguousarray(db5[:, 4:5]))
    s1, _ = engine.Query(lone, None, wei).run(1000, False, engine.MODE_STRICT)
    assert np.array_equal(bits(s1), bits(alone_s))
    shard.free(), lone.free()
    c.close()
