"""
snpm_genotype_once (-m gpu): one sample against a resident panel in ONE library call -- the host threads gather the matched
weight rows, the weight properties come from that pass, counts and likelihoods return in one copy.  Everything it returns must
equal the three-call path (Query + run + likelihood) and the C oracle (Genotyper.genotyper, core/snpmatch.py:207-241).
"""
import os

import numpy as np
import pytest

from oracle import c_oracle
from oracle import snpmatch_oracle as orc
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


@pytest.mark.parametrize("packed", [False, True])
@pytest.mark.parametrize("n_snp,n_acc,n_match,n_in", [(20000, 1135, 7545, 10000), (6000, 64, 3001, 3001), (4000, 1, 1000, 1500),
                                                       (130, 17, 129, 200), (300000, 300, 250000, 260000)])
def test_once_equals_three_calls_and_oracle(ctx, packed, n_snp, n_acc, n_match, n_in):
    rng = np.random.default_rng(n_snp + n_acc + int(packed))
    db = rand_db(rng, n_snp, n_acc)
    panel = engine.Panel.from_host(ctx, db, packed=packed)
    rows = np.sort(rng.choice(n_snp, size=n_match, replace=False)).astype(np.int64)
    codes = rng.choice(np.array([0, 1, 2], dtype=np.int8), size=n_in, p=[0.6, 0.35, 0.05])
    for frac_pl in (0.8, 0.0):                                   # PL-weighted and hard-call samples (k_fast_bits on packed panels)
        wei_all = synth.sample_weights(rng, codes, frac_pl)
        sidx = np.sort(rng.choice(n_in, size=n_match, replace=False)).astype(np.int64)
        wei = wei_all[sidx]
        for skip in (False, True):
            want_s, want_n = c_oracle.genotyper(db, rows, wei, 1000, skip)
            q = engine.Query(panel, rows, wei)
            s3, n3 = q.run(1000, skip, engine.MODE_EXACT)
            lik3, lrt3 = ctx.likelihood(s3, n3, truncate=True)
            q.free()
            out = panel.genotype_once(rows, wei_all, sidx, 1000, skip, engine.MODE_EXACT)
            assert np.array_equal(out["ninfo"], want_n) and np.array_equal(out["score"].astype(int), want_s.astype(int))
            assert np.array_equal(bits(out["score"]), bits(s3))              # the same kernels in the same geometry
            assert np.array_equal(bits(out["lik"]), bits(lik3)) and np.array_equal(bits(out["lrt"]), bits(lrt3))
            assert out["all_integer_weights"] == (frac_pl == 0.0)
            lik_o, lrt_o = orc.calculate_likelihoods(want_s.astype(int), want_n)
            ok = ~np.isnan(lik_o)
            assert np.array_equal(np.isnan(out["lik"]), ~ok) and np.allclose(out["lik"][ok], lik_o[ok], rtol=1e-12, atol=0)
            # reference order: the reference's fp64 bits; without sample_idx the weights are taken row for row
            out = panel.genotype_once(rows, wei, None, 1000, skip, engine.MODE_STRICT, likelihoods=False)
            assert np.array_equal(bits(out["score"]), bits(want_s)) and np.array_equal(out["ninfo"], want_n) and "lik" not in out
    panel.free()


def test_once_forced_reevaluation_and_refusals(golden_dir):
    os.environ["SNPM_DEBUG_REEVAL"] = "3"
    try:
        c = engine.Context(0)
    finally:
        del os.environ["SNPM_DEBUG_REEVAL"]
    rng = np.random.default_rng(99)
    db = rand_db(rng, 9000, 300)
    panel = engine.Panel.from_host(c, db)
    rows = np.arange(9000, dtype=np.int64)
    wei = synth.sample_weights(rng, rng.choice(np.array([0, 1, 2], dtype=np.int8), size=9000), 0.8)
    want_s, want_n = c_oracle.genotyper(db, None, wei, 1000, False)
    out = panel.genotype_once(rows, wei)
    assert out["n_strict_reeval"] >= 3 and np.array_equal(bits(out["score"][:3]), bits(want_s[:3]))     # re-scored in reference order
    assert np.array_equal(out["ninfo"], want_n) and np.array_equal(out["score"].astype(int), want_s.astype(int))
    # refusals with the reference's messages
    with pytest.raises(AssertionError, match="outside the panel"):
        panel.genotype_once(np.array([1, 9000], dtype=np.int64), wei[:2])
    with pytest.raises(AssertionError, match="outside the panel"):
        panel.genotype_once(np.array([1, 2], dtype=np.int64), wei[:2], np.array([0, 5], dtype=np.int64))
    with pytest.raises(AssertionError, match="same number of positions"):
        panel.genotype_once(rows[:10], wei[:9])
    bad = wei[:100].copy()
    bad[7, 1] = np.nan
    with pytest.raises(AssertionError, match="finite"):
        panel.genotype_once(rows[:100], bad)
    heavy = wei[:100].copy() * 3.0                               # counts above the informative sites: the reference's assert (:43)
    with pytest.raises(AssertionError, match="greater than n"):
        panel.genotype_once(rows[:100], heavy)
    # an empty match list is a valid (if useless) sample
    out = panel.genotype_once(np.zeros(0, dtype=np.int64), np.zeros((0, 3)))
    assert np.all(out["ninfo"] == 0) and np.all(out["score"] == 0) and np.all(np.isnan(out["lik"]))
    panel.free()
    c.close()


def test_genotyper_uses_the_one_call_path(ctx, golden_dir, tmp_path, monkeypatch):
    """Genotyper on a resident single-GPU panel goes through snpm_genotype_once and writes the reference's files"""
    import json
    from snpmatch_amd.core import parsers, snp_genotype, snpmatch
    toy = np.load(os.path.join(golden_dir, "toy_db.npz"))
    gold = json.load(open(os.path.join(golden_dir, "g2_inbred.json")))
    g = snp_genotype.Genotype.from_arrays(toy["snps"], toy["accs"], toy["positions"], toy["chrs"], toy["regions"])
    calls = []
    real = engine.Panel.genotype_once
    monkeypatch.setattr(engine.Panel, "genotype_once", lambda self, *a, **k: calls.append(1) or real(self, *a, **k))
    inp = parsers.ParseInputs("")
    inp.load_snp_info(toy["s_chrs"], toy["s_pos"], toy["s_gt"], toy["s_wei"], toy["s_dp"])
    out = str(tmp_path / "once")
    gt = snpmatch.Genotyper(inp, g, out, run_genotyper=True)
    assert calls and hasattr(gt.result, "_device_likelihoods")
    want = gold["inbred_skip0"]
    assert open(out + ".matches.json").read() == want["matches.json"]
    got = [l.split("\t") for l in open(out + ".scores.txt").read().strip().split("\n")]
    ref = [l.split("\t") for l in want["scores.txt"].strip().split("\n")]
    for a, b in zip(got, ref):
        assert a[:4] == b[:4] and a[6:] == b[6:]
        assert abs(float(a[4]) - float(b[4])) <= 1e-12 * abs(float(b[4])) and abs(float(a[5]) - float(b[5])) <= 1e-12 * abs(float(b[5]))
    # changing the counts afterwards invalidates the cached likelihoods
    gt.result.scores = gt.result.scores.copy()
    gt.result.scores[0] -= 1
    gt.result.get_likelihoods()
    lik, _ = snpmatch.GenotyperOutput.calculate_likelihoods(gt.result.scores, gt.result.ninfo)
    assert np.array_equal(bits(gt.result.likelis), bits(lik))


def test_coded_one_call_equals_fp64_one_call(ctx, golden_dir, tmp_path):
    """snpm_genotype_once_coded (weights as uint16 codes into a table, rows as 32-bit indices) returns what the fp64 form returns,
    bit for bit; Genotyper takes it for a parsed VCF and writes the same files as with plain weights"""
    import shutil
    from snpmatch_amd.core import parsers, snp_genotype, snpmatch
    rng = np.random.default_rng(77)
    n_snp, n_acc, n_in, n_match = 50000, 1135, 30000, 24000
    db = rand_db(rng, n_snp, n_acc)
    pl = rng.integers(0, 256, size=(n_in, 3)).astype(float)
    pl[rng.random(n_in) < 0.1] = -1.0                      # rows without PL: one-hot
    pl[5, 1] = -1.0                                        # a partly missing triple: exp(0.1), as the reference computes it
    no_pl = np.all(pl == -1, axis=1)
    wei = np.exp(pl / (-10))
    hot = np.zeros((int(no_pl.sum()), 3))
    hot[np.arange(len(hot)), rng.integers(0, 3, len(hot))] = 1.0
    wei[no_pl] = hot
    pair = parsers._weight_codes(pl, no_pl, wei)
    assert pair is not None
    codes, table = pair
    rows = np.sort(rng.choice(n_snp, size=n_match, replace=False)).astype(np.int64)
    sidx = np.sort(rng.choice(n_in, size=n_match, replace=False)).astype(np.int64)
    for packed in (False, True):
        panel = engine.Panel.from_host(ctx, db, packed=packed)
        for skip in (False, True):
            for mode in (engine.MODE_EXACT, engine.MODE_STRICT):
                a = panel.genotype_once(rows, wei, sidx, 1000, skip, mode)
                b = panel.genotype_once(rows, codes, sidx, 1000, skip, mode, table=table)
                for k in ("score", "ninfo", "lik", "lrt"):
                    assert np.array_equal(a[k].view(np.uint64), b[k].view(np.uint64)), (packed, skip, mode, k)
                assert a["all_integer_weights"] == b["all_integer_weights"] is False
        want_s, want_n = c_oracle.genotyper(db, rows, wei[sidx], 1000, False)
        b = panel.genotype_once(rows, codes, sidx, 1000, False, engine.MODE_STRICT, table=table)
        assert np.array_equal(bits(b["score"]), bits(want_s)) and np.array_equal(b["ninfo"], want_n)
        hard = np.where(wei[no_pl] == 1.0, 0, len(table) - 2).astype(np.uint16)         # an all-hard-call sample through the codes
        h = panel.genotype_once(rows[:len(hard)], hard, None, 1000, False, engine.MODE_EXACT, table=table)
        hw, hn = c_oracle.genotyper(db, rows[:len(hard)], wei[no_pl], 1000, False)
        assert h["all_integer_weights"] and np.array_equal(bits(h["score"]), bits(hw)) and np.array_equal(h["ninfo"], hn)
        with pytest.raises(AssertionError, match="outside"):
            panel.genotype_once(rows[:4], np.full((4, 3), len(table), dtype=np.uint16), None, table=table)
        panel.free()
    # the product: Genotyper on the sample VCF (codes from the parser) writes what it writes with plain weights
    toy_vcf = str(tmp_path / "s.vcf.gz")
    shutil.copy(os.path.join(golden_dir, "701_501.filter.vcf.gz"), toy_vcf)
    inp = parsers.ParseInputs(toy_vcf)
    assert inp.weight_codes() is not None
    rng = np.random.default_rng(3)
    order = np.lexsort((inp.pos, inp.chrs))
    chrs, pos = inp.chrs[order], inp.pos[order]
    regions, start = [], 0
    names = []
    for c in np.unique(chrs):
        k = int(np.sum(chrs == c))
        regions.append((start, start + k))
        names.append(c)
        start += k
    snps = rand_db(rng, len(pos), 40)
    g = snp_genotype.Genotype.from_arrays(snps, [str(i) for i in range(40)], pos, names, regions)
    out_a, out_b = str(tmp_path / "coded"), str(tmp_path / "plain")
    snpmatch.Genotyper(inp, g, out_a, run_genotyper=True)
    plain = parsers.ParseInputs("")
    plain.load_snp_info(inp.chrs, inp.pos, inp.gt, inp.wei, inp.dp)
    assert plain.weight_codes() is None
    snpmatch.Genotyper(plain, g, out_b, run_genotyper=True)
    for suf in (".scores.txt", ".matches.json"):
        assert open(out_a + suf).read() == open(out_b + suf).read()


def _ctx_with(**env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        return engine.Context(0)
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


def test_fused_and_unfused_forms_of_the_call_agree():
    """the short form of snpm_genotype_once (k_once_prep + k_once_finish; by default the sample is read in place from the pinned
    slab), the same with the sample sent through the copy engine behind the fill, and the first version's kernels and copies
    return the same bits -- plain and coded,
    with forced sparse-tier and dense-tier re-evaluations, and with a chunk above the fused form's limit"""
    from snpmatch_amd.core import parsers
    rng = np.random.default_rng(2024)
    n_snp, n_acc, n_in, n_match = 40000, 300, 30000, 21001
    db = rand_db(rng, n_snp, n_acc)
    pl = rng.integers(0, 200, size=(n_in, 3)).astype(float)
    pl[rng.random(n_in) < 0.2] = -1.0
    no_pl = np.all(pl == -1, axis=1)
    wei = np.exp(pl / (-10))
    hot = np.zeros((int(no_pl.sum()), 3))
    hot[np.arange(len(hot)), rng.integers(0, 3, len(hot))] = 1.0
    wei[no_pl] = hot
    codes, table = parsers._weight_codes(pl, no_pl, wei)
    rows = np.sort(rng.choice(n_snp, size=n_match, replace=False)).astype(np.int64)
    sidx = np.sort(rng.choice(n_in, size=n_match, replace=False)).astype(np.int64)
    for reeval in (0, 3, 100):                     # 100 > the sparse tier's 64: the deferred dense tier
        forms = {"fused": _ctx_with(SNPM_DEBUG_REEVAL=reeval), "copies": _ctx_with(SNPM_DEBUG_REEVAL=reeval, SNPM_ONCE_ZEROCOPY=0),
                 "in-place": _ctx_with(SNPM_DEBUG_REEVAL=reeval, SNPM_ONCE_ZEROCOPY=1),
                 "unfused": _ctx_with(SNPM_DEBUG_REEVAL=reeval, SNPM_ONCE_FUSED=0),
                 # round 4's launch chain: two reduce launches, chain kernel and finish kernel apart (round 5 fused each pair)
                 "seven-launches": _ctx_with(SNPM_DEBUG_REEVAL=reeval, SNPM_FUSED_REDUCE=0, SNPM_ONCE_TAIL=0)}
        res = {}
        for name, c in forms.items():
            for packed in (False, True):
                panel = engine.Panel.from_host(c, db, packed=packed)
                for chunk in (1000, 333, 5000):
                    for skip in (False, True):
                        a = panel.genotype_once(rows, wei, sidx, chunk, skip, engine.MODE_EXACT)
                        b = panel.genotype_once(rows, codes, sidx, chunk, skip, engine.MODE_EXACT, table=table)
                        for k in ("score", "ninfo", "lik", "lrt"):
                            assert np.array_equal(a[k].view(np.uint64), b[k].view(np.uint64)), (name, packed, chunk, skip, k)
                        assert a["n_strict_reeval"] >= reeval and b["n_strict_reeval"] >= reeval
                        res[(name, packed, chunk, skip)] = a
                if reeval == 100 and not packed:
                    want_s, want_n = c_oracle.genotyper(db, rows, wei[sidx], 1000, False)
                    assert np.array_equal(bits(res[(name, False, 1000, False)]["score"]), bits(want_s))      # every accession in reference order
                    assert np.array_equal(res[(name, False, 1000, False)]["ninfo"], want_n)
                # a bad code / a bad row are refused by every form
                with pytest.raises(AssertionError, match="outside"):
                    panel.genotype_once(rows[:4], np.full((4, 3), len(table), dtype=np.uint16), None, table=table)
                with pytest.raises(AssertionError, match="outside the panel"):
                    panel.genotype_once(np.array([1, n_snp], dtype=np.int64), codes[:2], None, table=table)
                # ... and the call after a refusal is a normal one
                again = panel.genotype_once(rows, codes, sidx, 1000, False, engine.MODE_EXACT, table=table)
                assert np.array_equal(bits(again["score"]), bits(res[(name, packed, 1000, False)]["score"]))
                panel.free()
        for key, a in res.items():
            if key[0] == "fused":
                continue
            f = res[("fused",) + key[1:]]
            for k in ("score", "ninfo", "lik", "lrt"):
                assert np.array_equal(a[k].view(np.uint64), f[k].view(np.uint64)), (key, k)
        for c in forms.values():
            c.close()


def test_code_properties_follow_the_table_across_both_forms(ctx):
    """the per-code property tables of the fused form describe the table of the LAST coded call, whichever form made it (found by
    the fuzzer: a hard-call table, then the same fractional table through the unfused form (chunk above its limit) and the fused
    one -- the second call used to keep the first table's "all integers" verdict and counted 1.0 weights only)"""
    rng = np.random.default_rng(4242)
    n_snp, n_acc, n = 6000, 300, 2700
    db = rand_db(rng, n_snp, n_acc)
    rows = np.sort(rng.choice(n_snp, size=n, replace=False)).astype(np.int64)
    calls = rng.choice(np.array([0, 1, 2], dtype=np.int8), size=n)
    hard = synth.sample_weights(rng, calls, 0.0)
    soft = synth.sample_weights(rng, calls, 1.0)
    for packed in (False, True):
        panel = engine.Panel.from_host(ctx, db, packed=packed)
        for first_chunk in (5000, 1000):                       # unfused / fused form makes the fractional table current
            t_h, inv_h = np.unique(hard.ravel(), return_inverse=True)
            t_s, inv_s = np.unique(soft.ravel(), return_inverse=True)
            a = panel.genotype_once(rows, inv_h.reshape(hard.shape).astype(np.uint16), None, 1000, False, engine.MODE_EXACT, table=t_h)
            assert a["all_integer_weights"]
            panel.genotype_once(rows, inv_s.reshape(soft.shape).astype(np.uint16), None, first_chunk, False, engine.MODE_EXACT, table=t_s)
            b = panel.genotype_once(rows, inv_s.reshape(soft.shape).astype(np.uint16), None, 7, False, engine.MODE_EXACT, table=t_s)
            want_s, want_n = c_oracle.genotyper(db, rows, soft, 7, False)
            assert not b["all_integer_weights"]
            assert np.array_equal(b["ninfo"], want_n) and np.array_equal(b["score"].astype(int), want_s.astype(int))
        panel.free()
