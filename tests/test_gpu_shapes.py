"""
The per-GPU shapes of BASELINE.json configs[3] and configs[4] under test (-m gpu), through size-independent
properties plus the C oracle on a few whole columns:

  * 1252 accessions x 50M SNPs int8 (64 GB): the shard every rank of the 8-GPU run of configs[3] holds
    (5-wave blocks with the resident-block cap, 8 accumulation epochs per part, 50 000 reference chunks);
  * 12 500 accessions x 16M SNPs int8 (200 GB): the per-GPU slab of configs[4] (the accession-major copy does
    not fit beside it, so the re-evaluation takes the SNP-major path).

A third case holds the WHOLE 10 000 x 50M job on one GPU as a 2-bit packed panel.

All contexts run with SNPM_DEBUG_REEVAL=2: accessions 0 and 1 are re-evaluated in reference order in every
certified run, so k_pack_transpose + k_strict_sparse_T (or the strided k_strict_sparse) + k_scan_few + k_patch
execute at full length whether or not the certificate flags anything.  Reference: core/snpmatch.py:207-233.
"""
import os

import numpy as np
import pytest

from oracle import c_oracle
from oracle import snpmatch_oracle as orc
from snpmatch_amd import engine, synth

pytestmark = pytest.mark.gpu

SEED = 10050
BLOCK = 1_250_000


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def twin_quad(n_snp, acc0, c4):
    """columns [c4, c4 + 4) of the shard starting at accession acc0, all n_snp rows, from the numpy twin"""
    return np.concatenate([synth.panel_values(SEED, r0, min(BLOCK, n_snp - r0), acc0 + c4, 4)
                           for r0 in range(0, n_snp, BLOCK)])


def reeval_context():
    os.environ["SNPM_DEBUG_REEVAL"] = "2"
    try:
        return engine.Context(0)
    finally:
        del os.environ["SNPM_DEBUG_REEVAL"]


def check_shape(n_snp, n_acc, acc0, planted, oracle_quads, ninfo_quads, packed=False):
    ctx = reeval_context()
    try:
        panel = engine.Panel(ctx, n_snp, n_acc, packed=packed)
        panel.fill_synthetic(SEED, 0, acc0)
        # (1) all-ones weights: every informative call matches exactly one class -> score == ninfo (fast and
        #     certified mode agree; integer weights need no re-evaluation)
        q = engine.Query(panel, None, np.ones((n_snp, 3)))
        s, ni, info = q.run(1000, False, engine.MODE_EXACT, return_info=True)
        assert info["all_integer_weights"] and info["n_strict_reeval"] == 0
        assert np.array_equal(s, ni.astype(np.float64)) and ni.min() > 0.94 * n_snp
        s2, ni2 = q.run(1000, False, engine.MODE_FAST)
        assert np.array_equal(s2, s) and np.array_equal(ni2, ni)
        q.free()
        # (2) informative counts of whole accession quads against the numpy twin of the generator
        quads = {}
        for c4 in sorted(set(ninfo_quads) | set(oracle_quads) | {planted // 4 * 4}):
            quads[c4] = twin_quad(n_snp, acc0, c4)
        for c4 in ninfo_quads:
            assert np.array_equal(ni[c4:c4 + 4], n_snp - (quads[c4] < 0).sum(axis=0)), c4
        # (3) a hard-call sample planted on one accession matches it perfectly and is the unique top hit
        p4 = planted // 4 * 4
        col = quads[p4][:, planted - p4]
        codes = col.copy()
        codes[codes < 0] = 0
        hard = orc.weights_from_gt_codes(codes)
        s, ni = engine.Query(panel, None, hard).run(1000, False, engine.MODE_EXACT)
        assert s[planted] == ni[planted] and int(np.argmax(s / ni)) == planted
        lik, lrt = ctx.likelihood(s, ni, truncate=True)
        assert lik[planted] == 1.0 and (lrt < 3.841).sum() == 1
        del hard
        # (4) mixed PL weights over the whole SNP axis: certified counts == reference-order counts for every
        #     accession; the forced re-evaluation of accessions 0, 1 carries the reference's bits; whole columns
        #     agree with the C oracle bit for bit
        rng = np.random.default_rng(n_acc)
        _, wpl = synth.planted_sample(rng, col, 0.02)
        q = engine.Query(panel, None, wpl)
        bound = q.error_bound(1000)
        assert 0 < bound < 1e-2
        se, ne, info = q.run(1000, False, engine.MODE_EXACT, return_info=True)
        assert 2 <= info["n_strict_reeval"] < 60
        ss, ns = q.run(1000, False, engine.MODE_STRICT)
        assert np.array_equal(ne, ns)
        assert np.array_equal(np.array(se, dtype=np.int64), np.array(ss, dtype=np.int64))
        assert np.max(np.abs(se - ss)) <= bound
        assert np.array_equal(bits(se[:2]), bits(ss[:2]))              # re-evaluated columns are patched in
        for c4 in oracle_quads:
            ws, wn = c_oracle.genotyper(quads[c4], None, wpl, 1000, False)
            assert np.array_equal(bits(ss[c4:c4 + 4]), bits(ws)) and np.array_equal(ns[c4:c4 + 4], wn), c4
            assert np.array_equal(np.array(se[c4:c4 + 4], dtype=np.int64), np.array(ws, dtype=np.int64)), c4
        assert int(np.argmax(se / ne)) == planted
        # skip_hets: hets of the DB count as missing (core/snpmatch.py:78-79)
        sk, nk = q.run(1000, True, engine.MODE_EXACT)
        c4 = oracle_quads[0]
        ws, wn = c_oracle.genotyper(quads[c4], None, wpl, 1000, True)
        assert np.array_equal(nk[c4:c4 + 4], wn)
        assert np.array_equal(np.array(sk[c4:c4 + 4], dtype=np.int64), np.array(ws, dtype=np.int64))
        q.free()
        panel.free()
    finally:
        ctx.close()


def test_config3_per_gpu_shard_1252_x_50M():
    """rank 0's shard of the 8-GPU run (accessions 0..1251 of 10 000, all 50M SNPs)"""
    check_shape(50_000_000, 1252, 0, planted=417, oracle_quads=[0, 416], ninfo_quads=[0, 1248])


def test_config3_last_rank_shard_1236_x_50M():
    """rank 7's shard: 10 000 - 7 * 1252 = 1236 accessions starting at accession 8764 (uneven shards)"""
    check_shape(50_000_000, 1236, 8764, planted=1001, oracle_quads=[1000], ninfo_quads=[1232])


def test_config4_per_gpu_slab_12500_x_16M():
    """the per-GPU slab of configs[4] (100k x 100M over 8 GPUs: 12 500 accessions x <= 16M SNPs, 200 GB)"""
    check_shape(16_000_000, 12_500, 0, planted=417, oracle_quads=[0, 416], ninfo_quads=[0, 12_496])


def test_config3_whole_job_on_one_gpu_packed_10000_x_50M():
    """the whole configs[3] job resident on ONE GPU as a 2-bit packed panel (125 GB, plus its accession-major copy): the
    four-rows-per-lookup fast pass (k_fast_packed_q4) with its certificate, the bit-parallel hard-call pass (k_fast_bits),
    the table-driven reference-order kernel (k_strict4 on packed rows) over all 50M SNPs, forced re-evaluations"""
    check_shape(50_000_000, 10_000, 0, planted=417, oracle_quads=[416], ninfo_quads=[0, 9996], packed=True)


def test_slab_streamed_job_equals_resident_10000_x_12M():
    """bench-shaped slab streaming at scale: 10 000 accessions x 12M SNPs scored as two 6M-SNP slabs regenerated on the
    device (carry, certificate over the totals, second pass for the flagged and the two forced accessions) against the
    same panel resident in one piece: certified counts equal, strict totals bit-identical (the chain continues across
    slabs), three slabs of uneven size as well; device-generated sample weights."""
    import torch
    ctx = reeval_context()
    try:
        n_snp, n_acc, planted = 12_000_000, 10_000, 417
        wei = torch.empty((n_snp, 3), dtype=torch.float64, device="cuda:0")
        ctx.sample_synthetic(SEED, 0, n_snp, planted, wei.data_ptr())
        resident = engine.Panel(ctx, n_snp, n_acc)
        resident.fill_synthetic(SEED)
        q = engine.Query.from_device(resident, None, wei.data_ptr(), n_snp)
        se, ne = q.run(1000, False, engine.MODE_EXACT)
        ss, ns = q.run(1000, False, engine.MODE_STRICT)
        assert np.array_equal(ne, ns) and np.array_equal(se.astype(np.int64), ss.astype(np.int64))
        q.free()
        for slabs in ([6_000_000, 6_000_000], [5_000_000, 5_000_000, 2_000_000]):
            starts = np.concatenate([[0], np.cumsum(slabs)])
            buf = engine.Panel(ctx, max(slabs), n_acc)
            sc = engine.SlabScorer(buf, slabs, lambda k, p: p.fill_synthetic(SEED, snp0=int(starts[k]), row0=0, nrows=slabs[k]),
                                   lambda k: wei[int(starts[k]):].data_ptr(), device_weights=True)
            s1, n1, info = sc.run(engine.MODE_EXACT)
            assert info["second_pass"] and 2 <= info["n_strict_reeval"] <= 64
            assert np.array_equal(n1, ne) and np.array_equal(s1.astype(np.int64), se.astype(np.int64))
            assert np.array_equal(bits(s1[:2]), bits(ss[:2]))                  # re-scored in reference order across the slabs
            s2, n2, _ = sc.run(engine.MODE_STRICT)
            assert np.array_equal(bits(s2), bits(ss)) and np.array_equal(n2, ns)
            sc.free()
            buf.free()
        assert int(np.argmax(se / ne)) == planted
    finally:
        ctx.close()


def synthetic_slab_scorer(ctx, n_acc, slabs, wei_ptr_of, packed=False, acc0=0, buf=None):
    """SlabScorer over device-regenerated slabs of the synthetic panel SEED: one resident buffer of max(slabs) rows"""
    starts = np.concatenate([[0], np.cumsum(slabs)]).astype(np.int64)
    buf = buf or engine.Panel(ctx, max(slabs), n_acc, packed=packed)
    sc = engine.SlabScorer(buf, slabs, lambda k, p: p.fill_synthetic(SEED, snp0=int(starts[k]), acc0=acc0, row0=0, nrows=slabs[k]),
                           lambda k: wei_ptr_of(int(starts[k])), device_weights=True)
    return sc, buf


def test_config4_looped_job_12500_x_100M_in_seven_slabs():
    """configs[4] as SURVEY 8d describes it -- the per-GPU share of 100k x 100M (12 500 accessions x 100M SNPs = 1.25 TB
    of int8) looped through ONE resident buffer as seven device-regenerated slabs of <= 16M SNPs with a carry over
    100 000 reference chunks (core/snpmatch.py:218-225 adds every chunk onto ScoreList / NumInfoSites):
      * all-ones weights: score == ninfo for every accession, ninfo of two quads == the numpy twin over all 100M rows;
      * PL-weighted planted sample: EXACT counts == STRICT counts for every accession (certificate over the totals,
        second pass over the slabs for the flagged ones), one quad of the STRICT carry bit for bit == the C oracle
        over all 100M rows, the planted accession is the top hit;
      * with 70 forced accessions the certificate flags more than the sparse tier takes: the job is re-run in
        reference order for every accession (the > 64 branch) and returns the strict bits."""
    import torch
    n_snp, n_acc, planted, chunk = 100_000_000, 12_500, 417, 1000
    slabs = [16_000_000] * 6 + [4_000_000]
    ctx = reeval_context()              # accessions 0, 1 forced into every second pass
    try:
        wei = torch.ones((n_snp, 3), dtype=torch.float64, device="cuda:0")
        sc, buf = synthetic_slab_scorer(ctx, n_acc, slabs, lambda r0: wei[r0:].data_ptr())
        s, ni, info = sc.run(engine.MODE_EXACT)
        assert not info["second_pass"] and info["n_strict_reeval"] == 0          # integer weights: any order is exact
        assert np.array_equal(s, ni.astype(np.float64)) and ni.min() > 0.94 * n_snp
        sc.free()
        quads = {c4: twin_quad(n_snp, 0, c4) for c4 in (416, 12_496)}
        for c4, cols in quads.items():
            assert np.array_equal(ni[c4:c4 + 4], n_snp - (cols < 0).sum(axis=0)), c4
        del quads[12_496]
        # the bench sample (planted accession 417, 2 % error, 80 % PL weights), generated on the device
        ctx.sample_synthetic(SEED, 0, n_snp, planted, wei.data_ptr())
        ctx.synchronize()
        sc, _ = synthetic_slab_scorer(ctx, n_acc, slabs, lambda r0: wei[r0:].data_ptr(), buf=buf)
        se, ne, info = sc.run(engine.MODE_EXACT)
        n_flagged = info["n_strict_reeval"]
        assert info["second_pass"] and 2 <= n_flagged <= 64, info
        ss, ns, _ = sc.run(engine.MODE_STRICT)
        assert np.array_equal(ne, ns) and np.array_equal(ne, ni)
        assert np.array_equal(se.astype(np.int64), ss.astype(np.int64))
        assert np.array_equal(bits(se[:2]), bits(ss[:2]))              # the forced accessions carry the reference's bits
        assert np.max(np.abs(se - ss)) < 1e-2
        assert int(np.argmax(se / ne)) == planted
        wei_host = wei.cpu().numpy()
        ws, wn = c_oracle.genotyper(quads[416], None, wei_host, chunk, False)
        assert np.array_equal(bits(ss[416:420]), bits(ws)) and np.array_equal(ns[416:420], wn)
        assert np.array_equal(se[416:420].astype(np.int64), ws.astype(np.int64))
        del wei_host
        sc.free()
        buf.free()
    finally:
        ctx.close()
    # more flagged totals than the sparse tier takes (here: forced) -> second pass in reference order for everyone
    os.environ["SNPM_DEBUG_REEVAL"] = "70"
    try:
        ctx = engine.Context(0)
    finally:
        del os.environ["SNPM_DEBUG_REEVAL"]
    try:
        sc, buf = synthetic_slab_scorer(ctx, n_acc, slabs, lambda r0: wei[r0:].data_ptr())
        s70, n70, info = sc.run(engine.MODE_EXACT)
        assert info["second_pass"] and info["n_strict_reeval"] > 64, info
        assert np.array_equal(bits(s70), bits(ss)) and np.array_equal(n70, ns)
        sc.free()
        buf.free()
    finally:
        ctx.close()
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "config5_looped_test_record.json"), "w") as fh:
            import json
            json.dump({"shape": "12500 x 100M in slabs %s" % slabs, "flagged_totals_incl_2_forced": int(n_flagged),
                       "max_abs_exact_minus_strict": float(np.max(np.abs(se - ss)))}, fh)


def test_bench_job_int8_slabs_equal_packed_resident_10000_x_50M():
    """the exact N=1 job of bench.py (10 000 x 50M int8 as slabs of 20.019M + 20.019M + 9.962M rows with a carry) against
    the same job on ONE resident 2-bit packed panel: every fp64 total bit-identical in reference order, certified
    counts identical in both (two panel formats, two kernels families, slab carry vs one pass)"""
    import torch
    n_snp, n_acc, planted = 50_000_000, 10_000, 417
    slabs = [20_019_000, 20_019_000, 9_962_000]
    ctx = reeval_context()
    try:
        wei = torch.empty((n_snp, 3), dtype=torch.float64, device="cuda:0")
        ctx.sample_synthetic(SEED, 0, n_snp, planted, wei.data_ptr())
        sc, buf = synthetic_slab_scorer(ctx, n_acc, slabs, lambda r0: wei[r0:].data_ptr())
        ss, ns, _ = sc.run(engine.MODE_STRICT)
        se, ne, info = sc.run(engine.MODE_EXACT)
        assert info["second_pass"] and 2 <= info["n_strict_reeval"] <= 64
        sc.free()
        buf.free()
        packed = engine.Panel(ctx, n_snp, n_acc, packed=True)
        packed.fill_synthetic(SEED)
        q = engine.Query.from_device(packed, None, wei.data_ptr(), n_snp)
        ps, pn = q.run(1000, False, engine.MODE_STRICT)
        pe, pne = q.run(1000, False, engine.MODE_EXACT)
        assert np.array_equal(bits(ps), bits(ss)) and np.array_equal(pn, ns)
        assert np.array_equal(pne, ns) and np.array_equal(ne, ns)
        assert np.array_equal(pe.astype(np.int64), ss.astype(np.int64))
        assert np.array_equal(se.astype(np.int64), ss.astype(np.int64))
        assert int(np.argmax(ss / ns)) == planted
        q.free()
        packed.free()
    finally:
        ctx.close()
