"""
The split layout of 2-bit packed panels (-m gpu): the whole 256-B column blocks of a row in a main matrix, its ragged tail in a
narrow-pitch matrix of its own (snpm_k_common.hpp).  Every reader and writer of a packed panel goes through the same layout
descriptor; these tests walk the widths around every boundary of the rule (tail of 4 ... 128 bytes, no tail, a tail too long to
split, main part empty / one / several column blocks) through all of them -- upload (host-packed and device-packed), download,
synthetic fill, the fast passes (PL weights, hard calls; dense, gathered, windows, batches), the reference-order kernels, the
sparse re-evaluation tiers incl. the accession-major copy, --refine's scan and the in-silico F1s -- against the C oracle and
against the same panel in round 3's whole-row layout.
"""
import os

import numpy as np
import pytest

from oracle import c_oracle
from snpmatch_amd import engine, synth

pytestmark = pytest.mark.gpu

WIDTHS = [1, 5, 16, 17, 64, 100, 128, 129, 500, 512, 513, 1023, 1024, 1025, 1028, 1040, 1135, 1536, 1537, 1540, 2048, 2049, 2100,
          3000, 3073, 4097, 4600]


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def want_split(n_acc):
    row = (n_acc + 3) // 4
    main, rem = row // 256 * 256, row % 256
    tp = 4
    while tp < rem:
        tp *= 2
    return (main, tp) if (0 < rem <= 128 and (256 - tp) * 20 >= main + 256) else None


def rand_db(rng, n, n_acc):
    return rng.choice(np.array([-1, 0, 1, 2], dtype=np.int8), size=(n, n_acc), p=[0.05, 0.60, 0.33, 0.02])


def make_ctx(**env):
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        return engine.Context(0)
    finally:
        for k in env:
            del os.environ[k]


@pytest.mark.parametrize("n_acc", WIDTHS)
def test_split_layout_every_reader_and_writer(n_acc):
    rng = np.random.default_rng(7000 + n_acc)
    n_snp = 2600 if n_acc < 3000 else 1100
    db = rand_db(rng, n_snp, n_acc)
    if n_acc > 6:
        db[:, n_acc - 1] = -1                  # the very last accession (the end of the tail): nothing informative
        db[::3, n_acc - 2] = 2
    ctx = make_ctx(SNPM_ACC_MAJOR_MIN_ROWS=0, SNPM_DEBUG_REEVAL=min(3, n_acc))        # forced re-evaluation of the first accessions
    split = want_split(n_acc)
    pitch = ctx.row_pitch(n_acc, True)
    assert pitch == (sum(split) if split else ((n_acc + 3) // 4 + 255) // 256 * 256)
    panel = engine.Panel.from_host(ctx, db, packed=True)                  # host-packed rows through the loader
    assert panel.pitch == pitch
    assert np.array_equal(panel.download_rows(0, n_snp), db)
    assert np.array_equal(panel.download_rows(n_snp - 7, 7), db[n_snp - 7:])
    os.environ["SNPM_PACKED_SPLIT"] = "0"
    try:
        whole = engine.Panel.from_host(ctx, db, packed=True)                  # the same DB in whole rows
    finally:
        del os.environ["SNPM_PACKED_SPLIT"]
    assert whole.pitch % 256 == 0 and (split is None) == (whole.pitch == panel.pitch)
    rows = np.sort(rng.choice(n_snp, size=n_snp // 2, replace=False)).astype(np.int64)
    codes = rng.choice(np.array([0, 1, 2], dtype=np.int8), size=n_snp, p=[0.6, 0.35, 0.05])
    for frac_pl in (0.8, 0.0):                                            # k_fast_packed_q4 / k_fast_bits
        wei = synth.sample_weights(rng, codes, frac_pl)
        for row_idx, w in ((None, wei), (rows, wei[rows])):
            for skip in (False, True):
                want_s, want_n = c_oracle.genotyper(db, row_idx, w, 1000, skip)
                for p in (panel, whole):
                    q = engine.Query(p, row_idx, w)
                    s, ni = q.run(1000, skip, engine.MODE_STRICT)
                    assert np.array_equal(bits(s), bits(want_s)) and np.array_equal(ni, want_n), (n_acc, frac_pl, skip)
                    s, ni, info = q.run(1000, skip, engine.MODE_EXACT, return_info=True)
                    assert np.array_equal(ni, want_n) and np.array_equal(s.astype(int), want_s.astype(int))
                    if frac_pl > 0:                                       # the forced accessions went through the sparse tier
                        assert info["n_strict_reeval"] >= 1 and np.array_equal(bits(s[:min(3, n_acc)]), bits(want_s[:min(3, n_acc)]))
                    s, ni = q.run(1000, skip, engine.MODE_FAST)
                    assert np.array_equal(ni, want_n) and np.allclose(s, want_s, rtol=0, atol=1e-9)
                    q.free()
        # windows (segmented fast pass + reference order) and a batch of three samples on the split panel
        off = np.array([0, 0, len(rows) // 3, len(rows) // 3 + 1, len(rows)], dtype=np.int64)
        ws, wn, ts, tn = c_oracle.windows(db, rows, wei[rows], off, False)
        q = engine.Query(panel, rows, wei[rows])
        s, ni, s_tot, n_tot = q.run_windows(off, False, totals=True, fast=False)
        assert np.array_equal(bits(s), bits(ws)) and np.array_equal(ni, wn) and np.array_equal(bits(s_tot), bits(ts))
        s, ni, s_tot, n_tot = q.run_windows(off, False, totals=True, fast=True)
        assert np.array_equal(ni, wn) and np.array_equal(s.astype(int), ws.astype(int)) and np.array_equal(n_tot, tn)
        q.free()
        out = engine.score_batch(panel, [(rows[:900], wei[rows[:900]]), (rows, wei[rows]), (rows[3:], wei[rows[3:]])])
        want_s, want_n = c_oracle.genotyper(db, rows, wei[rows], 1000, False)
        assert np.array_equal(out["ninfo"][1], want_n) and np.array_equal(out["score"][1].astype(int), want_s.astype(int))
    # --refine's segregating scan and the in-silico F1s read single calls through the descriptor
    cols = np.unique(rng.choice(n_acc, size=min(n_acc, 5), replace=False)).astype(np.int32)
    if n_acc >= 4 and len(cols) <= n_acc // 2:
        assert np.array_equal(panel.segregating_rows(cols), whole.segregating_rows(cols))
    if n_acc >= 3:
        sel = np.array(sorted({0, n_acc // 2, n_acc - 1}), dtype=np.int32)
        qa, qb = engine.Query(panel, rows, wei[rows]), engine.Query(whole, rows, wei[rows])
        assert np.array_equal(qa.gather_columns(sel), qb.gather_columns(sel))
        fa, fb = qa.f1_pairs(sel), qb.f1_pairs(sel)
        assert np.array_equal(bits(fa[0]), bits(fb[0])) and np.array_equal(fa[1], fb[1])
        qa.free(), qb.free()
    # the synthetic fill and device-side packing (SNPM_HOST_PACK=0) write through the same descriptor
    syn = engine.Panel(ctx, 300, n_acc, packed=True)
    syn.fill_synthetic(99)
    assert np.array_equal(syn.download_rows(0, 300), synth.panel_values(99, 0, 300, 0, n_acc))
    syn.free()
    panel.free(), whole.free()
    ctx.close()
    ctx2 = make_ctx(SNPM_HOST_PACK=0)
    dev_packed = engine.Panel.from_host(ctx2, db, packed=True)
    assert dev_packed.pitch == pitch and np.array_equal(dev_packed.download_rows(0, n_snp), db)
    dev_packed.free()
    ctx2.close()


def test_split_layout_shard_and_streamed_slabs():
    """an accession shard (columns 1024 .. 2159 of a wider DB = 1136 accessions, split rows) and the same columns streamed through
    two half-buffers give the unsharded run's numbers"""
    rng = np.random.default_rng(31)
    n_snp, n_acc = 5200, 2160
    db = rand_db(rng, n_snp, n_acc)
    wei = synth.sample_weights(rng, rng.choice(np.array([0, 1, 2], dtype=np.int8), size=n_snp), 0.8)
    want_s, want_n = c_oracle.genotyper(db, None, wei, 1000, False)
    ctx = engine.Context(0)
    shard = engine.Panel.from_host(ctx, db, packed=True, cols=(1024, 2160))
    assert shard.pitch == 256 + 32
    s, ni = engine.Query(shard, None, wei).run(1000, False, engine.MODE_STRICT)
    assert np.array_equal(bits(s), bits(want_s[1024:])) and np.array_equal(ni, want_n[1024:])
    sp = engine.StreamedPanel(ctx, engine.RowStore(snps=db), cols=(1024, 2160), packed=True, budget_bytes=2 * (2000 + 32) * shard.pitch + 8)
    s2, n2 = sp.query(None, wei).run(1000, False, engine.MODE_EXACT)
    assert np.array_equal(bits(s2), bits(want_s[1024:])) and np.array_equal(n2, want_n[1024:]) and sp.loads == 3
    sp.free(), shard.free()
    ctx.close()
