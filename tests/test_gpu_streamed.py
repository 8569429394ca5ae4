"""
DBs that do not fit the HBM budget, and the loader that feeds them (-m gpu).

The reference reads any DB size through ``g.g.snps[idx, :]`` (core/snpmatch.py:218-225, pygwas/genotype.py:548-550).
Here ``Genotype.panel()`` plans the residency by the HBM budget: int8 whole -> 2-bit packed whole -> SNP slabs streamed
through two half-buffers (``engine.StreamedPanel``); SNPM_HBM_BUDGET_GB forces the last on toy DBs.  Streamed runs score
every piece in the reference's order, so their fp64 results are bit-identical with one resident pass and with the oracle.
Also here: snpm_panel_load_file_rows (row lists, column ranges, O_DIRECT), host-side 2-bit packing against the device
packer, and uploads into one panel while another one is being scored.
"""
import json
import os

import numpy as np
import pytest

from oracle import c_oracle
from snpmatch_amd import engine, synth
from snpmatch_amd.core import csmatch, snp_genotype, snpmatch

from test_gpu_pipeline import cmp_scores_table, cmp_window_table, make_inputs

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


def row_pitch(n_acc, packed=False):
    """the library's bytes per row (snpm_panel_row_pitch): 256-B padding, 128 B for int8 rows where that saves 5 % or more; packed
    panels: the split layout (whole 256-B column blocks + the ragged tail at a power-of-two pitch) where that saves 5 % or more"""
    if packed:
        row = (n_acc + 3) // 4
        main, rem = row // 256 * 256, row % 256
        if 0 < rem <= 128 and os.environ.get("SNPM_PACKED_SPLIT", "1") != "0":
            tp = 4
            while tp < rem:
                tp *= 2
            if (256 - tp) * 20 >= main + 256:
                return main + tp
        return (row + 255) // 256 * 256
    p256, p128 = (n_acc + 255) // 256 * 256, (n_acc + 127) // 128 * 128
    return p128 if (p256 - p128) * 20 >= p256 else p256


def budget_for(rows_cap, n_acc, packed=False):
    return 2 * (rows_cap + 32) * row_pitch(n_acc, packed) + 8      # (+ 8: the budget travels as a decimal number of GB)


@pytest.fixture(scope="module")
def case(tmp_path_factory):
    rng = np.random.default_rng(77)
    n, n_acc = 30_000, 300
    db = rand_db(rng, n, n_acc)
    codes = db[:, 5].copy()
    codes[codes < 0] = 0
    wei = synth.sample_weights(rng, codes, 0.8)
    path = str(tmp_path_factory.mktemp("db") / "snps.npy")
    np.save(path, db)
    return db, wei, path


@pytest.mark.parametrize("source", ["file", "array"])
@pytest.mark.parametrize("packed", [False, True])
def test_streamed_panel_equals_resident(case, source, packed):
    db, wei, path = case
    n, n_acc = db.shape
    ctx = make_ctx()
    store = engine.RowStore(npy=path) if source == "file" else engine.RowStore(snps=db)
    cols = (40, 297) if source == "file" else None           # a column range of the file: an accession shard
    a0, a1 = cols or (0, n_acc)
    sub = np.ascontiguousarray(db[:, a0:a1])
    whole = engine.Panel.from_host(ctx, sub, packed=packed)
    sp = engine.StreamedPanel(ctx, store, cols=cols, packed=packed, budget_bytes=budget_for(4000, a1 - a0, packed))
    assert sp.rows_cap == 4000
    # dense: all 30 000 rows in 8 pieces, chain of 30 reference chunks across them
    want_s, want_n = c_oracle.genotyper(sub, None, wei, 1000, False)
    s, ni, info = sp.query(None, wei).run(1000, False, engine.MODE_EXACT, return_info=True)
    assert info["pieces"] == 8 and np.array_equal(bits(s), bits(want_s)) and np.array_equal(ni, want_n)
    # a sample's matched rows (sparse list), skip_hets, odd chunk size
    rows = np.sort(np.random.default_rng(3).choice(n, size=9001, replace=False)).astype(np.int64)
    want_s, want_n = c_oracle.genotyper(sub, rows, wei[rows], 700, True)
    s, ni = sp.query(rows, wei[rows]).run(700, True, engine.MODE_EXACT)
    assert np.array_equal(bits(s), bits(want_s)) and np.array_equal(ni, want_n)
    # windows: pieces hold whole windows, totals chained across pieces; empty windows included
    off = np.concatenate([[0, 0], np.sort(np.random.default_rng(4).choice(9001, size=40, replace=False)), [9001]]).astype(np.int64)
    want = engine.Query(whole, rows, wei[rows]).run_windows(off)
    got = sp.query(rows, wei[rows]).run_windows(off)
    for a, b in zip(got, want):
        assert np.array_equal(np.ascontiguousarray(a).view(np.uint64), np.ascontiguousarray(b).view(np.uint64))
    # column reads, in-silico crosses, --refine scan
    best = np.array([5, 0, 200, 17, 256])
    assert np.array_equal(sp.query(rows, wei[rows]).gather_columns(best), engine.Query(whole, rows, wei[rows]).gather_columns(best))
    fs, fn = sp.query(rows, wei[rows]).f1_pairs(best)
    ws, wn = engine.Query(whole, rows, wei[rows]).f1_pairs(best)
    assert np.array_equal(bits(fs), bits(ws)) and np.array_equal(fn, wn)
    assert np.array_equal(sp.segregating_rows(best), whole.segregating_rows(best))
    assert sp.loads > 20
    sp.free()
    whole.free()
    ctx.close()


def test_loader_row_lists_column_ranges_and_host_packing(case, tmp_path):
    """snpm_panel_load_file_rows against numpy slicing; host-packed rows (AVX2 / scalar) against the device packer"""
    db, wei, path = case
    n, n_acc = db.shape
    store = engine.RowStore(npy=path)
    rows = np.sort(np.random.default_rng(8).choice(n, size=5000, replace=False)).astype(np.int64)
    for env in ({}, {"SNPM_HOST_PACK": 0}, {"SNPM_NO_AVX2": 1, "SNPM_STAGE_MB": 1}, {"SNPM_ODIRECT": 1, "SNPM_STAGE_THREADS": 3}):
        ctx = make_ctx(**env)
        for packed in (False, True):
            p = engine.Panel(ctx, 6000, 130, packed=packed)
            store.load(p, (100, 230), rows, 7)                                # row list, column range, row offset
            assert np.array_equal(p.download_rows(7, 5000), db[rows, 100:230])
            store.load(p, (100, 230), (20_000, 6000), 0)                      # a contiguous range of a wider matrix
            assert np.array_equal(p.download_rows(0, 6000), db[20_000:26_000, 100:230])
            p.free()
            p = engine.Panel.from_npy(ctx, path, packed=packed)               # contiguous full rows (O_DIRECT when asked for)
            assert np.array_equal(p.download_rows(0, n), db)
            p.upload_rows(11, db[rows[:100]])
            assert np.array_equal(p.download_rows(11, 100), db[rows[:100]])
            p.free()
        ctx.close()
    # what a packed panel cannot hold is refused by the host packer as by the device packer; the int8 panel maps it to "other"
    ctx = make_ctx()
    odd = db[:64].copy()
    odd[17, 33] = 5
    with pytest.raises(AssertionError):
        engine.Panel.from_host(ctx, odd, packed=True)
    assert engine.Panel.from_host(ctx, odd).download_rows(17, 1)[0, 33] == 3
    # a file that ends early
    short = str(tmp_path / "short.npy")
    np.save(short, db[:100])
    p = engine.Panel(ctx, 200, n_acc)
    with pytest.raises(AssertionError):
        p.load_file_rows(short, 128, n_acc, 0, None, 0, 0, 200)
    with pytest.raises(AssertionError):
        p.load_file_rows(short, 128, n_acc, 0, np.array([5, 100]), 0, 0, 2)
    ctx.close()


def test_upload_into_one_panel_while_another_is_scored():
    """two resident buffers alternate under a stream of async scoring calls: an upload waits for the work queued on ITS panel
    (not for the other buffer's), and every scoring call sees exactly the rows that were loaded for it"""
    ctx = make_ctx()
    rng = np.random.default_rng(5)
    n, n_acc, k_pieces = 200_000, 1135, 6
    pieces = [rand_db(rng, n, n_acc) for _ in range(2)]         # two distinct contents, re-uploaded alternately
    wei = synth.sample_weights(rng, rng.choice(np.array([0, 1, 2], dtype=np.int8), size=n), 0.8)
    bufs = [engine.Panel(ctx, n, n_acc), engine.Panel(ctx, n, n_acc)]
    want = [engine.Query(engine.Panel.from_host(ctx, pc), None, wei).run(1000, False, engine.MODE_STRICT) for pc in pieces]
    carry = [engine.Carry(ctx, n_acc) for _ in range(k_pieces)]
    bufs[0].upload_rows(0, pieces[0])
    qs = []
    for k in range(k_pieces):
        q = engine.Query(bufs[k % 2], None, wei)
        q.run_carry(carry[k], 1000, False, engine.MODE_STRICT, 0)          # async
        qs.append(q)
        if k + 1 < k_pieces:
            # content alternates per LOAD, so buffer (k + 1) % 2 receives pieces[(k + 1) // 2 % 2]: a stale or early read shows
            bufs[(k + 1) % 2].upload_rows(0, pieces[(k + 1) // 2 % 2])
    for k in range(k_pieces):
        s, ni, _ = carry[k].finish()
        src = 0 if k == 0 else (k // 2 % 2)
        assert np.array_equal(bits(s), bits(want[src][0])) and np.array_equal(ni, want[src][1]), k
    ctx.close()


@pytest.fixture
def tiny_budget(monkeypatch):
    def set_rows(rows_cap, n_acc):
        monkeypatch.setenv("SNPM_HBM_BUDGET_GB", repr(budget_for(rows_cap, n_acc) / 1e9))
    monkeypatch.setenv("SNPMATCH_GPUS", "1")
    monkeypatch.setenv("SNPMATCH_PACKED", "0")          # the budgets below are sized for int8 slabs
    # ... and for a packed DB that does not fit them whole either: with round 4's split layout a 50-accession packed panel has
    # 16-B rows and would; these tests are about the slab mechanics, so they keep round 3's 256-B packed rows
    monkeypatch.setenv("SNPM_PACKED_SPLIT", "0")
    return set_rows


def native_g(toy, path):
    snp_genotype.save_native(path, toy["snps"], toy["accs"], toy["positions"], toy["chrs"], toy["regions"])
    return snp_genotype.Genotype(path, None)


def test_product_path_streams_slabs_and_matches_reference_files(golden_dir, tmp_path, tiny_budget):
    """G2 / G3 / G5 with a budget that forces the DB through >= 3 slabs, from a native .snpm file: the reference's files"""
    toy = np.load(os.path.join(golden_dir, "toy_db.npz"))
    gold = json.load(open(os.path.join(golden_dir, "g2_inbred.json")))
    tiny_budget(1000, 50)
    for skip in (False, True):
        g = native_g(toy, str(tmp_path / "toy.snpm"))
        out = str(tmp_path / ("inbred%d" % skip))
        gt = snpmatch.Genotyper(make_inputs(toy), g, out, run_genotyper=True, skip_db_hets=skip)
        assert isinstance(g.panel(), engine.StreamedPanel) and g.panel().loads == 3          # 2400 matched rows, 1000 per piece
        want = gold["inbred_skip%d" % int(skip)]
        cmp_scores_table(open(out + ".scores.txt").read(), want["scores.txt"])
        assert open(out + ".matches.json").read() == want["matches.json"]
        assert len(gt.commonSNPs[0]) == 2400
    toy = np.load(os.path.join(golden_dir, "toy_db_refine.npz"))
    gold = json.load(open(os.path.join(golden_dir, "g3_refine.json")))
    tiny_budget(1000, 40)
    out = str(tmp_path / "refine")
    g = native_g(toy, str(tmp_path / "refine.snpm"))
    gt = snpmatch.Genotyper(make_inputs(toy), g, out, run_genotyper=False)
    gt.filter_tophits()
    assert isinstance(g.panel(), engine.StreamedPanel) and g.panel().loads >= 13             # + the --refine scan of all 10 000 rows
    assert hasattr(gt, "result_fine") == gold["has_result_fine"]
    cmp_scores_table(open(out + ".scores.txt").read(), gold["scores.txt"])
    cmp_scores_table(open(out + ".refined.scores.txt").read(), gold["refined.scores.txt"])
    assert open(out + ".matches.json").read() == gold["matches.json"]
    toy = np.load(os.path.join(golden_dir, "toy_db_cross.npz"))
    gold = json.load(open(os.path.join(golden_dir, "g5_cross.json")))
    tiny_budget(700, 30)
    for skip in (False, True):
        out = str(tmp_path / ("cross%d" % skip))
        g = snp_genotype.Genotype.from_arrays(toy["snps"], toy["accs"], toy["positions"], toy["chrs"], toy["regions"])
        csmatch.CrossIdentifier(make_inputs(toy), g, "athaliana_tair10", 300000, out, run_identifier=True, skip_db_hets=skip)
        assert isinstance(g.panel(), engine.StreamedPanel) and g.panel().loads >= 6
        want = gold["cross_skip%d" % int(skip)]
        cmp_window_table(open(out + ".windowscore.txt").read(), want[".windowscore.txt"])
        cmp_scores_table(open(out + ".scores.txt").read(), want[".scores.txt"])
        assert open(out + ".scores.txt.matches.json").read() == want[".scores.txt.matches.json"]
        assert os.path.exists(out + ".matches.json") == (".matches.json" in want)


def test_residency_plan_prefers_whole_panels(tmp_path, monkeypatch):
    """SNPMATCH_PACKED=0: int8 whole -> packed whole -> slabs, by the budget; a DB with an odd call code skips the packed step.
    Default (auto): packed whole first, int8 whole for a DB a packed panel cannot hold, int8 slabs when nothing fits whole."""
    monkeypatch.setenv("SNPMATCH_GPUS", "1")
    monkeypatch.setenv("SNPMATCH_PACKED", "0")
    rng = np.random.default_rng(1)
    n, n_acc = 20_000, 1024
    db = rand_db(rng, n, n_acc)
    meta = (np.array(["a%d" % i for i in range(n_acc)]), np.arange(1, n + 1), np.array(["1"]), np.array([[0, n]]))
    int8_bytes, packed_bytes = (n + 32) * row_pitch(n_acc) + 256, (n + 32) * row_pitch(n_acc, True) + 256

    def plan(budget, snps=db):
        monkeypatch.setenv("SNPM_HBM_BUDGET_GB", repr(budget / 1e9))
        g = snp_genotype.Genotype.from_arrays(snps, *meta)
        p = g.panel()
        kind = (type(p).__name__, bool(p.packed))
        rows = np.arange(0, n, 3)
        wei = synth.sample_weights(rng, rng.choice(np.array([0, 1, 2], dtype=np.int8), size=len(rows)), 0.8)
        s, ni = p.query(rows, wei).run(1000, False, engine.MODE_STRICT)
        ws, wn = c_oracle.genotyper(snps, rows, wei, 1000, False)
        assert np.array_equal(bits(s), bits(ws)) and np.array_equal(ni, wn)
        p.free()
        return kind

    assert plan(int8_bytes) == ("Panel", False)
    assert plan(int8_bytes - 1) == ("Panel", True)
    assert plan(packed_bytes - 1) == ("StreamedPanel", False)
    odd = db.copy()
    odd[n - 5, 7] = 4
    assert plan(int8_bytes - 1, odd) == ("StreamedPanel", False)
    monkeypatch.setenv("SNPMATCH_PACKED", "1")
    assert plan(packed_bytes - 1) == ("StreamedPanel", True)
    monkeypatch.delenv("SNPMATCH_PACKED")                 # auto
    assert plan(int8_bytes) == ("Panel", True) and plan(packed_bytes) == ("Panel", True)
    assert plan(int8_bytes, odd) == ("Panel", False)      # unpackable: the int8 panel after the packed load refused
    assert plan(packed_bytes - 1) == ("StreamedPanel", False) and plan(packed_bytes, odd) == ("StreamedPanel", False)


def test_reference_hdf5_db_through_the_native_reader(golden_dir, tmp_path, monkeypatch):
    """`-d toy_db.hdf5`: the reference's own DB format (lzf chunks of (1000, n_acc), written by real h5py) read by the
    library -- chunks decompressed by the loader's threads into the staging slabs -- resident, packed, as an accession
    shard, streamed through a budget, and through the CLI: the reference's files"""
    import subprocess
    import sys
    monkeypatch.setenv("SNPMATCH_GPUS", "1")
    toy = np.load(os.path.join(golden_dir, "toy_db.npz"))
    path = os.path.join(golden_dir, "h5", "toy_db.hdf5")
    gold = json.load(open(os.path.join(golden_dir, "g2_inbred.json")))["inbred_skip0"]
    ctx = make_ctx(SNPM_STAGE_MB=1)
    from snpmatch_amd import h5
    f = h5.File(path)
    store = engine.RowStore(h5=(f, "snps"))
    for packed in (False, True):
        p = engine.Panel.from_store(ctx, store, packed=packed)
        assert np.array_equal(p.download_rows(0, 10000), toy["snps"])
        p.free()
        p = engine.Panel.from_store(ctx, store, packed=packed, cols=(12, 40))
        assert np.array_equal(p.download_rows(0, 10000), toy["snps"][:, 12:40])
        rows = np.array([9999, 3, 1000, 999, 1001, 5000], dtype=np.int64)
        store.load(p, (12, 40), rows, 100)
        assert np.array_equal(p.download_rows(100, 6), toy["snps"][rows, 12:40])
        store.load(p, (12, 40), (2500, 1300), 0)                                   # starts and ends inside chunks
        assert np.array_equal(p.download_rows(0, 1300), toy["snps"][2500:3800, 12:40])
        p.free()
    ctx.close()
    # (an int8 source that fits no whole panel streams as int8 slabs: the scenario needs a packed DB larger than two int8 half-buffers,
    # i.e. round 3's 256-B packed rows -- with the split layout the 50-accession DB packs into 16-B rows and always fits first)
    monkeypatch.setenv("SNPM_PACKED_SPLIT", "0")
    for budget_rows in (None, 1000):
        if budget_rows:
            monkeypatch.setenv("SNPM_HBM_BUDGET_GB", repr(budget_for(budget_rows, 50, packed=True) / 1e9))     # too small for the packed DB whole
        g = snp_genotype.Genotype(path, None)
        out = str(tmp_path / ("h5_%s" % budget_rows))
        snpmatch.Genotyper(make_inputs(toy), g, out, run_genotyper=True)
        assert type(g.panel()).__name__ == ("StreamedPanel" if budget_rows else "Panel")
        cmp_scores_table(open(out + ".scores.txt").read(), gold["scores.txt"])
        assert open(out + ".matches.json").read() == gold["matches.json"]
    monkeypatch.delenv("SNPM_HBM_BUDGET_GB")
    sample = str(tmp_path / "sample.npz")
    np.savez(sample, chr=toy["s_chrs"], pos=toy["s_pos"], gt=toy["s_gt"], wei=toy["s_wei"], dp=toy["s_dp"])
    out = str(tmp_path / "cli_h5")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "snpmatch_amd", "inbred", "-i", sample, "-d", path, "-o", out],
                       env=dict(os.environ, PYTHONPATH=root), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    cmp_scores_table(open(out + ".scores.txt").read(), gold["scores.txt"])
    assert open(out + ".matches.json").read() == gold["matches.json"]


def test_packed_flat_panel_file_feeds_both_panel_kinds(case, golden_dir, tmp_path, monkeypatch):
    """a .snpm written with 2 bits per call: its rows travel as they are (snpm_panel_load_file_rows_packed) into packed panels
    (re-pitch) and into int8 panels (unpacked on the device) -- whole, as accession shards starting at multiples of 4, as row
    lists, streamed through a budget -- and Genotyper writes the reference's files from it"""
    db, wei, _ = case
    n, n_acc = db.shape
    monkeypatch.setenv("SNPMATCH_GPUS", "1")
    out = str(tmp_path / "packed.snpm")
    meta = (np.array(["a%d" % i for i in range(n_acc - 1)]), np.arange(1, n + 1), np.array(["1"]), np.array([[0, n]]))
    snp_genotype.save_native(out, db[:, :n_acc - 1], *meta, packed=True)          # 299 accessions: a partial last byte
    store = engine.RowStore(npy_packed=(os.path.join(out, "snps.p2.npy"), n_acc - 1))
    rows = np.sort(np.random.default_rng(8).choice(n, size=5000, replace=False)).astype(np.int64)
    for env in ({}, {"SNPM_ODIRECT": 1, "SNPM_STAGE_MB": 1}):
        ctx = make_ctx(**env)
        for packed in (False, True):
            p = engine.Panel.from_store(ctx, store, packed=packed)
            assert np.array_equal(p.download_rows(0, n), db[:, :n_acc - 1])
            p.free()
            p = engine.Panel(ctx, 6000, 103, packed=packed)                        # accessions 196 .. 298: a shard that ends in the partial byte
            store.load(p, (196, 299), rows, 3)
            assert np.array_equal(p.download_rows(3, 5000), db[rows, 196:299])
            store.load(p, (196, 299), (21_000, 6000), 0)
            assert np.array_equal(p.download_rows(0, 6000), db[21_000:27_000, 196:299])
            p.free()
            p = engine.Panel(ctx, 100, 50, packed=packed)                          # a shard in the middle: the neighbours' calls in its last byte are masked
            store.load(p, (100, 150), (0, 100), 0)
            assert np.array_equal(p.download_rows(0, 100), db[:100, 100:150])
            with pytest.raises(AssertionError):
                store.load(p, (101, 151), (0, 100), 0)
            p.free()
        ctx.close()
    toy = np.load(os.path.join(golden_dir, "toy_db.npz"))
    gold = json.load(open(os.path.join(golden_dir, "g2_inbred.json")))["inbred_skip0"]
    path = str(tmp_path / "toy_packed.snpm")
    snp_genotype.save_native(path, toy["snps"], toy["accs"], toy["positions"], toy["chrs"], toy["regions"], packed=True)
    for budget_rows in (None, 1000):
        if budget_rows:
            monkeypatch.setenv("SNPM_HBM_BUDGET_GB", repr(budget_for(budget_rows, 50, packed=True) / 1e9))
        g = snp_genotype.Genotype(path, None)
        o = str(tmp_path / ("packedfile_%s" % budget_rows))
        snpmatch.Genotyper(make_inputs(toy), g, o, run_genotyper=True)
        assert type(g.panel()).__name__ == ("StreamedPanel" if budget_rows else "Panel")
        cmp_scores_table(open(o + ".scores.txt").read(), gold["scores.txt"])
        assert open(o + ".matches.json").read() == gold["matches.json"]


def test_latest_format_hdf5_through_the_loader(golden_dir):
    """a file in the "latest" HDF5 file format (version-2 object headers, a PAGED fixed-array index over 1763 lzf chunks of
    (40, 8), raw chunks among them): the loader's threads decode its chunks into the staging slabs -- whole, as a column
    range, as a row list, into an int8 and into a packed panel -- and a query on it equals the oracle"""
    from snpmatch_amd import h5
    want = np.load(os.path.join(golden_dir, "h5", "stress_expected.npz"))["snps"]
    want = np.where(want < 0, -1, np.where(want > 2, 3, want)).astype(np.int8)     # a panel's canonical codes: missing -1, "matches nothing" 3
    f = h5.File(os.path.join(golden_dir, "h5", "latest_stress.hdf5"))
    store = engine.RowStore(h5=(f, "snps"))
    ctx = make_ctx(SNPM_STAGE_MB=1)
    p = engine.Panel.from_store(ctx, store, packed=False)
    assert np.array_equal(p.download_rows(0, want.shape[0]), want)
    p.free()
    with pytest.raises(AssertionError, match="packed panel"):       # rows 10 000 - 13 000 hold arbitrary bytes: not packable
        engine.Panel.from_store(ctx, store, packed=True)
    rows = np.sort(np.random.default_rng(2).choice(want.shape[0], size=3000, replace=False)).astype(np.int64)
    p = engine.Panel.from_store(ctx, store, packed=False, cols=(2, 7))
    assert np.array_equal(p.download_rows(40_900, 200), want[40_900:41_100, 2:7])      # across the index's page border (chunk 1024)
    store.load(p, (2, 7), rows, 10)
    assert np.array_equal(p.download_rows(10, len(rows)), want[rows, 2:7])
    p.free()
    # the rows a packed panel can hold, as a row list into a packed panel, scored against the oracle
    keep = np.concatenate([np.arange(0, 10_000), np.arange(13_000, want.shape[0])]).astype(np.int64)
    p = engine.Panel(ctx, len(keep), 8, packed=True)
    store.load(p, (0, 8), keep, 0)
    assert np.array_equal(p.download_rows(0, len(keep)), want[keep])
    sel = np.sort(np.random.default_rng(4).choice(len(keep), size=5000, replace=False)).astype(np.int64)
    codes = np.where(want[keep][sel, 3] < 0, 0, want[keep][sel, 3]).astype(np.int8)
    wei = synth.sample_weights(np.random.default_rng(5), codes, 0.8)
    s, n = engine.Query(p, sel, wei).run(1000, False, engine.MODE_STRICT)
    ws, wn = c_oracle.genotyper(want[keep], sel, wei, 1000, False)
    assert np.array_equal(np.ascontiguousarray(s).view(np.uint64), np.ascontiguousarray(ws).view(np.uint64)) and np.array_equal(n, wn)
    p.free()
    ctx.close()
    f.close()
