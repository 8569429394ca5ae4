"""
The native HDF5 reader (csrc/snpm_h5.cpp, snpmatch_amd/h5.py) against files written by REAL h5py / libhdf5 1.10.6.
toy_db.hdf5 / toy_db.acc.hdf5 come out of the REFERENCE'S OWN writers (Genotype.save_as_hdf5, pygwas/genotype.py:310-326;
makedb.save_as_hdf5_acc, core/makedb.py:64-81), called by tests/golden/make_hdf5_fixtures_ref.py under the image's
/opt/conda/bin/python3.9; toy_db_ref_read.npz holds what the reference's reader (HDF5Genotype, pygwas/genotype.py:534-673)
returned for them.  The stress / latest-format files (tests/golden/make_hdf5_fixtures.py) exercise format features those
writers never produce.  No GPU: reading is host code.
"""
import os

import numpy as np
import pytest

from snpmatch_amd import h5

H5DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "h5")


@pytest.fixture(scope="module")
def toy():
    return np.load(os.path.join(os.path.dirname(H5DIR), "toy_db.npz"))


def test_reference_layout_row_chunked_lzf(toy):
    f = h5.File(os.path.join(H5DIR, "toy_db.hdf5"))
    assert sorted(f.keys()) == ["accessions", "positions", "snps"]
    snps = f["snps"]
    assert snps.shape == (10000, 50) and snps.dtype == np.int8 and snps.chunks == (1000, 50)
    assert np.array_equal(snps[:], toy["snps"])
    assert np.array_equal(snps[2500:3700], toy["snps"][2500:3700])                 # across chunk borders
    idx = np.array([9999, 0, 1000, 999, 5, 5, 7777])
    assert np.array_equal(snps[idx, :], toy["snps"][idx, :])                        # the reference's g.g.snps[idx, :]
    assert np.array_equal(snps[:, 17], toy["snps"][:, 17])
    assert np.array_equal(snps[3, :], toy["snps"][3]) and snps[9999, 49] == toy["snps"][9999, 49]
    assert np.array_equal(snps[100:200, 10:20], toy["snps"][100:200, 10:20])
    assert np.array_equal(np.asarray(snps), toy["snps"])
    assert snps.attrs["num_snps"] == 10000 and snps.attrs["num_accessions"] == 50
    assert snps.attrs["data_format"] in (b"binary", "binary")
    pos = f["positions"]
    assert pos.shape == (10000,) and pos.dtype == np.int32 and np.array_equal(pos[:], toy["positions"])
    assert np.array_equal(pos[5000:5010], toy["positions"][5000:5010])
    assert pos.attrs["chrs"].astype("U").tolist() == [str(c) for c in toy["chrs"]]
    assert np.array_equal(pos.attrs["chr_regions"], toy["regions"]) and pos.attrs["chr_regions"].dtype == np.int64
    assert sorted(pos.attrs.keys()) == ["chr_regions", "chrs"]
    assert f["accessions"][:].astype("U").tolist() == [str(a) for a in toy["accs"]]
    f.close()


def test_native_reader_returns_what_the_reference_reader_returned(toy):
    """both files, every member the path touches: rows g.g.snps[idx, :] (core/snpmatch.py:222), columns g_acc.snps[:, i]
    (core/csmatch.py:116), positions, accessions, chrs / chr_regions -- against the values HDF5Genotype handed out"""
    want = np.load(os.path.join(H5DIR, "toy_db_ref_read.npz"))
    rows, cols = want["rows"], want["cols"]
    for tag, name in (("row", "toy_db.hdf5"), ("acc", "toy_db.acc.hdf5")):
        with h5.File(os.path.join(H5DIR, name)) as f:
            snps = f["snps"]
            assert np.array_equal(snps[rows, :], want[tag + "_snps_rows"]) and snps[rows, :].dtype == want[tag + "_snps_rows"].dtype
            assert np.array_equal(np.stack([snps[:, int(c)] for c in cols], axis=1), want[tag + "_snps_cols"])
            assert np.array_equal(f["positions"][:], want[tag + "_positions"])
            assert np.array_equal(f["accessions"][:].astype("S"), want[tag + "_accessions"])
            assert np.array_equal(f["positions"].attrs["chrs"].astype("S"), want[tag + "_chrs"])
            assert np.array_equal(f["positions"].attrs["chr_regions"], want[tag + "_chr_regions"])
            assert str(np.asarray(snps.attrs["data_format"]).astype("U")) == want[tag + "_data_format"].astype("U")[0]
    # HDF5Genotype.chromosomes (pygwas/genotype.py:156-161): one chromosome name per SNP -- what core/snp_genotype rebuilds
    # from chrs + chr_regions
    from snpmatch_amd.core import snp_genotype
    g = snp_genotype.Genotype(os.path.join(H5DIR, "toy_db.hdf5"), os.path.join(H5DIR, "toy_db.acc.hdf5"))
    assert np.array_equal(np.asarray(g.g.chromosomes).astype("S"), want["row_chromosomes"])
    assert np.array_equal(np.asarray(g.g.positions), want["row_positions"])


def test_reference_layout_accession_chunked_gzip(toy):
    with h5.File(os.path.join(H5DIR, "toy_db.acc.hdf5")) as f:
        snps = f["snps"]
        assert snps.chunks == (10000, 1)
        assert np.array_equal(snps[:, 31], toy["snps"][:, 31])                      # g.g_acc.snps[:, i]
        assert np.array_equal(snps[:], toy["snps"])
        assert np.array_equal(snps[np.array([3, 9000]), :], toy["snps"][[3, 9000]])


def test_stress_file_two_level_btree_raw_chunks_shuffle_vlen():
    want = np.load(os.path.join(H5DIR, "stress_expected.npz"))
    with h5.File(os.path.join(H5DIR, "stress.hdf5")) as f:
        assert sorted(f.keys()) == ["accessions", "grp", "noise", "plain", "positions", "snps", "tiny", "wide_gzip"]
        snps = f["snps"]                                                            # 71 chunks: a two-level chunk B-tree; the last
        assert snps.shape == (70_500, 8) and snps.chunks == (1000, 8)               # chunk sticks out; rows 10 000-13 000 stored raw
        assert np.array_equal(snps[:], want["snps"])
        rows = np.sort(np.random.default_rng(0).choice(70_500, size=5000, replace=False))
        assert np.array_equal(snps[rows, :], want["snps"][rows])
        assert np.array_equal(snps[69_990:70_500, 2:7], want["snps"][69_990:70_500, 2:7])
        assert np.array_equal(f["noise"][:], want["noise"]) and f["noise"].chunks == (777, 2)      # ragged 2-D chunk grid
        assert np.array_equal(f["noise"][770:1600, 1:3], want["noise"][770:1600, 1:3])
        assert np.array_equal(f["wide_gzip"][:], want["wide"])                                     # gzip + shuffle
        assert np.array_equal(f["plain"][:], want["wide"][:10]) and f["plain"].chunks is None      # contiguous
        assert np.array_equal(f["tiny"][:], np.arange(5)) and f["tiny"].dtype == np.int64
        assert np.array_equal(f["positions"][:], want["pos"])
        assert f["accessions"][:].astype("U").tolist() == ["acc_%d" % i for i in range(8)]        # variable-length strings
        assert f["snps"].attrs["ratio"] == 0.25 and f["snps"].attrs["num_snps"] == 70_500
        assert np.array_equal(f["snps"].attrs["small"], np.arange(6, dtype="i2").reshape(2, 3))
        assert f["positions"].attrs["chr_regions"].tolist() == [[0, 40_000], [40_000, 70_500]]
        assert np.array_equal(f["grp"]["inner"][:], np.arange(12.0).reshape(3, 4))
        assert "nothing" not in f and "snps" in f
        with pytest.raises(IOError):
            f["nothing"]
        with pytest.raises(IndexError):
            snps[np.array([70_500]), :]


def test_damaged_and_foreign_files_give_errors(tmp_path):
    raw = open(os.path.join(H5DIR, "toy_db.hdf5"), "rb").read()
    p = str(tmp_path / "not_hdf5.bin")
    open(p, "wb").write(b"\x00" * 4096)
    with pytest.raises(IOError, match="signature"):
        h5.File(p)
    with pytest.raises(IOError, match="cannot open"):
        h5.File(str(tmp_path / "absent.hdf5"))
    # truncated in the middle of the chunk data: metadata still parses, reads fail cleanly
    p = str(tmp_path / "truncated.hdf5")
    open(p, "wb").write(raw[:len(raw) // 2])
    try:
        f = h5.File(p)
        with pytest.raises(IOError):
            f["snps"][:]
            f["positions"][:]
            f["accessions"][:]
    except IOError:
        pass
    # random damage anywhere: an error or data, never a crash
    rng = np.random.default_rng(3)
    for trial in range(40):
        b = bytearray(raw)
        for _ in range(int(rng.integers(1, 30))):
            b[int(rng.integers(8, len(b)))] = int(rng.integers(0, 256))
        p = str(tmp_path / ("fuzz%d.hdf5" % trial))
        open(p, "wb").write(bytes(b))
        try:
            with h5.File(p) as f:
                for name in f.keys():
                    f[name][:]
                f["positions"].attrs["chrs"]
        except (IOError, MemoryError, KeyError):
            pass


def test_genotype_opens_the_reference_hdf5_and_makedb_native_converts_it(toy, tmp_path):
    """core/snp_genotype.Genotype on the reference's two files (row-chunked + .acc twin found beside it, as the reference's
    constructor does, core/snp_genotype.py:26-41) and the CLI converter to the flat panel format -- all host code"""
    from snpmatch_amd import cli
    from snpmatch_amd.core import snp_genotype
    g = snp_genotype.Genotype(os.path.join(H5DIR, "toy_db.hdf5"), None)
    assert g.g_acc is not g.g and g.g_acc.snps.chunks == (10000, 1)
    assert g.accessions.tolist() == [str(a) for a in toy["accs"]] and g.chrs.tolist() == [str(c) for c in toy["chrs"]]
    assert np.array_equal(g.g.positions, toy["positions"]) and np.array_equal(g.g.chr_regions, toy["regions"])
    assert np.array_equal(g.g.snps[np.array([0, 4321, 9999]), :], toy["snps"][[0, 4321, 9999]])      # g.g.snps[idx, :]
    assert np.array_equal(g.g_acc.snps[:, 7], toy["snps"][:, 7])                                     # g.g_acc.snps[:, i]
    assert g.g.chromosomes[2500] == "2" and g.g.num_snps == 10000
    rows, sample = g.get_positions_idxs(toy["s_chrs"], toy["s_pos"])
    assert len(rows) == 2400
    out = str(tmp_path / "converted.snpm")
    assert cli.main(["makedb-native", "-i", os.path.join(H5DIR, "toy_db.hdf5"), "-o", out]) == 0
    g2 = snp_genotype.Genotype(out, None)
    assert np.array_equal(np.asarray(g2.g.snps), toy["snps"]) and g2.accessions.tolist() == g.accessions.tolist()
    assert np.array_equal(g2.g.positions, toy["positions"]) and np.array_equal(g2.g.chr_regions, toy["regions"])


def test_latest_file_format_reference_layout():
    """libver='latest': version-2 object headers, link messages in the root group, a version-4 layout with a fixed-array chunk
    index -- the same DB layout as the reference's writer, read natively and opened by core/snp_genotype.Genotype"""
    from snpmatch_amd.core import snp_genotype
    want = np.load(os.path.join(H5DIR, "stress_expected.npz"))
    path = os.path.join(H5DIR, "latest_format.hdf5")
    with h5.File(path) as f:
        assert sorted(f.keys()) == ["accessions", "positions", "snps"]
        assert f["snps"].shape == (2000, 8) and f["snps"].chunks == (1000, 8) and f["snps"].dtype == np.int8
        assert np.array_equal(f["snps"][:], want["snps"][:2000])
        assert np.array_equal(f["snps"][np.array([1999, 0, 1000, 999]), :], want["snps"][[1999, 0, 1000, 999]])
        assert np.array_equal(f["positions"][:], want["pos"][:2000])
        assert f["positions"].attrs["chrs"].astype("U").tolist() == ["1"]
        assert f["positions"].attrs["chr_regions"].tolist() == [[0, 2000]]
        assert f["accessions"][:].astype("U").tolist() == ["a%d" % i for i in range(8)]
    g = snp_genotype.Genotype(path, None)
    assert g.accessions.tolist() == ["a%d" % i for i in range(8)] and g.g.num_snps == 2000 and g.chrs.tolist() == ["1"]
    assert np.array_equal(g.g.snps[np.array([1, 1999]), :], want["snps"][[1, 1999]])
    assert hasattr(g.g, "h5_source")                       # served by the native reader (the loader threads read its chunks)


def test_latest_file_format_stress():
    """a paged fixed array (1763 filtered chunks, raw ones among them), unfiltered chunks, single-chunk datasets with and
    without filters, a chunk that was never written, nested new-style groups, variable-length strings, attributes; an unlimited
    dimension (extensible array) and a group of 12 members (dense link storage) are refused by name, the rest of the file
    stays readable"""
    want = np.load(os.path.join(H5DIR, "stress_expected.npz"))
    with h5.File(os.path.join(H5DIR, "latest_stress.hdf5")) as f:
        assert sorted(f.keys()) == ["accessions", "extra", "grp", "many", "noise", "positions", "snps", "wide_gzip"]
        x = f["extra"]
        assert sorted(x.keys()) == ["growing", "holes", "one_plain_chunk", "plain", "tiny"]
        snps = f["snps"]
        assert snps.shape == (70_500, 8) and snps.chunks == (40, 8)
        assert np.array_equal(snps[:], want["snps"])
        rows = np.sort(np.random.default_rng(1).choice(70_500, size=4000, replace=False))
        assert np.array_equal(snps[rows, :], want["snps"][rows])
        assert np.array_equal(snps[40_950:41_050, 1:6], want["snps"][40_950:41_050, 1:6])          # across the page border (chunk 1024)
        assert np.array_equal(f["noise"][:], want["noise"]) and f["noise"].chunks == (777, 2)
        assert np.array_equal(f["wide_gzip"][:], want["wide"]) and f["wide_gzip"].chunks == (2500, 1135)
        assert np.array_equal(x["one_plain_chunk"][:], want["wide"][:100])
        assert np.array_equal(x["plain"][:], want["wide"][:10]) and x["plain"].chunks is None
        assert np.array_equal(x["tiny"][:], np.arange(5))
        holes = x["holes"][:]
        assert np.array_equal(holes[:1000], want["snps"][:1000, :4]) and np.array_equal(holes[2000:], want["snps"][2000:3000, :4])
        assert not holes[1000:2000].any()                                                          # the unwritten chunk reads as the fill value
        assert np.array_equal(f["positions"][:], want["pos"])
        assert f["positions"].attrs["chr_regions"].tolist() == [[0, 40_000], [40_000, 70_500]]
        assert f["snps"].attrs["num_snps"] == 70_500 and f["snps"].attrs["data_format"] in (b"binary", "binary")
        assert f["accessions"][:].astype("U").tolist() == ["acc_%d" % i for i in range(8)]
        assert np.array_equal(f["grp"]["inner"][:], np.arange(12.0).reshape(3, 4))
        assert np.array_equal(f["grp"]["deeper"]["leaf"][:], np.arange(7)) and sorted(f["grp"].keys()) == ["deeper", "inner"]
        with pytest.raises(IOError, match="extensible array"):
            x["growing"][:]
        with pytest.raises(IOError, match="dense link storage"):
            f["many"].keys()
        assert np.array_equal(x["tiny"][:], np.arange(5))                                          # the file is still usable


def test_unsupported_index_falls_back_to_h5py(monkeypatch):
    """a DB the native reader refuses (its snps dataset has an unlimited dimension: extensible-array chunk index): it says so,
    and Genotype falls back to h5py where that is importable -- here a stand-in module that serves the expected arrays, so that
    the fallback branch of core/snp_genotype._load_any runs (the test image has no h5py for the interpreter the tests use)"""
    import sys
    import types
    from snpmatch_amd.core import snp_genotype
    path = os.path.join(H5DIR, "latest_unlimited.hdf5")
    with pytest.raises(IOError, match="extensible array"):
        h5.File(path)["snps"][:]
    monkeypatch.setitem(sys.modules, "h5py", None)                  # import h5py -> ImportError
    with pytest.raises(IOError, match="h5py is not installed"):
        snp_genotype.Genotype(path, None)
    want = np.load(os.path.join(H5DIR, "stress_expected.npz"))

    class FakeDataset(object):
        def __init__(self, arr, attrs=None):
            self.arr, self.attrs, self.shape = arr, attrs or {}, arr.shape

        def __getitem__(self, key):
            return self.arr[key]

    class FakeFile(dict):
        def __init__(self, name, mode="r"):
            assert name == path and mode == "r"
            dict.__init__(self, snps=FakeDataset(want["snps"][:2000]), accessions=FakeDataset(np.array(["a%d" % i for i in range(8)], dtype="S")),
                          positions=FakeDataset(want["pos"][:2000], {"chrs": np.array(["1"], dtype="S"), "chr_regions": np.array([[0, 2000]])}))

    fake = types.ModuleType("h5py")
    fake.File = FakeFile
    monkeypatch.setitem(sys.modules, "h5py", fake)
    g = snp_genotype.Genotype(path, None)
    assert g.accessions.tolist() == ["a%d" % i for i in range(8)] and g.g.num_snps == 2000 and g.chrs.tolist() == ["1"]
    assert np.array_equal(g.g.snps[np.array([1, 1999]), :], want["snps"][[1, 1999]])
    assert not hasattr(g.g, "h5_source")
