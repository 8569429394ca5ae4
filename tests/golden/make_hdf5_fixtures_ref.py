#!/opt/conda/bin/python3.9
"""
Build-container-only (needs /root/reference and the image's second interpreter with h5py):

    /opt/conda/bin/python3.9 tests/golden/make_hdf5_fixtures_ref.py

Writes tests/golden/h5/toy_db.hdf5 and toy_db.acc.hdf5 with the REFERENCE'S OWN writers -- Genotype.save_as_hdf5
(pygwas/genotype.py:310-326) and makedb.save_as_hdf5_acc (core/makedb.py:64-81), imported from /root/reference, not re-typed --
from the arrays of tests/golden/toy_db.npz, then reads both files back with the reference's reader (HDF5Genotype,
pygwas/genotype.py:534-673: snps rows and columns, positions, accessions, chrs, chr_regions, chromosomes) and stores what
it returned in tests/golden/h5/toy_db_ref_read.npz: the expected values of the native reader (csrc/snpm_h5.cpp,
tests/test_h5_native_cpu.py).  Placeholder modules stand in for the reference's unused imports (allel, hmmlearn), as in
make_golden.py; h5py, numpy and scipy are the real ones of this interpreter (h5py 3.3, numpy 1.26, scipy 1.7).
The stress / latest-format files of make_hdf5_fixtures.py are not touched: they exercise file-format features the reference's
writers never produce.
"""
import os
import sys
import types
import warnings

import numpy as np

if not os.path.isdir("/root/reference"):
    sys.exit("the reference is not present here")
warnings.filterwarnings("ignore")
sys.dont_write_bytecode = True
for _m in ("allel", "hmmlearn", "hmmlearn.hmm"):
    sys.modules[_m] = types.ModuleType(_m)
sys.modules["hmmlearn"].hmm = sys.modules["hmmlearn.hmm"]
sys.path.insert(0, "/root/reference")
from snpmatch.pygwas import genotype as pg  # noqa: E402
from snpmatch.core import makedb  # noqa: E402

assert "/root/reference" in pg.__file__ and "/root/reference" in makedb.__file__
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "h5")


def main():
    toy = np.load(os.path.join(HERE, "toy_db.npz"))
    accs = np.asarray(toy["accs"]).astype("S")
    regions = [tuple(int(v) for v in r) for r in toy["regions"]]
    chrs = [str(c) for c in toy["chrs"]]
    g = pg.Genotype(toy["snps"], toy["positions"], accs, regions, chrs, data_format="binary")
    p_row, p_acc = os.path.join(OUT, "toy_db.hdf5"), os.path.join(OUT, "toy_db.acc.hdf5")
    g.save_as_hdf5(p_row)                    # the reference's row-chunked writer
    makedb.save_as_hdf5_acc(g, p_acc)        # ... and its accession-chunked one
    # what the reference's reader hands out
    rng = np.random.default_rng(3)
    n_snp, n_acc = toy["snps"].shape
    rows = np.sort(rng.choice(n_snp, size=300, replace=False))
    cols = np.sort(rng.choice(n_acc, size=7, replace=False))
    out = {"rows": rows, "cols": cols}
    for tag, path in (("row", p_row), ("acc", p_acc)):
        h = pg.load_hdf5_genotype_data(path)
        assert isinstance(h, pg.HDF5Genotype)
        out[tag + "_snps_all"] = np.asarray(h.snps[:, :])
        out[tag + "_snps_rows"] = np.asarray(h.snps[rows, :])                       # g.g.snps[idx, :], core/snpmatch.py:222
        out[tag + "_snps_cols"] = np.stack([np.asarray(h.snps[:, int(c)]) for c in cols], axis=1)   # g_acc.snps[:, i], core/csmatch.py:116
        out[tag + "_positions"] = np.asarray(h.positions)
        out[tag + "_accessions"] = np.asarray(h.accessions).astype("S")
        out[tag + "_chrs"] = np.asarray(h.chrs).astype("S")
        out[tag + "_chr_regions"] = np.asarray(h.chr_regions)
        out[tag + "_chromosomes"] = np.asarray(h.chromosomes).astype("S")
        out[tag + "_data_format"] = np.asarray([h.data_format]).astype("S")
        del h
    assert np.array_equal(out["row_snps_all"], toy["snps"]) and np.array_equal(out["acc_snps_all"], toy["snps"])
    del out["row_snps_all"], out["acc_snps_all"]          # equal to toy_db.npz (asserted): not stored twice
    np.savez_compressed(os.path.join(OUT, "toy_db_ref_read.npz"), **out)
    for fn in ("toy_db.hdf5", "toy_db.acc.hdf5", "toy_db_ref_read.npz"):
        print(fn, os.path.getsize(os.path.join(OUT, fn)))


if __name__ == "__main__":
    main()
