#!/opt/conda/bin/python3.9
"""
Writes the HDF5 fixtures under tests/golden/h5/ with h5py (the image's /opt/conda/bin/python3.9 has h5py 3.3 / HDF5 1.10.6
with the LZF filter; the interpreter the tests run on has no h5py):

    /opt/conda/bin/python3.9 tests/golden/make_hdf5_fixtures.py

The datasets are created exactly as the reference's writers do (pygwas/genotype.py:310-326 `save_as_hdf5`: accessions,
positions (i4) with attrs chrs / chr_regions, snps int8 lzf-compressed in chunks of (1000, num_accessions); core/makedb.py:64-81
`save_as_hdf5_acc`: the same with gzip chunks of (num_snps, 1)), so the native reader (csrc/snpm_h5.cpp) is tested on
files of the layout a user's all_chromosomes_binary.hdf5 / .acc.hdf5 have.  Expected contents are the .npz inputs
(tests/golden/toy_db.npz) and, for the stress file, the arrays stored beside it in h5/stress_expected.npz (latest_stress.hdf5 holds the same arrays).
"""
import os

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "h5")


def save_as_hdf5(path, snps, accessions, positions, chrs, chr_regions, data_format="binary"):
    h5file = h5py.File(path, "w")
    num_snps, num_accessions = snps.shape
    h5file.create_dataset("accessions", data=accessions, shape=(num_accessions,))
    h5file.create_dataset("positions", data=positions, shape=(num_snps,), dtype="i4")
    h5file["positions"].attrs["chrs"] = np.array(chrs, dtype="S")
    h5file["positions"].attrs["chr_regions"] = chr_regions
    h5file.create_dataset("snps", shape=(num_snps, num_accessions), dtype="int8", compression="lzf",
                          chunks=((min(1000, num_snps), num_accessions)), data=snps)
    h5file["snps"].attrs["data_format"] = data_format
    h5file["snps"].attrs["num_snps"] = num_snps
    h5file["snps"].attrs["num_accessions"] = num_accessions
    h5file.close()


def save_as_hdf5_acc(path, snps, accessions, positions, chrs, chr_regions, data_format="binary"):
    h5file = h5py.File(path, "w")
    num_snps, num_acc = snps.shape
    h5file.create_dataset("accessions", data=accessions, shape=(num_acc,))
    h5file.create_dataset("positions", data=positions, shape=(num_snps,), dtype="i4")
    h5file["positions"].attrs["chrs"] = np.array(chrs, dtype="S")
    h5file["positions"].attrs["chr_regions"] = chr_regions
    h5file.create_dataset("snps", shape=(num_snps, num_acc), dtype="int8", compression="gzip", chunks=((num_snps, 1)))
    for i in range(num_acc):
        h5file["snps"][:, i] = snps[:, i]
    h5file["snps"].attrs["data_format"] = data_format
    h5file["snps"].attrs["num_snps"] = num_snps
    h5file["snps"].attrs["num_accessions"] = num_acc
    h5file.close()


def main():
    os.makedirs(OUT, exist_ok=True)
    toy = np.load(os.path.join(HERE, "toy_db.npz"))
    accs = np.asarray(toy["accs"]).astype("S")
    regions = [tuple(int(v) for v in r) for r in toy["regions"]]
    chrs = [str(c) for c in toy["chrs"]]
    save_as_hdf5(os.path.join(OUT, "toy_db.hdf5"), toy["snps"], accs, toy["positions"], chrs, regions)
    save_as_hdf5_acc(os.path.join(OUT, "toy_db.acc.hdf5"), toy["snps"], accs, toy["positions"], chrs, regions)
    # stress file: > 64 chunks (a two-level chunk B-tree), a last chunk that sticks out of the dataset, chunks LZF cannot shrink
    # (stored raw, filter mask set), a gzip + shuffle dataset, variable-length string accessions, a scalar-ish attribute zoo
    rng = np.random.default_rng(5)
    n, a = 70_500, 8
    snps = rng.choice(np.array([-1, 0, 1, 2], dtype=np.int8), size=(n, a), p=[0.05, 0.6, 0.33, 0.02])
    snps[10_000:13_000] = rng.integers(-128, 128, size=(3000, a), dtype=np.int8)          # incompressible rows
    noise = rng.integers(-128, 128, size=(5000, 3), dtype=np.int8)
    wide = rng.choice(np.array([-1, 0, 1, 2], dtype=np.int8), size=(2500, 1135), p=[0.05, 0.6, 0.33, 0.02])
    pos = np.sort(rng.choice(10_000_000, size=n, replace=False)).astype("i4")
    path = os.path.join(OUT, "stress.hdf5")
    with h5py.File(path, "w") as f:
        f.create_dataset("accessions", data=np.array(["acc_%d" % i for i in range(a)], dtype=object), dtype=h5py.string_dtype())
        f.create_dataset("positions", data=pos, dtype="i4")
        f["positions"].attrs["chrs"] = np.array(["Chr1", "Chr2"], dtype="S")
        f["positions"].attrs["chr_regions"] = [(0, 40_000), (40_000, n)]
        f.create_dataset("snps", data=snps, dtype="int8", compression="lzf", chunks=(1000, a))
        f["snps"].attrs["data_format"] = "binary"
        f["snps"].attrs["num_snps"] = n
        f["snps"].attrs["ratio"] = 0.25
        f["snps"].attrs["small"] = np.arange(6, dtype="i2").reshape(2, 3)
        f.create_dataset("noise", data=noise, compression="lzf", chunks=(777, 2))              # 2-D chunk grid, ragged edges
        f.create_dataset("wide_gzip", data=wide, compression="gzip", shuffle=True, chunks=(1000, 1135))
        f.create_dataset("plain", data=wide[:10])                                              # contiguous
        f.create_dataset("tiny", data=np.arange(5, dtype="i8"))                                # compact or contiguous
        g = f.create_group("grp")
        g.create_dataset("inner", data=np.arange(12, dtype="f8").reshape(3, 4))
    np.savez_compressed(os.path.join(OUT, "stress_expected.npz"), snps=snps, noise=noise, wide=wide, pos=pos)
    # files in the "latest" HDF5 file format (version-2 object headers, link messages, version-4 layouts with their own chunk
    # indexes).  latest_format.hdf5: a small DB of the reference's layout (fixed-array index, one page)
    with h5py.File(os.path.join(OUT, "latest_format.hdf5"), "w", libver="latest") as f:
        f.create_dataset("snps", data=snps[:2000], compression="lzf", chunks=(1000, a))
        f.create_dataset("accessions", data=np.array(["a%d" % i for i in range(a)], dtype="S"))
        f.create_dataset("positions", data=pos[:2000])
        f["positions"].attrs["chrs"] = np.array(["1"], dtype="S")
        f["positions"].attrs["chr_regions"] = [(0, 2000)]
    # latest_stress.hdf5: a PAGED fixed array (1763 chunks > 1024 per page) of filtered chunks with raw (incompressible) ones
    # among them, an unfiltered chunked dataset, single-chunk datasets (filtered and not), a chunk that was never written, a
    # nested group, variable-length strings, attributes; and two things the reader refuses by name: a dataset with an unlimited
    # dimension (extensible-array index) and a group with dense link storage (more than 8 members)
    with h5py.File(os.path.join(OUT, "latest_stress.hdf5"), "w", libver="latest") as f:
        f.create_dataset("accessions", data=np.array(["acc_%d" % i for i in range(a)], dtype=object), dtype=h5py.string_dtype())
        f.create_dataset("positions", data=pos, dtype="i4")
        f["positions"].attrs["chrs"] = np.array(["Chr1", "Chr2"], dtype="S")
        f["positions"].attrs["chr_regions"] = [(0, 40_000), (40_000, n)]
        f.create_dataset("snps", data=snps, dtype="int8", compression="lzf", chunks=(40, a))          # 1763 chunks: two pages
        f["snps"].attrs["data_format"] = "binary"
        f["snps"].attrs["num_snps"] = n
        f.create_dataset("noise", data=noise, chunks=(777, 2))                                         # chunked, no filter
        f.create_dataset("wide_gzip", data=wide, compression="gzip", shuffle=True, chunks=(2500, 1135))   # ONE filtered chunk
        x = f.create_group("extra")                    # (the root keeps 8 members: one more and its links move to dense storage)
        x.create_dataset("one_plain_chunk", data=wide[:100], chunks=(100, 1135))                      # ONE unfiltered chunk
        x.create_dataset("plain", data=wide[:10])                                                      # contiguous
        x.create_dataset("tiny", data=np.arange(5, dtype="i8"))
        holes = x.create_dataset("holes", shape=(3000, 4), dtype="int8", chunks=(1000, 4), compression="lzf", fillvalue=0)
        holes[0:1000] = snps[0:1000, :4]
        holes[2000:3000] = snps[2000:3000, :4]                                                         # chunk 1 never written
        g = f.create_group("grp")
        g.create_dataset("inner", data=np.arange(12, dtype="f8").reshape(3, 4))
        g.create_group("deeper").create_dataset("leaf", data=np.arange(7, dtype="i2"))
        x.create_dataset("growing", data=snps[:3000], maxshape=(None, a), chunks=(1000, a), compression="lzf")
        many = f.create_group("many")
        for i in range(12):
            many.create_dataset("d%02d" % i, data=np.arange(3) + i)
    # latest_unlimited.hdf5: a DB whose snps dataset can grow (extensible-array chunk index): the native reader refuses it by
    # name and core/snp_genotype falls back to h5py where that is importable
    with h5py.File(os.path.join(OUT, "latest_unlimited.hdf5"), "w", libver="latest") as f:
        f.create_dataset("snps", data=snps[:2000], compression="lzf", chunks=(1000, a), maxshape=(None, a))
        f.create_dataset("accessions", data=np.array(["a%d" % i for i in range(a)], dtype="S"))
        f.create_dataset("positions", data=pos[:2000])
        f["positions"].attrs["chrs"] = np.array(["1"], dtype="S")
        f["positions"].attrs["chr_regions"] = [(0, 2000)]
    for fn in sorted(os.listdir(OUT)):
        print(fn, os.path.getsize(os.path.join(OUT, fn)))


if __name__ == "__main__":
    main()
