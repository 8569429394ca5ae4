#!/usr/bin/env python3
"""
Throughput of the native HDF5 path on the GPU box (profiles/r03c_time_h5_loader.txt):
a 1001-Genomes-shaped DB (1135 accessions x N SNPs, the value mix of SURVEY 8d) is written by REAL h5py with the reference's
layout (lzf chunks of (1000, n_acc)) -- by the image's /opt/conda/bin/python3.9, in a child process -- then loaded into an int8
and a packed panel by the library's own reader (chunks -> loader threads -> pinned slabs -> HBM), next to the same DB as .snpm.
"""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

WRITER = r'''
import sys, numpy as np, h5py
src, dst = sys.argv[1], sys.argv[2]
snps = np.load(src, mmap_mode="r")
n, a = snps.shape
f = h5py.File(dst, "w")
f.create_dataset("accessions", data=np.array(["%d" % (6000 + i) for i in range(a)], dtype="S"))
f.create_dataset("positions", data=np.arange(1, n + 1, dtype="i4"))
f["positions"].attrs["chrs"] = np.array(["1"], dtype="S")
f["positions"].attrs["chr_regions"] = [(0, n)]
ds = f.create_dataset("snps", shape=(n, a), dtype="int8", compression="lzf", chunks=(1000, a))
for r in range(0, n, 100000):
    ds[r:r + 100000] = snps[r:r + 100000]
f.close()
'''


def main():
    from snpmatch_amd import engine, h5
    n_acc = 1135
    n = int(float(sys.argv[1]) * 1e9 / n_acc) // 1000 * 1000 if len(sys.argv) > 1 else 4_000_000
    work = sys.argv[2] if len(sys.argv) > 2 else "/dev/shm"
    ctx = engine.Context(0)
    gen = engine.Panel(ctx, n, n_acc)
    gen.fill_synthetic(1001)
    host = gen.download_rows(0, n)
    gen.free()
    npy = os.path.join(work, "h5_timing_snps.npy")
    np.save(npy, host)
    path = os.path.join(work, "h5_timing.hdf5")
    t0 = time.perf_counter()
    subprocess.check_call(["/opt/conda/bin/python3.9", "-c", WRITER, npy, path])
    print("DB %d x %d int8 = %.2f GB; h5py wrote %s (%.2f GB, lzf chunks of (1000, %d)) in %.1f s"
          % (n, n_acc, host.nbytes / 1e9, path, os.path.getsize(path) / 1e9, n_acc, time.perf_counter() - t0))
    try:
        f = h5.File(path)
        for label, store in (("HDF5 (lzf), native reader", engine.RowStore(h5=(f, "snps"))), (".npy flat file", engine.RowStore(npy=npy))):
            for packed in (False, True):
                best = 1e30
                for _ in range(3):
                    p = engine.Panel(ctx, n, n_acc, packed=packed)
                    ctx.synchronize()
                    t0 = time.perf_counter()
                    store.load(p, (0, n_acc), None, 0, n)
                    p.upload_wait()
                    best = min(best, time.perf_counter() - t0)
                    if _ == 0:
                        assert np.array_equal(p.download_rows(n - 5000, 5000), host[n - 5000:]) and np.array_equal(p.download_rows(0, 3000), host[:3000])
                    p.free()
                print("  %-28s -> %-6s panel: %.3f s = %.1f GB/s of int8 DB bytes" % (label, "packed" if packed else "int8", best, host.nbytes / 1e9 / best))
        t0 = time.perf_counter()
        rows = np.sort(np.random.default_rng(0).choice(n, size=200_000, replace=False)).astype(np.int64)
        p = engine.Panel(ctx, len(rows), n_acc)
        engine.RowStore(h5=(f, "snps")).load(p, (0, n_acc), rows, 0)
        p.upload_wait()
        dt = time.perf_counter() - t0
        assert np.array_equal(p.download_rows(0, 1000), host[rows[:1000]])
        print("  the matched rows of a 200k-SNP sample (g.g.snps[idx, :]) from the HDF5 file: %.3f s" % dt)
        f.close()
    finally:
        for q in (npy, path):
            if os.path.exists(q):
                os.remove(q)
    ctx.close()


if __name__ == "__main__":
    main()
