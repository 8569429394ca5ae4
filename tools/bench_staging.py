#!/usr/bin/env python3
"""
The DB staging path as bench legs (north_star: "the snpmatch.pygwas HDF5 loader is replaced by a pinned-host staging path feeding
hipMemcpyAsync on a side stream"; it replaces the h5py read of the reference, core/snpmatch.py:222, pygwas/genotype.py:548-550):

  a host-memory int8 panel of --gb 10^9 bytes (10 000 accessions wide)  ->  HBM
     through snpm_panel_upload_rows (int8 panel; 2-bit packed panel, packed by the filling threads),
     through snpm_panel_load_file_rows from a flat file in /dev/shm (int8 and packed panels),
  a 1135-accession DB in the reference's HDF5 layout (lzf chunks of (1000, n_acc), written by real h5py in a child process of the
     image's second interpreter when it is there)  ->  HBM through the library's own HDF5 reader,
  the wire itself in the same run: hipMemcpyAsync of one pinned 64-MiB block, over and over (the ceiling of every leg),
  an upload into one panel WHILE a resident query is scored again and again on the compute stream (overlap efficiency),
  and what the rates mean for the one-time load of configs[3] (10 000 x 50M int8 = 500 GB of DB bytes).

Rates are int8 DB bytes per second (rows x accessions / time): what the reference's g.g.snps[idx, :] would have to deliver.
bench.py runs this as a child process and puts the JSON it prints under "staging".

usage: tools/bench_staging.py [--gb 20] [--h5-gb 4] [--shm-dir /dev/shm] [--threads 16]
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from snpmatch_amd import engine  # noqa: E402

H5_WRITER = r'''
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


def ctx_with(**env):
    for k, v in env.items():
        os.environ[k] = str(v)
    try:
        return engine.Context(0)
    finally:
        for k in env:
            del os.environ[k]


def best_of(fn, panel, reps):
    best = 1e30
    for _ in range(reps):
        panel.ctx.synchronize()
        t0 = time.perf_counter()
        fn()
        panel.upload_wait()
        best = min(best, time.perf_counter() - t0)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gb", type=float, default=20.0)
    ap.add_argument("--h5-gb", type=float, default=4.0)
    ap.add_argument("--n-acc", type=int, default=10000)
    ap.add_argument("--shm-dir", default="/dev/shm")
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--reps", type=int, default=2)
    args = ap.parse_args()
    import torch

    n_acc = args.n_acc
    n = int(args.gb * 1e9 / n_acc) // 1000 * 1000
    out = {"legs": []}
    ctx = ctx_with(SNPM_STAGE_THREADS=args.threads)
    gen = engine.Panel(ctx, n, n_acc)
    gen.fill_synthetic(31337)
    host = gen.download_rows(0, n)
    gbytes = host.nbytes / 1e9
    out["workload"] = "host int8 panel %d accessions x %d SNPs = %.1f GB (device-generated, seed 31337), %d host threads fill the pinned slabs" % (
        n_acc, n, gbytes, args.threads)

    # ---- the wire: one pinned 64 MiB block, hipMemcpyAsync host -> device, no fill
    pin = torch.empty(64 << 20, dtype=torch.uint8).pin_memory()
    dev = torch.empty(64 << 20, dtype=torch.uint8, device="cuda:0")
    dev.copy_(pin, non_blocking=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        dev.copy_(pin, non_blocking=True)
    torch.cuda.synchronize()
    ceiling = 100 * (64 << 20) / 1e9 / (time.perf_counter() - t0)
    out["hipMemcpyAsync_ceiling_GBs"] = ceiling
    out["ceiling_note"] = "pinned host -> device, 100 x 64 MiB, same process, same run"
    del pin, dev

    def leg(name, seconds, wire_bytes, **extra):
        e = {"leg": name, "seconds": seconds, "int8_GBs": gbytes / seconds, "wire_GBs": wire_bytes / 1e9 / seconds,
             "wire_frac_of_ceiling": wire_bytes / 1e9 / seconds / ceiling}
        e.update(extra)
        out["legs"].append(e)
        return e

    # ---- (1) host memory -> HBM (snpm_panel_upload_rows)
    p8 = gen                                           # reuse the generator's buffer as the int8 destination
    dt = best_of(lambda: p8.upload_rows(0, host), p8, args.reps)
    assert np.array_equal(p8.download_rows(n - 2000, 2000), host[n - 2000:])
    leg("host_memory_to_int8_panel", dt, host.nbytes, entry="snpm_panel_upload_rows")
    pk = engine.Panel(ctx, n, n_acc, packed=True)
    dt = best_of(lambda: pk.upload_rows(0, host), pk, args.reps)
    assert np.array_equal(pk.download_rows(n - 2000, 2000), host[n - 2000:])
    leg("host_memory_to_packed_panel", dt, host.nbytes / 4, entry="snpm_panel_upload_rows",
        note="2-bit packing by the filling host threads (AVX2): a quarter of the bytes cross PCIe")
    pk.free()

    # ---- (2) upload into one panel while a resident query is scored on the compute stream
    # K asynchronous strict passes over panel A are queued (each returns at once), then panel B is loaded from host memory
    n_q = min(n, 2_000_000)
    wei = np.zeros((n_q, 3))
    wei[:, 0] = 1.0
    pa = engine.Panel(ctx, n_q, n_acc)
    pa.fill_synthetic(7)
    q = engine.Query(pa, None, wei)
    carry = engine.Carry(ctx, n_acc)
    q.run_carry(carry, 1000, False, engine.MODE_FAST, 0)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        q.run_carry(carry, 1000, False, engine.MODE_FAST, 0)
    ctx.synchronize()
    per_pass = (time.perf_counter() - t0) / 8
    t_up = best_of(lambda: p8.upload_rows(0, host), p8, 1)
    k_pass = max(4, int(t_up / per_pass))
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(k_pass):
        q.run_carry(carry, 1000, False, engine.MODE_FAST, 0)
    t_enq = time.perf_counter() - t0
    ctx.synchronize()
    t_q = time.perf_counter() - t0
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(k_pass):
        q.run_carry(carry, 1000, False, engine.MODE_FAST, 0)
    p8.upload_rows(0, host)
    p8.upload_wait()
    ctx.synchronize()
    t_both = time.perf_counter() - t0
    out["overlap"] = {"upload_alone_s": t_up, "scoring_alone_s": t_q, "scoring_passes": k_pass, "enqueue_of_the_passes_s": t_enq,
                      "both_s": t_both, "efficiency": max(t_up, t_q) / t_both,
                      "note": "efficiency = max(upload alone, scoring alone) / both together; the passes are k_fast over a resident "
                              "%d x %d slab on the compute stream, the upload runs on the copy stream" % (n_acc, n_q)}
    q.free()
    pa.free()

    # ---- (3) flat file in memory-backed storage -> HBM (snpm_panel_load_file_rows: pread into the pinned slabs)
    shm = os.path.join(args.shm_dir, "snpm_bench_staging.npy")
    try:
        np.save(shm, host)
        store = engine.RowStore(npy=shm)
        dt = best_of(lambda: store.load(p8, (0, n_acc), None, 0, n), p8, args.reps)
        assert np.array_equal(p8.download_rows(0, 2000), host[:2000])
        leg("flat_file_shm_to_int8_panel", dt, host.nbytes, entry="snpm_panel_load_file_rows")
        pk = engine.Panel(ctx, n, n_acc, packed=True)
        dt = best_of(lambda: store.load(pk, (0, n_acc), None, 0, n), pk, args.reps)
        leg("flat_file_shm_to_packed_panel", dt, host.nbytes / 4, entry="snpm_panel_load_file_rows")
        pk.free()
    finally:
        if os.path.exists(shm):
            os.remove(shm)
    p8.free()
    del host

    # ---- (4) the reference's HDF5 file (lzf chunks of (1000, n_acc)) -> HBM through the native reader
    py39 = "/opt/conda/bin/python3.9"
    if args.h5_gb > 0 and os.path.exists(py39):
        from snpmatch_amd import h5
        a5 = 1135
        n5 = int(args.h5_gb * 1e9 / a5) // 1000 * 1000
        g5 = engine.Panel(ctx, n5, a5)
        g5.fill_synthetic(1001)
        h5host = g5.download_rows(0, n5)
        npy = os.path.join(args.shm_dir, "snpm_bench_staging_h5src.npy")
        path = os.path.join(args.shm_dir, "snpm_bench_staging.hdf5")
        try:
            np.save(npy, h5host)
            t0 = time.perf_counter()
            subprocess.check_call([py39, "-c", H5_WRITER, npy, path], timeout=600)
            t_write = time.perf_counter() - t0
            f = h5.File(path)
            st5 = engine.RowStore(h5=(f, "snps"))
            for packed in (False, True):
                p5 = g5 if not packed else engine.Panel(ctx, n5, a5, packed=True)
                best = best_of(lambda: st5.load(p5, (0, a5), None, 0, n5), p5, args.reps)
                assert np.array_equal(p5.download_rows(n5 - 3000, 3000), h5host[n5 - 3000:])
                out["legs"].append({"leg": "reference_hdf5_lzf_to_%s_panel" % ("packed" if packed else "int8"), "seconds": best,
                                    "int8_GBs": h5host.nbytes / 1e9 / best, "entry": "snpm_panel_load_h5",
                                    "db": "%d x %d int8 = %.2f GB, file %.2f GB (h5py %.0f s to write)" % (
                                        n5, a5, h5host.nbytes / 1e9, os.path.getsize(path) / 1e9, t_write)})
                if packed:
                    p5.free()
            f.close()
        except Exception as e:          # noqa: BLE001
            out["legs"].append({"leg": "reference_hdf5_lzf", "error": str(e)[:200]})
        finally:
            for pth in (npy, path):
                if os.path.exists(pth):
                    os.remove(pth)
        g5.free()
    else:
        out["legs"].append({"leg": "reference_hdf5_lzf", "skipped": "no %s with h5py on this box" % py39})

    # ---- what it means for configs[3] (10 000 x 50M int8 = 500 GB of DB bytes; one GPU holds it 2-bit packed: 125 GB)
    by = {e["leg"]: e for e in out["legs"] if "int8_GBs" in e}
    c3 = {}
    if "host_memory_to_int8_panel" in by:
        c3["int8_slabs_from_host_memory_s"] = 500.0 / by["host_memory_to_int8_panel"]["int8_GBs"]
    if "host_memory_to_packed_panel" in by:
        c3["packed_panel_from_host_int8_s"] = 500.0 / by["host_memory_to_packed_panel"]["int8_GBs"]
    c3["note"] = ("one-time load of the 10 000 x 50M DB: the int8 panel (500 GB) exceeds one GPU's HBM and would be re-streamed per job "
                  "at the first rate; the 2-bit packed panel (125 GB) is resident after one load at the second rate (per GPU of 8: an eighth)")
    out["configs3_one_time_load"] = c3
    ctx.close()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
