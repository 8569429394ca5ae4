#!/usr/bin/env python3
"""
Loader throughput on the GPU box (profiles/r03_time_loader.txt):

  python tools/time_loader.py [--gb 20] [--n-acc 10000] [--disk-dir /tmp] [--shm-dir /dev/shm]

  (a) host memory -> int8 panel / packed panel (host-side 2-bit packing) / packed panel (device-side packing)
  (b) the same from a flat file in /dev/shm (page-cache speed) and from a file on disk, cold (O_DIRECT vs buffered)
  (c) a DB larger than the HBM budget streamed through two half-buffers (engine.StreamedPanel): wall time of the job
      against the time of its loads alone and the device time of its scoring kernels alone.
Rates are int8-equivalent DB bytes per second (n_rows * n_acc / time): what the reference's g.g.snps[idx, :] would deliver.
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def ctx_with(**env):
    from snpmatch_amd import engine
    for k, v in env.items():
        os.environ[k] = str(v)
    try:
        return engine.Context(0)
    finally:
        for k in env:
            del os.environ[k]


def timed_load(fn, panel, reps=3):
    best = 1e30
    for _ in range(reps):
        panel.ctx.synchronize()
        t0 = time.perf_counter()
        fn()
        panel.upload_wait()
        best = min(best, time.perf_counter() - t0)
    return best


def drop_cache(path):
    os.sync()
    fd = os.open(path, os.O_RDONLY)
    try:
        os.posix_fadvise(fd, 0, 0, os.POSIX_FADV_DONTNEED)
    finally:
        os.close(fd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gb", type=float, default=20.0)
    ap.add_argument("--n-acc", type=int, default=10000)
    ap.add_argument("--disk-dir", default="/tmp")
    ap.add_argument("--shm-dir", default="/dev/shm")
    ap.add_argument("--stream-gb", type=float, default=100.0, help="size of the streamed DB (0: skip)")
    ap.add_argument("--budget-gb", type=float, default=40.0)
    ap.add_argument("--threads", default="8,16,32")
    args = ap.parse_args()
    from snpmatch_amd import engine

    n_acc = args.n_acc
    n = int(args.gb * 1e9 / n_acc) // 1000 * 1000
    ctx = ctx_with()
    src_panel = engine.Panel(ctx, n, n_acc)
    src_panel.fill_synthetic(31337)
    t0 = time.perf_counter()
    host = src_panel.download_rows(0, n)
    print("source: %d x %d int8 = %.1f GB (device -> host in %.2f s)" % (n, n_acc, host.nbytes / 1e9, time.perf_counter() - t0))
    src_panel.free()
    ctx.close()
    gbytes = host.nbytes / 1e9

    # the wire itself: one pinned 64 MiB block copied to the device over and over (no fill)
    import torch
    pin = torch.empty(64 << 20, dtype=torch.uint8).pin_memory()
    dev = torch.empty(64 << 20, dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        dev.copy_(pin, non_blocking=True)
    torch.cuda.synchronize()
    print("\nPCIe ceiling: pinned host -> device, 100 x 64 MiB: %.1f GB/s" % (100 * (64 << 20) / 1e9 / (time.perf_counter() - t0)))
    del pin, dev

    print("\n(a) host memory -> panel, threads filling the pinned slabs")
    for thr in [int(x) for x in args.threads.split(",")]:
        for label, env, packed in (("int8 panel", {}, False), ("int8 panel, plain memcpy into the slabs", {"SNPM_NO_NT": 1}, False),
                                   ("packed panel, packed on the host", {}, True),
                                   ("packed panel, packed on the device", {"SNPM_HOST_PACK": 0}, True)):
            c = ctx_with(SNPM_STAGE_THREADS=thr, **env)
            p = engine.Panel(c, n, n_acc, packed=packed)
            dt = timed_load(lambda: p.upload_rows(0, host), p)
            print("  %2d threads  %-36s %6.1f GB/s  (%.3f s)" % (thr, label, gbytes / dt, dt))
            p.free()
            c.close()

    def file_rates(path, title, cold):
        print("\n(b) %s" % title)
        store = engine.RowStore(npy=path)
        for label, env, packed in (("int8 panel, buffered", {"SNPM_ODIRECT": 0}, False), ("int8 panel, O_DIRECT", {"SNPM_ODIRECT": 1}, False),
                                   ("packed panel (host), buffered", {"SNPM_ODIRECT": 0}, True),
                                   ("packed panel (host), O_DIRECT", {"SNPM_ODIRECT": 1}, True)):
            c = ctx_with(SNPM_STAGE_THREADS=16, **env)
            p = engine.Panel(c, n, n_acc, packed=packed)

            def go():
                if cold:
                    drop_cache(path)
                store.load(p, (0, n_acc), None, 0, n)
            try:
                dt = timed_load(go, p, reps=2)
                print("  %-36s %6.1f GB/s  (%.3f s)" % (label, gbytes / dt, dt))
            except Exception as e:          # noqa: BLE001
                print("  %-36s failed: %s" % (label, str(e)[:120]))
            p.free()
            c.close()

    def packed_file_rates(path_p2, title, cold):
        print("\n(b') %s" % title)
        store = engine.RowStore(npy_packed=(path_p2, n_acc))
        for label, env, packed in (("packed panel, buffered", {"SNPM_ODIRECT": 0}, True), ("packed panel, O_DIRECT", {"SNPM_ODIRECT": 1}, True),
                                   ("int8 panel (unpacked on the device), buffered", {"SNPM_ODIRECT": 0}, False)):
            c = ctx_with(SNPM_STAGE_THREADS=16, **env)
            p = engine.Panel(c, n, n_acc, packed=packed)

            def go():
                if cold:
                    drop_cache(path_p2)
                store.load(p, (0, n_acc), None, 0, n)
            dt = timed_load(go, p, reps=2)
            print("  %-46s %6.1f GB/s int8-equivalent  (%.3f s)" % (label, gbytes / dt, dt))
            if not packed:
                assert np.array_equal(p.download_rows(n - 1000, 1000), host[n - 1000:])
            p.free()
            c.close()

    from snpmatch_amd import _lib
    shm = os.path.join(args.shm_dir, "snpm_time_loader.npy")
    t0 = time.perf_counter()
    np.save(shm, host)
    print("\nwrote %s in %.2f s" % (shm, time.perf_counter() - t0))
    shm_p2 = os.path.join(args.shm_dir, "snpm_time_loader.p2.npy")
    try:
        file_rates(shm, "flat file in %s (memory-backed), 16 threads" % args.shm_dir, cold=False)
        mm = np.lib.format.open_memmap(shm_p2, mode="w+", dtype=np.uint8, shape=(n, (n_acc + 3) // 4))
        t0 = time.perf_counter()
        for r0 in range(0, n, 1 << 18):
            mm[r0:r0 + (1 << 18)] = _lib.pack_rows_host(host[r0:r0 + (1 << 18)])
        mm.flush()
        del mm
        print("\npacked the DB on one host thread in %.2f s (%.1f GB/s of int8)" % (time.perf_counter() - t0, gbytes / (time.perf_counter() - t0)))
        packed_file_rates(shm_p2, "PACKED flat file (2 bits per call) in %s, 16 threads" % args.shm_dir, cold=False)
    finally:
        os.remove(shm)
        if os.path.exists(shm_p2):
            os.remove(shm_p2)
    disk = os.path.join(args.disk_dir, "snpm_time_loader.npy")
    try:
        t0 = time.perf_counter()
        np.save(disk, host)
        os.sync()
        print("\nwrote %s in %.2f s" % (disk, time.perf_counter() - t0))
        file_rates(disk, "flat file on disk (%s), cache dropped before every read, 16 threads" % args.disk_dir, cold=True)
        file_rates(disk, "the same file, warm (page cache)", cold=False)
        os.remove(disk)
        disk_p2 = os.path.join(args.disk_dir, "snpm_time_loader.p2.npy")
        try:
            mm = np.lib.format.open_memmap(disk_p2, mode="w+", dtype=np.uint8, shape=(n, (n_acc + 3) // 4))
            for r0 in range(0, n, 1 << 18):
                mm[r0:r0 + (1 << 18)] = _lib.pack_rows_host(host[r0:r0 + (1 << 18)])
            mm.flush()
            del mm
            os.sync()
            packed_file_rates(disk_p2, "PACKED flat file on disk (%s), cache dropped before every read" % args.disk_dir, cold=True)
        finally:
            if os.path.exists(disk_p2):
                os.remove(disk_p2)
    finally:
        if os.path.exists(disk):
            os.remove(disk)

    if args.stream_gb > 0:
        # (c) a .snpm larger than the budget: written slab by slab into /dev/shm from the device generator
        from snpmatch_amd import synth  # noqa: F401
        n_big = int(args.stream_gb * 1e9 / n_acc) // 1000 * 1000
        path = os.path.join(args.shm_dir, "snpm_stream_test.npy")
        mm = np.lib.format.open_memmap(path, mode="w+", dtype=np.int8, shape=(n_big, n_acc))
        c = ctx_with(SNPM_STAGE_THREADS=16)
        gen = engine.Panel(c, n, n_acc)
        t0 = time.perf_counter()
        for r0 in range(0, n_big, n):
            nr = min(n, n_big - r0)
            gen.fill_synthetic(31337, snp0=r0, row0=0, nrows=nr)
            mm[r0:r0 + nr] = gen.download_rows(0, nr)
        mm.flush()
        del mm
        gen.free()
        print("\n(c) streamed DB: %d x %d int8 = %.1f GB in %s (written in %.1f s), HBM budget %.0f GB"
              % (n_big, n_acc, n_big * n_acc / 1e9, path, time.perf_counter() - t0, args.budget_gb))
        try:
            wei = torch.empty((n_big, 3), dtype=torch.float64, device="cuda:0")
            c.sample_synthetic(31337, 0, n_big, 417, wei.data_ptr())
            c.synchronize()
            wei_h = wei.cpu().numpy()
            del wei
            store = engine.RowStore(npy=path)
            for packed in (False, True):
                sp = engine.StreamedPanel(c, store, packed=packed, budget_bytes=int(args.budget_gb * 1e9))
                q = sp.query(None, wei_h)
                t0 = time.perf_counter()
                q.run(1000, False, engine.MODE_EXACT)
                first = time.perf_counter() - t0           # first read of the freshly written file
                sp.loads = 0
                c.profile(True)
                c.profile_reset()
                t0 = time.perf_counter()
                s, ni = q.run(1000, False, engine.MODE_EXACT)
                wall = time.perf_counter() - t0
                k_n, k_ms = c.profile_read("strict")
                c.profile(False)
                pieces = sp.loads
                # the loads alone: the same pieces into the same half-buffers, nothing scored
                per = sp.rows_cap // 1000 * 1000
                t0 = time.perf_counter()
                for k, r0 in enumerate(range(0, n_big, per)):
                    sp._load(k % 2, (r0, min(per, n_big - r0)))
                sp.halves[0].upload_wait()
                loads = time.perf_counter() - t0
                print("  %-7s %2d pieces of <= %d rows: job %.2f s = %.1f GB/s (first read of the new file: %.2f s);  loads alone %.2f s;  "
                      "scoring kernels alone %.3f s (%d launches);  job / max(load, score) = %.3f, job / (load + score) = %.3f;  top hit %d"
                      % ("packed" if packed else "int8", pieces, sp.rows_cap, wall, n_big * n_acc / 1e9 / wall, first, loads, k_ms / 1e3, k_n,
                         wall / max(loads, k_ms / 1e3), wall / (loads + k_ms / 1e3), int(np.argmax(s / ni))))
                sp.free()
        finally:
            os.remove(path)
            c.close()


if __name__ == "__main__":
    main()
