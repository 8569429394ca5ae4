"""
Thin object layer over the C ABI: Context (one GPU), Panel (DB genotype matrix resident in HBM),
Query (one sample's matched SNPs resident in HBM).  All compute happens in libsnpmatch_hip.so.
"""
import atexit
import ctypes as C
import weakref

import numpy as np

from . import _lib
from ._lib import MODE_EXACT, MODE_FAST, MODE_STRICT, check, ptr  # noqa: F401

_default_ctx = None


class Context(object):
    def __init__(self, device_id=0, _borrowed=None):
        lib = _lib.load()
        if _borrowed is None:
            h = C.c_void_p()
            rc = lib.snpm_init(int(device_id), C.byref(h))
            check(rc, None)
        else:                   # a member context of a Group: created and destroyed by the group (snpm_group_create_local)
            h = _borrowed
        self._owns = _borrowed is None
        self.lib = lib
        self.h = h
        self.device_id = int(device_id)
        self._children = weakref.WeakSet()      # panels and queries: freed before the context
        self._pinned = []                       # pinned host blocks handed out by pinned_empty
        self._pinned_bufs = []                  # ... and the ctypes objects their numpy views hang on
        # every context is closed before the interpreter (and with it the HIP runtime) goes down, also when the
        # script ends with an exception; the weak reference keeps the hook from pinning the object.  The library
        # tolerates the other order too (snpm_destroy orphans live panels / queries, include/snpmatch_hip.h).
        ref = weakref.ref(self)
        self._atexit = lambda: (ref() is not None) and ref().close(_at_exit=True)
        atexit.register(self._atexit)

    def close(self, _at_exit=False):
        if self.h:
            kids = list(self._children)
            for k in kids:
                if isinstance(k, Query):
                    k.free()
            for k in kids:
                if isinstance(k, (Panel, Carry, StreamedPanel)):
                    k.free()
            import sys
            if not _at_exit and not sys.is_finalizing():
                # a numpy view of a pinned block that is still referenced would dangle after the free below
                live = [b for b in self._pinned_bufs if sys.getrefcount(b) > 3]
                if live:
                    raise RuntimeError("Context.close(): %d array(s) from pinned_empty() are still referenced; "
                                       "delete them before closing the context" % len(live))
            for blk in self._pinned:
                self.lib.snpm_host_free(self.h, blk)
            self._pinned = []
            self._pinned_bufs = []
            if self._owns:
                self.lib.snpm_destroy(self.h)
            self.h = None
            try:
                atexit.unregister(self._atexit)
            except Exception:
                pass

    def __del__(self):
        try:
            self.close(_at_exit=True)
        except Exception:
            pass

    def set_stream(self, hip_stream):
        """Launch the library's kernels on the given hipStream_t handle (e.g. torch.cuda.Stream().cuda_stream).
        None restores the library's own stream.  Handle 0 (the legacy default stream) is refused: pass a
        real stream so that ordering with the caller's other work is explicit."""
        if hip_stream is None:
            check(self.lib.snpm_set_stream(self.h, None), self.h)
            return
        if int(hip_stream) == 0:
            raise ValueError("set_stream needs a non-default stream handle (got 0); use torch.cuda.Stream()")
        check(self.lib.snpm_set_stream(self.h, C.c_void_p(int(hip_stream))), self.h)

    def synchronize(self):
        check(self.lib.snpm_synchronize(self.h), self.h)

    def row_pitch(self, n_acc, packed=False):
        """bytes per row of a panel of ``n_acc`` accessions on this context (before it is created)"""
        out = C.c_int64(0)
        check(self.lib.snpm_panel_row_pitch(self.h, int(n_acc), 1 if packed else 0, C.byref(out)), self.h)
        return out.value

    def mem_info(self):
        """(free, total) device memory in bytes"""
        f, t = C.c_int64(0), C.c_int64(0)
        check(self.lib.snpm_device_mem_info(self.h, C.byref(f), C.byref(t)), self.h)
        return f.value, t.value

    # ---- profiling (HIP events on the stream the kernels run on)
    def profile(self, on=True):
        check(self.lib.snpm_profile_enable(self.h, 1 if on else 0), self.h)

    def profile_reset(self):
        check(self.lib.snpm_profile_reset(self.h), self.h)

    def profile_read(self, kernel):
        n = C.c_int64(0)
        ms = C.c_double(0)
        check(self.lib.snpm_profile_read(self.h, kernel.encode(), C.byref(n), C.byref(ms)), self.h)
        return n.value, ms.value

    # ---- one-shot forms
    def score_dense(self, wei, db, skip_hets=False):
        """matchGTsAccs on host arrays (fp64 bit-exact with the reference)."""
        db = np.asarray(db)
        wei = np.asarray(wei)
        assert wei.shape[0] == db.shape[0], "please provide same number of positions for both sample and db"
        assert wei.ndim == 2 and wei.shape[1] == 3, "SNP weights should be a np.array with  shape == n,3"
        db = np.ascontiguousarray(db, dtype=np.int8)
        wei = np.ascontiguousarray(wei, dtype=np.float64)
        n, n_acc = db.shape
        score = np.zeros(n_acc, dtype=np.float64)
        ninfo = np.zeros(n_acc, dtype=np.int64)
        if n_acc == 0:
            return score, ninfo
        check(self.lib.snpm_score_dense_host(self.h, ptr(db), n_acc, n, n_acc, ptr(wei), int(bool(skip_hets)),
                                             ptr(score), ptr(ninfo)), self.h)
        return score, ninfo

    def likelihood_device(self, d_y, d_n, m, length, d_lik, d_lrt, truncate=False, amin=None, check_domain=False):
        """likeliTest rows on device pointers (no host round trip unless check_domain)."""
        a = float("nan") if amin is None else float(amin)
        dom = C.c_int(0)
        check(self.lib.snpm_likelihood_device(self.h, C.c_void_p(d_y), C.c_void_p(d_n), int(m), int(length),
                                              int(bool(truncate)), a, C.c_void_p(d_lik), C.c_void_p(d_lrt),
                                              C.byref(dom) if check_domain else None), self.h)
        if check_domain and dom.value:
            raise AssertionError("provided y is greater than n")

    def likelihood(self, y, n, truncate=False, amin=None):
        """Rows of (matches y, informative n) -> (likelihood, ratio to the row minimum)."""
        y = np.ascontiguousarray(y, dtype=np.float64)
        n = np.ascontiguousarray(n, dtype=np.int64)
        assert y.shape == n.shape
        shape = y.shape
        if y.ndim == 1:
            m, ln = 1, y.shape[0]
        else:
            m, ln = y.shape
        lik = np.empty(shape, dtype=np.float64)
        lrt = np.empty(shape, dtype=np.float64)
        if y.size == 0:
            return lik, lrt
        a = float("nan") if amin is None else float(amin)
        check(self.lib.snpm_likelihood(self.h, ptr(y), ptr(n), m, ln, int(bool(truncate)), a, ptr(lik), ptr(lrt)), self.h)
        return lik, lrt

    def sample_synthetic(self, seed, snp0, n, planted, d_wei, err=0.02, frac_pl=0.8):
        """Benchmark sample generated on the device into ``d_wei`` (raw device pointer of a float64 [n, 3] buffer):
        accession ``planted`` of the synthetic panel ``seed`` with a fraction ``err`` of random calls and a fraction
        ``frac_pl`` of PL-derived weights.  ``snpmatch_amd.synth.sample_weights_twin`` is the numpy twin."""
        from . import synth
        tab = synth.exp_table()
        check(self.lib.snpm_sample_synthetic(self.h, C.c_uint64(int(seed)), int(snp0), int(n), int(planted),
                                             int(round(err * 1000)), int(round(frac_pl * 1000)), ptr(tab),
                                             C.c_void_p(int(d_wei))), self.h)

    def pinned_empty(self, shape, dtype):
        """numpy array in pinned host memory (hipHostMalloc): inputs of ``score_batch`` built here skip the staging copy.
        The block is released when the context closes: the array (and every view of it) must not be used after
        ``close()`` -- ``close`` refuses while such arrays are still referenced elsewhere."""
        dtype = np.dtype(dtype)
        n = int(np.prod(shape)) * dtype.itemsize
        h = C.c_void_p()
        check(self.lib.snpm_host_alloc(self.h, n, C.byref(h)), self.h)
        buf = (C.c_char * max(n, 1)).from_address(h.value)
        self._pinned.append(h)
        self._pinned_bufs.append(buf)           # numpy arrays made from it hold references to this ctypes object
        return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def binom_identity(self, x, n, error_rate=0.0005, pthres=0.05, return_sf=False):
        x = np.ascontiguousarray(x, dtype=np.float64)
        n = np.ascontiguousarray(n, dtype=np.int64)
        out = np.empty(len(x), dtype=np.int64)
        sf = np.empty(len(x), dtype=np.float64)
        check(self.lib.snpm_binom_identity(self.h, ptr(x), ptr(n), len(x), float(error_rate), float(pthres), ptr(out),
                                           ptr(sf)), self.h)
        return (out, sf) if return_sf else out


def default_context():
    """Process-wide context on the device named by SNPMATCH_DEVICE / LOCAL_RANK (default 0)."""
    global _default_ctx
    if _default_ctx is None:
        import os
        dev = int(os.environ.get("SNPMATCH_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        _default_ctx = Context(dev)         # closed at interpreter exit by its own atexit hook
    return _default_ctx


class RowStore(object):
    """Where the rows of a DB come from on the host side: a flat file of int8 rows (``npy=``: a C-ordered int8 .npy, read
    natively with pread / O_DIRECT by the library) or an array-like ``snps`` [n_snp, n_acc] (numpy array, memmap, h5py
    dataset: sliced here, uploaded from host memory).  ``load`` puts rows into a panel through the pinned staging path --
    the counterpart of the reference's ``g.g.snps[idx, :]`` (core/snpmatch.py:222, pygwas/genotype.py:548-550)."""

    def __init__(self, snps=None, npy=None, h5=None, npy_packed=None):
        assert (snps is not None) + (npy is not None) + (h5 is not None) + (npy_packed is not None) == 1
        self.snps, self.path, self.offset, self.h5, self.packed_file = snps, None, 0, h5, False
        if npy_packed is not None:          # (path of a uint8 .npy [n_snp, ceil(n_acc / 4)] with 2 bits per call, n_acc): a packed .snpm
            npy, n_acc_true = npy_packed
            with open(npy, "rb") as fh:
                major, _ = np.lib.format.read_magic(fh)
                shape, fortran, dtype = (np.lib.format.read_array_header_1_0(fh) if major == 1
                                         else np.lib.format.read_array_header_2_0(fh))
                self.offset = fh.tell()
            assert len(shape) == 2 and not fortran and np.dtype(dtype) == np.uint8 and shape[1] == (int(n_acc_true) + 3) // 4, \
                "expected a C-ordered uint8 matrix of 2-bit calls"
            self.path, self.n_snp, self.n_acc, self.packed_file, self.file_pitch = npy, int(shape[0]), int(n_acc_true), True, int(shape[1])
        elif h5 is not None:                  # (snpmatch_amd.h5.File, dataset name): the reference's HDF5 DB, read natively
            ds = h5[0][h5[1]]
            assert ds.ndim == 2 and ds.dtype == np.int8, "expected a 2-D int8 dataset"
            self.n_snp, self.n_acc = int(ds.shape[0]), int(ds.shape[1])
        elif npy is not None:
            with open(npy, "rb") as fh:
                major, _ = np.lib.format.read_magic(fh)
                shape, fortran, dtype = (np.lib.format.read_array_header_1_0(fh) if major == 1
                                         else np.lib.format.read_array_header_2_0(fh))
                self.offset = fh.tell()
            assert len(shape) == 2 and not fortran and np.dtype(dtype) == np.int8, "expected a C-ordered int8 matrix"
            self.path, self.n_snp, self.n_acc = npy, int(shape[0]), int(shape[1])
        else:
            self.n_snp, self.n_acc = int(snps.shape[0]), int(snps.shape[1])

    def load(self, panel, cols, rows, row0, nrows=None, slab_rows=1 << 16):
        """panel rows [row0, row0 + n) <- DB rows ``rows`` (an increasing int64 list; None: the range [0, nrows), or
        (first, n) for a range), columns cols = (a0, a1)"""
        a0, a1 = cols
        assert a1 - a0 == panel.n_acc
        if rows is None:
            rows = (0, self.n_snp if nrows is None else nrows)
        if not isinstance(rows, tuple):
            rows = np.ascontiguousarray(rows, dtype=np.int64)
            if len(rows) and int(rows[-1]) - int(rows[0]) + 1 == len(rows) and (len(rows) < 2 or bool(np.all(np.diff(rows) == 1))):
                rows = (int(rows[0]), len(rows))            # a contiguous run: the range form (one big read per thread)
        if self.h5 is not None:
            if isinstance(rows, tuple):
                panel.load_h5(self.h5[0], self.h5[1], a0, None, rows[0], row0, rows[1])
            else:
                panel.load_h5(self.h5[0], self.h5[1], a0, rows, 0, row0, len(rows))
            return
        if self.packed_file:
            assert a0 % 4 == 0, "accession shards of a packed DB start at multiples of 4"
            if isinstance(rows, tuple):
                panel.load_file_rows_packed(self.path, self.offset, self.file_pitch, a0, None, rows[0], row0, rows[1])
            else:
                panel.load_file_rows_packed(self.path, self.offset, self.file_pitch, a0, rows, 0, row0, len(rows))
            return
        if self.path is not None:
            if isinstance(rows, tuple):
                panel.load_file_rows(self.path, self.offset, self.n_acc, a0, None, rows[0], row0, rows[1])
            else:
                panel.load_file_rows(self.path, self.offset, self.n_acc, a0, rows, 0, row0, len(rows))
            return
        whole = (a0 == 0 and a1 == self.n_acc)
        if isinstance(rows, tuple):
            for r in range(rows[0], rows[0] + rows[1], slab_rows):
                r1 = min(r + slab_rows, rows[0] + rows[1])
                blk = self.snps[r:r1] if whole else self.snps[r:r1, a0:a1]
                panel.upload_rows(row0 + r - rows[0], np.ascontiguousarray(blk, dtype=np.int8))
        else:
            for i in range(0, len(rows), slab_rows):
                idx = rows[i:i + slab_rows]
                blk = self.snps[idx, :] if whole else self.snps[idx, a0:a1]      # the reference's fancy row read
                panel.upload_rows(row0 + i, np.ascontiguousarray(blk, dtype=np.int8))


_default_group = None


def device_count():
    n = C.c_int(0)
    check(_lib.load().snpm_device_count(C.byref(n)), None)
    return n.value


def group_devices():
    """Device ids of the GPUs one process should drive, or None for a single GPU.  SNPMATCH_GPUS = "all" (default when
    several GPUs are visible and no torch.distributed job is active), a count, or a comma-separated list of device ids
    (a device named twice needs SNPMATCH_GROUP_LOOPBACK=1: rehearsal of the sharded path on a one-GPU box)."""
    import os
    spec = os.environ.get("SNPMATCH_GPUS", "all").strip().lower()
    if "," in spec:
        ids = [int(x) for x in spec.split(",") if x.strip() != ""]
    else:
        n_vis = device_count()
        n = n_vis if spec in ("", "all") else max(1, min(int(spec), n_vis))
        first = int(os.environ.get("SNPMATCH_DEVICE", "0"))
        ids = [(first + i) % max(n_vis, 1) for i in range(n)]
    return ids if len(ids) > 1 else None


def default_group(n_members=None):
    """Process-wide ``Group`` over ``group_devices()`` (first ``n_members`` of them), created on first use; None for one GPU."""
    global _default_group
    import os
    ids = group_devices()
    if ids is None:
        return None
    if n_members is not None:
        ids = ids[:max(1, int(n_members))]
        if len(ids) < 2:
            return None
    if _default_group is not None and _default_group.h and [c.device_id for c in _default_group.contexts] == ids:
        return _default_group
    if _default_group is not None:
        _default_group.free()
    loopback = os.environ.get("SNPMATCH_GROUP_LOOPBACK", "0") not in ("", "0")
    _default_group = Group.local(ids, loopback=loopback)
    return _default_group


class Panel(object):
    """int8 [n_snp, n_acc] genotype matrix in HBM (SNP-major, row pitch padded to 256 B; to 128 B where that saves 5 % of the row)."""

    def __init__(self, ctx, n_snp, n_acc, packed=False, n_acc_total=None):
        self.ctx = ctx
        self.n_snp = int(n_snp)
        self.n_acc = int(n_acc)
        self.packed = bool(packed)      # 2 bits per call instead of one byte (same results, 4x less HBM)
        h = C.c_void_p()
        create = ctx.lib.snpm_panel_create_packed if packed else ctx.lib.snpm_panel_create
        check(create(ctx.h, self.n_snp, self.n_acc, C.byref(h)), ctx.h)
        self.h = h
        if n_acc_total is not None and int(n_acc_total) != self.n_acc:
            # a shard of a wider DB: the reference-order kernels follow the summation rule of the WHOLE panel's width
            # (numpy sums a one-accession panel pairwise, core/snpmatch.py:85-87; every wider one row after row)
            check(ctx.lib.snpm_panel_set_total_accessions(h, int(n_acc_total)), ctx.h)
        pitch = C.c_int64(0)
        dptr = C.c_void_p()
        check(ctx.lib.snpm_panel_info(h, None, None, C.byref(pitch), C.byref(dptr)), ctx.h)
        self.pitch = pitch.value
        self.device_ptr = dptr.value
        self._queries = weakref.WeakSet()
        ctx._children.add(self)

    @classmethod
    def from_host(cls, ctx, snps, slab_rows=1 << 16, packed=False, cols=None):
        """Upload an array-like [n_snp, n_acc] (numpy array, memmap or h5py dataset) slab by slab; ``cols`` = (a0, a1)
        keeps only that accession range (this rank's shard of an accession-sharded job)."""
        return cls.from_store(ctx, RowStore(snps=snps), packed=packed, cols=cols)

    @classmethod
    def from_store(cls, ctx, store, packed=False, cols=None):
        """the whole DB (or its columns ``cols``) resident: every row through the staging path"""
        a0, a1 = (0, store.n_acc) if cols is None else cols
        p = cls(ctx, store.n_snp, a1 - a0, packed=packed, n_acc_total=store.n_acc)
        try:
            store.load(p, (a0, a1), None, 0, store.n_snp)
        except Exception:
            p.free()
            raise
        return p            # returns once the last slab is enqueued; scoring calls wait for the copies on the device

    def query(self, row_idx, wei, row0=0):
        """the matched SNPs of one sample against this panel (``GroupPanel`` / ``StreamedPanel`` offer the same call)"""
        return Query(self, row_idx, wei, row0=row0)

    def genotype_once(self, row_idx, wei, sample_idx=None, chunk=1000, skip_hets=False, mode=MODE_EXACT, likelihoods=True, table=None):
        """ONE sample in ONE call (snpm_genotype_once): ``row_idx`` int64 [n] matched DB rows, ``wei`` float64 [n_wei, 3] the
        sample's weights, ``sample_idx`` int64 [n] the weight row of each matched SNP (None: rows 0..n-1).  With ``table``
        (float64 [<= 65536]) ``wei`` holds uint16 codes instead, weight = table[code] (snpm_genotype_once_coded: 10 instead of 32
        bytes per matched SNP over PCIe).  Returns a dict with score / ninfo (and lik / lrt of the truncated counts) [n_acc] and
        the re-evaluation counters of ``Query.run``."""
        ctx = self.ctx
        row_idx = np.ascontiguousarray(row_idx, dtype=np.int64)
        wei = np.asarray(wei)
        assert wei.ndim == 2 and wei.shape[1] == 3, "SNP weights should be a np.array with  shape == n,3"
        wei = np.ascontiguousarray(wei, dtype=np.float64 if table is None else np.uint16)
        if sample_idx is not None:
            sample_idx = np.ascontiguousarray(sample_idx, dtype=np.int64)
            assert len(sample_idx) == len(row_idx), "please provide same number of positions for both sample and db"
        else:
            assert len(wei) == len(row_idx), "please provide same number of positions for both sample and db"
        out = {"score": np.empty(self.n_acc, dtype=np.float64), "ninfo": np.empty(self.n_acc, dtype=np.int64)}
        if likelihoods:
            out["lik"] = np.empty(self.n_acc, dtype=np.float64)
            out["lrt"] = np.empty(self.n_acc, dtype=np.float64)
        info = np.zeros(4, dtype=np.int64)
        if table is not None:
            table = np.ascontiguousarray(table, dtype=np.float64)
            assert table.ndim == 1 and 1 <= len(table) <= 65536
            check(ctx.lib.snpm_genotype_once_coded(self.h, ptr(row_idx), ptr(wei), ptr(table), len(table), ptr(sample_idx), len(wei),
                                                   len(row_idx), int(chunk), int(bool(skip_hets)), int(mode), ptr(out["score"]),
                                                   ptr(out["ninfo"]), ptr(out.get("lik")), ptr(out.get("lrt")), ptr(info)), ctx.h)
        else:
            check(ctx.lib.snpm_genotype_once(self.h, ptr(row_idx), ptr(wei), ptr(sample_idx), len(wei), len(row_idx), int(chunk),
                                             int(bool(skip_hets)), int(mode), ptr(out["score"]), ptr(out["ninfo"]), ptr(out.get("lik")),
                                             ptr(out.get("lrt")), ptr(info)), ctx.h)
        out["n_strict_reeval"], out["all_integer_weights"], out["reeval_path"] = int(info[0]), bool(info[1]), int(info[2])
        return out

    def upload_rows(self, row0, rows):
        rows = np.ascontiguousarray(rows, dtype=np.int8)
        assert rows.ndim == 2 and rows.shape[1] == self.n_acc
        check(self.ctx.lib.snpm_panel_upload_rows(self.h, int(row0), rows.shape[0], ptr(rows), rows.shape[1]), self.ctx.h)

    def load_file(self, path, file_offset, row0=0, nrows=None):
        """Stream rows from a file of tightly packed int8 rows (native flat panel) through the staging path."""
        nrows = self.n_snp - row0 if nrows is None else nrows
        check(self.ctx.lib.snpm_panel_load_file(self.h, str(path).encode(), int(file_offset), int(row0), int(nrows)), self.ctx.h)

    def load_file_rows(self, path, file_offset, file_pitch, col0=0, row_idx=None, file_row0=0, row0=0, nrows=None):
        """rows from a file holding an int8 matrix of ``file_pitch`` bytes per row: file rows ``row_idx`` (or file_row0 ..),
        columns [col0, col0 + n_acc) -> panel rows [row0, row0 + nrows)  (snpm_panel_load_file_rows)"""
        if row_idx is not None:
            row_idx = np.ascontiguousarray(row_idx, dtype=np.int64)
            nrows = len(row_idx) if nrows is None else nrows
        check(self.ctx.lib.snpm_panel_load_file_rows(self.h, str(path).encode(), int(file_offset), int(file_pitch), int(col0),
                                                     ptr(row_idx), int(file_row0), int(row0), int(nrows)), self.ctx.h)

    def load_file_rows_packed(self, path, file_offset, file_pitch, acc0=0, row_idx=None, file_row0=0, row0=0, nrows=None):
        """rows from a PACKED flat file (2 bits per call, ``file_pitch`` bytes per row): accessions [acc0, acc0 + n_acc), acc0 % 4 == 0"""
        if row_idx is not None:
            row_idx = np.ascontiguousarray(row_idx, dtype=np.int64)
            nrows = len(row_idx) if nrows is None else nrows
        check(self.ctx.lib.snpm_panel_load_file_rows_packed(self.h, str(path).encode(), int(file_offset), int(file_pitch), int(acc0),
                                                            ptr(row_idx), int(file_row0), int(row0), int(nrows)), self.ctx.h)

    def load_h5(self, h5_file, dataset, col0=0, row_idx=None, file_row0=0, row0=0, nrows=None):
        """rows of a 2-D int8 dataset of an open ``snpmatch_amd.h5.File`` (the reference's DB format): the loader's threads read
        and decompress the chunks straight into the staging slabs (snpm_panel_load_h5)"""
        if row_idx is not None:
            row_idx = np.ascontiguousarray(row_idx, dtype=np.int64)
            nrows = len(row_idx) if nrows is None else nrows
        check(self.ctx.lib.snpm_panel_load_h5(self.h, h5_file.h, dataset.encode(), int(col0), ptr(row_idx), int(file_row0), int(row0),
                                              int(nrows)), self.ctx.h)

    @classmethod
    def from_npy(cls, ctx, path, packed=False, cols=None):
        """Panel from an int8 [n_snp, n_acc] .npy file (C order), read natively (no numpy copy)."""
        return cls.from_store(ctx, RowStore(npy=path), packed=packed, cols=cols)

    def upload_wait(self):
        check(self.ctx.lib.snpm_panel_upload_wait(self.h), self.ctx.h)

    def download_rows(self, row0, nrows):
        out = np.empty((int(nrows), self.n_acc), dtype=np.int8)
        check(self.ctx.lib.snpm_panel_download_rows(self.h, int(row0), int(nrows), ptr(out), self.n_acc), self.ctx.h)
        return out

    def fill_synthetic(self, seed, snp0=0, acc0=0, row0=0, nrows=None):
        """rows [row0, row0 + nrows) <- SNPs snp0 .. of the synthetic panel ``seed`` (columns acc0 ..)"""
        nrows = self.n_snp - row0 if nrows is None else nrows
        check(self.ctx.lib.snpm_panel_fill_synthetic_rows(self.h, C.c_uint64(int(seed)), int(snp0), int(acc0), int(row0),
                                                          int(nrows)), self.ctx.h)

    def segregating_rows(self, cols):
        """uint8 mask [n_snp]: 1 where the informative calls of accessions `cols` differ (device scan)."""
        cols = np.ascontiguousarray(cols, dtype=np.int32)
        mask = np.zeros(self.n_snp, dtype=np.uint8)
        check(self.ctx.lib.snpm_panel_segregating(self.h, ptr(cols), len(cols), ptr(mask)), self.ctx.h)
        return mask

    def segregating_first(self, cols):
        """(mask, first) for an accession-sharded DB: local mask and the first informative call per row (0xFF: none)"""
        cols = np.ascontiguousarray(cols, dtype=np.int32)
        mask = np.zeros(self.n_snp, dtype=np.uint8)
        first = np.full(self.n_snp, 0xFF, dtype=np.uint8)
        check(self.ctx.lib.snpm_panel_segregating_first(self.h, ptr(cols), len(cols), ptr(mask), ptr(first)), self.ctx.h)
        return mask, first

    def stream_read(self):
        """PMC calibration: read every panel byte once; returns the byte count."""
        n = C.c_int64(0)
        check(self.ctx.lib.snpm_debug_stream_read(self.h, C.byref(n)), self.ctx.h)
        return n.value

    def free(self):
        if self.h:
            for q in list(self._queries):
                q.free()
            if self.ctx.h:
                self.ctx.lib.snpm_panel_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Query(object):
    """A sample's matched SNPs (panel rows + weights) resident on the device."""

    def __init__(self, panel, row_idx, wei, row0=0, _device=None):
        self.panel = panel
        ctx = panel.ctx
        if _device is not None:             # (d_row_idx or None, d_wei, n): raw device pointers, see from_device
            d_rows, d_wei, n = _device
            self.n = int(n)
            h = C.c_void_p()
            check(ctx.lib.snpm_query_create_device(panel.h, C.c_void_p(d_rows) if d_rows else None, int(row0), self.n,
                                                   C.c_void_p(int(d_wei)), C.byref(h)), ctx.h)
            self.h = h
            panel._queries.add(self)
            ctx._children.add(self)
            return
        wei = np.asarray(wei)
        assert wei.ndim == 2 and wei.shape[1] == 3, "SNP weights should be a np.array with  shape == n,3"
        wei = np.ascontiguousarray(wei, dtype=np.float64)
        n = wei.shape[0]
        if row_idx is not None:
            row_idx = np.ascontiguousarray(row_idx, dtype=np.int64)
            assert row_idx.shape == (n,), "please provide same number of positions for both sample and db"
        self.n = n
        h = C.c_void_p()
        check(ctx.lib.snpm_query_create(panel.h, ptr(row_idx), int(row0), n, ptr(wei), C.byref(h)), ctx.h)
        self.h = h
        panel._queries.add(self)
        ctx._children.add(self)

    @classmethod
    def from_device(cls, panel, d_row_idx, d_wei, n, row0=0):
        """Query whose row list (int64 [n] or None = dense from ``row0``) and weights (float64 [n, 3]) already are
        in device memory (raw pointers); they are copied."""
        return cls(panel, None, None, row0=row0, _device=(d_row_idx, d_wei, n))

    def run(self, chunk=1000, skip_hets=False, mode=MODE_EXACT, return_info=False):
        """Genotyper.genotyper accumulators: (ScoreList float64 [n_acc], NumInfoSites int64 [n_acc])."""
        ctx = self.panel.ctx
        score = np.empty(self.panel.n_acc, dtype=np.float64)
        ninfo = np.empty(self.panel.n_acc, dtype=np.int64)
        info = np.zeros(4, dtype=np.int64) if return_info else None
        check(ctx.lib.snpm_query_run(self.h, int(chunk), int(bool(skip_hets)), int(mode), ptr(score), ptr(ninfo), ptr(info)),
              ctx.h)
        if return_info:
            return score, ninfo, {"n_strict_reeval": int(info[0]), "all_integer_weights": bool(info[1]),
                                  "reeval_path": int(info[2])}
        return score, ninfo

    def run_device(self, chunk=1000, skip_hets=False, mode=MODE_EXACT):
        """Enqueue the scoring (nothing waits on the host, the certificate included); returns raw device pointers
        (d_score f64[n_acc], d_ninfo i64[n_acc]).  ``last_reeval()`` reads the re-evaluation count afterwards."""
        ctx = self.panel.ctx
        ds, dn = C.c_void_p(), C.c_void_p()
        check(ctx.lib.snpm_query_run_device(self.h, int(chunk), int(bool(skip_hets)), int(mode), C.byref(ds), C.byref(dn),
                                            None), ctx.h)
        return ds.value, dn.value

    def last_reeval(self):
        """accessions the last certified run re-evaluated in reference order (synchronises)"""
        n = C.c_int64(0)
        check(self.panel.ctx.lib.snpm_query_last_reeval(self.h, C.byref(n)), self.panel.ctx.h)
        return n.value

    def last_kernel(self):
        return (self.panel.ctx.lib.snpm_query_last_kernel(self.h) or b"").decode()

    def run_carry(self, carry, chunk=1000, skip_hets=False, mode=MODE_EXACT, chunks_after=0):
        """score these rows as the next SNP slab of a larger job (see ``Carry`` / ``SlabScorer``)"""
        ctx = self.panel.ctx
        check(ctx.lib.snpm_query_run_carry(self.h, int(chunk), int(bool(skip_hets)), int(mode), int(chunks_after), carry.h),
              ctx.h)

    def bind_outputs(self, d_score, d_ninfo):
        """Write results of later runs into caller-owned device buffers (raw pointers, e.g. tensor.data_ptr())."""
        ctx = self.panel.ctx
        check(ctx.lib.snpm_query_bind_outputs(self.h, C.c_void_p(d_score) if d_score else None,
                                              C.c_void_p(d_ninfo) if d_ninfo else None), ctx.h)

    def error_bound(self, chunk=1000):
        b = C.c_double(0)
        check(self.panel.ctx.lib.snpm_query_error_bound(self.h, int(chunk), C.byref(b)), self.panel.ctx.h)
        return b.value

    def run_windows(self, win_off, skip_hets=False, totals=True, fast=False):
        """Per-window matchGTsAccs: (score [n_win,n_acc], ninfo [n_win,n_acc], tot_score, tot_ninfo).
        ``fast``: the segmented streaming pass with the certificate (exact counts, scores to ~1e-12) instead of the
        reference-order pass (fp64 bits); ``self.last_windows_info`` then holds the re-evaluation counters."""
        ctx = self.panel.ctx
        if fast:
            win_off = np.ascontiguousarray(win_off, dtype=np.int64)
            n_win = len(win_off) - 1
            na = self.panel.n_acc
            score = np.empty((n_win, na), dtype=np.float64)
            ninfo = np.empty((n_win, na), dtype=np.int64)
            ts = np.empty(na, dtype=np.float64)
            tn = np.empty(na, dtype=np.int64)
            info = np.zeros(4, dtype=np.int64)
            check(ctx.lib.snpm_query_run_windows_fast(self.h, ptr(win_off), n_win, int(bool(skip_hets)), ptr(score), ptr(ninfo),
                                                      ptr(ts), ptr(tn), ptr(info)), ctx.h)
            self.last_windows_info = {"pairs_reeval": int(info[0]), "totals_reeval": int(info[1]), "strict_fallback": bool(info[2])}
            return score, ninfo, ts, tn
        win_off = np.ascontiguousarray(win_off, dtype=np.int64)
        n_win = len(win_off) - 1
        na = self.panel.n_acc
        score = np.empty((n_win, na), dtype=np.float64)
        ninfo = np.empty((n_win, na), dtype=np.int64)
        ts = np.empty(na, dtype=np.float64)
        tn = np.empty(na, dtype=np.int64)
        check(ctx.lib.snpm_query_run_windows(self.h, ptr(win_off), n_win, int(bool(skip_hets)), ptr(score), ptr(ninfo),
                                             ptr(ts) if totals else None, ptr(tn) if totals else None), ctx.h)
        return score, ninfo, ts, tn

    def gather_columns(self, acc_idx):
        """calls of accessions ``acc_idx`` at the matched rows: uint8 [len(acc_idx), n] (0xFF = missing)"""
        acc_idx = np.ascontiguousarray(acc_idx, dtype=np.int32)
        out = np.full((len(acc_idx), self.n), 0xFF, dtype=np.uint8)
        check(self.panel.ctx.lib.snpm_query_gather_columns(self.h, ptr(acc_idx), len(acc_idx), ptr(out)), self.panel.ctx.h)
        return out

    def f1_pairs(self, acc_idx):
        """In-silico crosses of every pair of ``acc_idx`` (order of itertools.combinations) over the query's
        rows: (score float64 [n_pairs] with numpy's np.sum bits, ninfo int64 [n_pairs])."""
        ctx = self.panel.ctx
        acc_idx = np.ascontiguousarray(acc_idx, dtype=np.int32)
        k = len(acc_idx)
        n_pairs = k * (k - 1) // 2
        score = np.zeros(n_pairs, dtype=np.float64)
        ninfo = np.zeros(n_pairs, dtype=np.int64)
        check(ctx.lib.snpm_query_f1_pairs(self.h, ptr(acc_idx), k, ptr(score), ptr(ninfo)), ctx.h)
        return score, ninfo

    def free(self):
        if self.h:
            if self.panel.ctx.h and self.panel.h:
                self.panel.ctx.lib.snpm_query_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def weight_codes(wei, table):
    """uint16 codes [n, 3] with table[codes] == wei bit for bit, or None when some weight is not in ``table``
    (float64 [<= 65536], e.g. ``pl_table()``).  A binary search per weight on the host; parsers that still hold the
    integer PLs produce the codes directly instead (code = PL)."""
    wei = np.ascontiguousarray(wei, dtype=np.float64)
    order = np.argsort(table, kind="stable")
    pos = np.searchsorted(table[order], wei.ravel())
    pos[pos >= len(order)] = len(order) - 1
    codes = order[pos].astype(np.uint16)
    if not np.array_equal(table[codes].view(np.uint64), wei.ravel().view(np.uint64)):
        return None
    return codes.reshape(wei.shape)


def pl_table(n=8192):
    """table[k] = exp(-k / 10) for PL k = 0 .. n - 1 as numpy rounds it (what ParseInputs computes from a VCF's PL field,
    core/parsers.py:147-150).  exp underflows to exactly 0.0 from k = 7451 on, so the last entries double as the zeros
    of hard-call rows (code n - 1)."""
    return np.exp(np.arange(n, dtype=np.float64) / (-10))


def score_batch(panel, samples, chunk=1000, skip_hets=False, mode=MODE_EXACT, likelihoods=True, device=None, table=None, out=None):
    """Many samples against one resident panel in one call (snpm_score_batch).  ``samples``: list of
    (row_idx int64 [n_b], wei float64 [n_b, 3]) pairs.  With ``table`` (float64 [256]) the second element of a sample is
    uint16 codes [n_b, 3] instead, wei = table[codes] (snpm_score_batch_coded: 10 instead of 32 bytes per SNP over PCIe).  ``device`` = (d_row_idx, d_wei, offsets): the concatenated
    inputs already in device memory (raw pointers) instead.  Returns a dict with score / ninfo (and lik / lrt)
    arrays [B, n_acc] and the re-evaluation counters.  ``out``: the dict of an earlier call of the same shape, whose arrays are
    written again instead of new ones (a service that scores batch after batch: fresh arrays cost their page faults on every
    call, 9 MB of them for 256 samples x 1135 accessions)."""
    if isinstance(panel, GroupPanel):
        return panel.score_batch(samples, chunk, skip_hets, mode, likelihoods, device, table)
    ctx = panel.ctx
    wdtype = np.float64 if table is None else np.uint16
    if device is None and isinstance(samples, tuple):
        # already concatenated: (row_idx int64 [N], wei float64 [N, 3] (or codes uint8 [N, 3]), offsets int64 [B + 1])
        rows, wei, off = samples
        rows = np.ascontiguousarray(rows, dtype=np.int64)
        wei = np.ascontiguousarray(wei, dtype=wdtype)
        off = np.ascontiguousarray(off, dtype=np.int64)
        assert wei.ndim == 2 and wei.shape[1] == 3 and len(rows) == len(wei) == off[-1]
        p_rows, p_wei, dev_flag = ptr(rows), ptr(wei), 0
    elif device is None:
        off = np.zeros(len(samples) + 1, dtype=np.int64)
        for b, (rows, wei) in enumerate(samples):
            wei = np.asarray(wei)
            assert wei.ndim == 2 and wei.shape[1] == 3, "SNP weights should be a np.array with  shape == n,3"
            assert len(rows) == wei.shape[0], "please provide same number of positions for both sample and db"
            off[b + 1] = off[b] + len(rows)
        rows = np.ascontiguousarray(np.concatenate([np.asarray(s[0], dtype=np.int64) for s in samples]) if samples
                                    else np.zeros(0, dtype=np.int64))
        wei = np.ascontiguousarray(np.concatenate([np.asarray(s[1], dtype=wdtype) for s in samples]) if samples
                                   else np.zeros((0, 3), dtype=wdtype))
        p_rows, p_wei, dev_flag = ptr(rows), ptr(wei), 0
    else:
        d_rows, d_wei, off = device
        off = np.ascontiguousarray(off, dtype=np.int64)
        p_rows, p_wei, dev_flag = C.c_void_p(int(d_rows)), C.c_void_p(int(d_wei)), 1
    nb, na = len(off) - 1, panel.n_acc
    reuse = out
    out = {}
    for key, dtype in (("score", np.float64), ("ninfo", np.int64)) + ((("lik", np.float64), ("lrt", np.float64)) if likelihoods else ()):
        arr = reuse.get(key) if reuse is not None else None
        if not (isinstance(arr, np.ndarray) and arr.shape == (nb, na) and arr.dtype == dtype and arr.flags.c_contiguous and arr.flags.writeable):
            arr = np.empty((nb, na), dtype=dtype)
        out[key] = arr
    info = np.zeros(4, dtype=np.int64)
    if table is not None:
        assert device is None, "coded weights come from host memory"
        table = np.ascontiguousarray(table, dtype=np.float64)
        assert table.ndim == 1 and 1 <= len(table) <= 65536
        check(ctx.lib.snpm_score_batch_coded(panel.h, nb, ptr(off), p_rows, p_wei, ptr(table), len(table), int(chunk), int(bool(skip_hets)),
                                             int(mode), ptr(out["score"]), ptr(out["ninfo"]), ptr(out.get("lik")),
                                             ptr(out.get("lrt")), ptr(info)), ctx.h)
    else:
        check(ctx.lib.snpm_score_batch(panel.h, nb, ptr(off), p_rows, p_wei, dev_flag, int(chunk), int(bool(skip_hets)), int(mode),
                                       ptr(out["score"]), ptr(out["ninfo"]), ptr(out.get("lik")), ptr(out.get("lrt")), ptr(info)),
              ctx.h)
    out["pairs_reeval"], out["strict_fallback"] = int(info[0]), bool(info[1])
    out["shared_rows"], out["union_rows"] = bool(info[2]), int(info[3])
    return out


SHARED_WHY_NOT = {0: "", 1: "policy", 2: "too few samples or rows", 3: "a row list is not strictly increasing", 4: "a weight outside [0, 1]",
                  5: "call codes > 2 in the panel", 6: "overlap below the threshold", 7: "sizes beyond 32-bit indices", 8: "a row index outside the panel"}


def batch_configure(ctx, shared_rows=-1, digits=0, min_density=-1.0):
    """Policy of score_batch for batches whose samples share DB rows (snpm_batch_configure): shared_rows -1 automatic (device
    inputs, >= 4 samples, enough overlap), 0 never, 1 whenever the batch allows it; digits 3..7 base-256 digits of the fixed-point
    weights (-1: chosen by the longest sample, 0: keep); min_density: threshold of the automatic choice (negative: keep)."""
    check(ctx.lib.snpm_batch_configure(ctx.h, int(shared_rows), int(digits), float(min_density)), ctx.h)


def batch_last_stats(ctx):
    """What the context's last score_batch call did about shared rows (snpm_batch_last_stats)."""
    st = np.zeros(8, dtype=np.int64)
    check(ctx.lib.snpm_batch_last_stats(ctx.h, ptr(st)), ctx.h)
    return {"taken": bool(st[0]), "why_not": SHARED_WHY_NOT.get(int(st[1]), str(int(st[1]))), "union_rows": int(st[2]),
            "density": float(st[3]) / 1e6, "row_tiles": int(st[4]), "groups": int(st[5]), "passes": int(st[6]), "digits": int(st[7])}


class Carry(object):
    """Running per-accession totals of a job scored SNP slab after SNP slab (include/snpmatch_hip.h, snpm_carry_*)."""

    def __init__(self, ctx, n_acc):
        self.ctx, self.n_acc = ctx, int(n_acc)
        h = C.c_void_p()
        check(ctx.lib.snpm_carry_create(ctx.h, self.n_acc, C.byref(h)), ctx.h)
        self.h = h
        ctx._children.add(self)

    def reset(self):
        check(self.ctx.lib.snpm_carry_reset(self.h), self.ctx.h)

    def bind_outputs(self, d_score, d_ninfo):
        """keep the totals in caller-owned device buffers (raw pointers of float64 [n_acc] / int64 [n_acc])"""
        check(self.ctx.lib.snpm_carry_bind_outputs(self.h, C.c_void_p(d_score) if d_score else None,
                                                   C.c_void_p(d_ninfo) if d_ninfo else None), self.ctx.h)

    def set_columns(self, cols):
        cols = np.ascontiguousarray(cols, dtype=np.int32)
        check(self.ctx.lib.snpm_carry_set_columns(self.h, ptr(cols), len(cols)), self.ctx.h)

    def finish(self, want_results=True):
        """(score, ninfo, flagged accessions): flagged is empty unless the job ran in MODE_EXACT and some totals could
        not be certified; len(flagged) > 64 is reported as a count only (``n_flagged``)."""
        score = np.empty(self.n_acc, dtype=np.float64) if want_results else None
        ninfo = np.empty(self.n_acc, dtype=np.int64) if want_results else None
        cols = np.zeros(64, dtype=np.int32)
        nf = C.c_int64(0)
        check(self.ctx.lib.snpm_carry_finish(self.h, ptr(score), ptr(ninfo), ptr(cols), 64, C.byref(nf)), self.ctx.h)
        self.n_flagged = nf.value
        return score, ninfo, np.sort(cols[:min(nf.value, 64)])

    def error_bound(self):
        """bound on |total - reference total| of the slabs scored so far in MODE_EXACT (0 for all-integer jobs)"""
        b = C.c_double(0)
        check(self.ctx.lib.snpm_carry_error_bound(self.h, C.byref(b)), self.ctx.h)
        return b.value

    def patch_from(self, cols_carry):
        check(self.ctx.lib.snpm_carry_patch(self.h, cols_carry.h), self.ctx.h)

    def device_ptrs(self):
        ds, dn = C.c_void_p(), C.c_void_p()
        check(self.ctx.lib.snpm_carry_device_ptrs(self.h, C.byref(ds), C.byref(dn)), self.ctx.h)
        return ds.value, dn.value

    def free(self):
        if self.h:
            if self.ctx.h:
                self.ctx.lib.snpm_carry_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class SlabScorer(object):
    """Genotyper.genotyper over a SNP axis that does not fit in HBM (core/snpmatch.py:207-233 for a panel scored slab
    by slab).  ``slabs`` is a list of row counts; ``load(k, panel)`` must leave slab k in rows [0, slabs[k]) of
    ``panel`` (upload, file, generator); ``weights(k)`` returns the float64 [slabs[k], 3] weights of its rows (a numpy
    array, or a raw device pointer when ``device_weights``).  Every slab but the last holds a multiple of ``chunk``
    rows.  ``run`` returns (ScoreList, NumInfoSites, info); in MODE_EXACT the accessions whose total the certificate
    cannot vouch for are re-scored in reference order by streaming the slabs a second time."""

    def __init__(self, panel, slabs, load, weights, chunk=1000, skip_hets=False, device_weights=False):
        self.panel, self.slabs, self.load, self.weights = panel, [int(s) for s in slabs], load, weights
        self.chunk, self.skip, self.device_weights = int(chunk), bool(skip_hets), device_weights
        assert all(s % self.chunk == 0 for s in self.slabs[:-1]), "every slab but the last must hold a multiple of chunk rows"
        assert max(self.slabs) <= panel.n_snp
        self.queries = [None] * len(self.slabs)         # kept: a second pass reuses them
        self.carry = Carry(panel.ctx, panel.n_acc)

    def _query(self, k):
        if self.queries[k] is None:
            w = self.weights(k)
            self.queries[k] = (Query.from_device(self.panel, None, w, self.slabs[k]) if self.device_weights
                               else Query(self.panel, None, w))
        return self.queries[k]

    def _after(self, k):
        return sum(-(-s // self.chunk) for s in self.slabs[k + 1:])

    def _pass(self, carry, mode):
        for k in range(len(self.slabs)):
            self.load(k, self.panel)
            self._query(k).run_carry(carry, self.chunk, self.skip, mode, self._after(k))

    def run(self, mode=MODE_EXACT):
        self.carry.reset()
        self._pass(self.carry, mode)
        score, ninfo, flagged = self.carry.finish()
        info = {"n_strict_reeval": int(self.carry.n_flagged), "second_pass": False}
        if self.carry.n_flagged > 0:
            info["second_pass"] = True
            if self.carry.n_flagged <= 64:
                cols = Carry(self.panel.ctx, self.panel.n_acc)
                cols.set_columns(flagged)
                self._pass(cols, MODE_STRICT)
                self.carry.patch_from(cols)
                cols.free()
            else:                                   # many exact-integer totals: everything in reference order
                self.carry.reset()
                self._pass(self.carry, MODE_STRICT)
            score, ninfo, _ = self.carry.finish()
        return score, ninfo, info

    def free(self):
        for q in self.queries:
            if q is not None:
                q.free()
        self.carry.free()


# ----------------------------------------------------------------------------------------------------------
# Several GPUs: accession shards + ONE RCCL all-gather of the per-accession results, behind the C ABI
# (include/snpmatch_hip.h, snpm_group_*).  Reference: accession columns never interact (core/snpmatch.py:84-88); the
# likelihood step needs the minimum over all accessions (core/snpmatch.py:112).
class Group(object):
    """The GPUs of one job.  ``Group.local([0, 1, ...])``: this process drives all of them (ncclCommInitAll, no launcher);
    ``Group.from_rank(ctx, id_bytes, world, rank)``: one process per GPU (``Group.unique_id()`` on rank 0, handed to the
    other ranks by the caller).  ``loopback=True`` replaces RCCL by device-to-device copies: a test transport that also
    takes one device several times."""

    def __init__(self, handle, contexts, owns_contexts):
        self.lib = _lib.load()
        self.h = handle
        self.contexts = contexts
        self._owns_contexts = owns_contexts
        w, r0, nl = C.c_int(0), C.c_int(0), C.c_int(0)
        _lib.check_group(self.lib.snpm_group_info(self.h, C.byref(w), C.byref(r0), C.byref(nl)), self.h)
        self.world, self.rank0, self.n_local = w.value, r0.value, nl.value
        ref = weakref.ref(self)
        self._atexit = lambda: (ref() is not None) and ref().free()
        atexit.register(self._atexit)

    @staticmethod
    def unique_id():
        lib = _lib.load()
        buf = C.create_string_buffer(_lib.GROUP_ID_BYTES)
        _lib.check_group(lib.snpm_group_unique_id(buf), None)
        return buf.raw

    @classmethod
    def local(cls, device_ids, loopback=False):
        lib = _lib.load()
        ids = (C.c_int * len(device_ids))(*[int(d) for d in device_ids])
        h = C.c_void_p()
        _lib.check_group(lib.snpm_group_create_local(ids, len(device_ids), _lib.GROUP_LOOPBACK if loopback else 0, C.byref(h)), None)
        ctxs = []
        for i, d in enumerate(device_ids):
            ch = C.c_void_p()
            _lib.check_group(lib.snpm_group_ctx(h, i, C.byref(ch)), h)
            ctxs.append(Context(d, _borrowed=ch))
        return cls(h, ctxs, True)

    @classmethod
    def from_rank(cls, ctx, id_bytes, world, rank):
        lib = _lib.load()
        assert len(id_bytes) == _lib.GROUP_ID_BYTES
        h = C.c_void_p()
        _lib.check_group(lib.snpm_group_create_rank(ctx.h, C.c_char_p(bytes(id_bytes)), int(world), int(rank), C.byref(h)), None)
        return cls(h, [ctx], False)

    @property
    def transport(self):
        return (self.lib.snpm_group_transport(self.h) or b"").decode()

    def shard(self, n_acc, rank):
        a0, a1 = C.c_int64(0), C.c_int64(0)
        _lib.check_group(self.lib.snpm_group_shard(self.h, int(n_acc), int(rank), C.byref(a0), C.byref(a1)), self.h)
        return a0.value, a1.value

    def local_shards(self, n_acc):
        return [self.shard(n_acc, self.rank0 + i) for i in range(self.n_local)]

    def gather(self, d_scores, d_ninfos, n_acc, m=1, in_ld=None, truncate=False, host=True, likelihoods=False):
        """ONE all-gather of the members' shard results (raw device pointers, one per local member): returns host arrays
        score / ninfo [n_acc] ([m, n_acc] for m > 1) (+ lik, lrt); ``host=False``: nothing is copied or waited for, the
        gathered vectors stay on the devices (``gathered_ptrs``)."""
        n_acc, m = int(n_acc), int(m)
        if in_ld is None:
            in_ld = max(a1 - a0 for a0, a1 in self.local_shards(n_acc))
        ps = (C.c_void_p * self.n_local)(*[C.c_void_p(int(p or 0)) for p in d_scores])
        pn = (C.c_void_p * self.n_local)(*[C.c_void_p(int(p or 0)) for p in d_ninfos])
        shape = (n_acc,) if m == 1 else (m, n_acc)
        out = {}
        if host:
            out["score"] = np.empty(shape, dtype=np.float64)
            out["ninfo"] = np.empty(shape, dtype=np.int64)
            if likelihoods:
                out["lik"] = np.empty(shape, dtype=np.float64)
                out["lrt"] = np.empty(shape, dtype=np.float64)
        _lib.check_group(self.lib.snpm_group_gather_scores(self.h, ps, pn, m, n_acc, int(in_ld), int(bool(truncate)),
                                                           ptr(out.get("score")), ptr(out.get("ninfo")), ptr(out.get("lik")),
                                                           ptr(out.get("lrt"))), self.h)
        return out

    def gathered_ptrs(self, member=0):
        ds, dn = C.c_void_p(), C.c_void_p()
        _lib.check_group(self.lib.snpm_group_gathered_ptrs(self.h, int(member), C.byref(ds), C.byref(dn)), self.h)
        return ds.value, dn.value

    def free(self):
        if self.h:
            for c in self.contexts:
                if self._owns_contexts:
                    c.close()           # frees the member's panels / queries / carries; the group destroys the context itself
            self.lib.snpm_group_free(self.h)
            self.h = None
            try:
                atexit.unregister(self._atexit)
            except Exception:
                pass

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _on_members(fns):
    """run one callable per group member, each on its own host thread (a context is used by one thread at a time; ctypes
    releases the GIL, so the members' uploads and scoring calls proceed side by side)"""
    if len(fns) == 1:
        return [fns[0]()]
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=len(fns)) as pool:
        futures = [pool.submit(f) for f in fns]
        return [f.result() for f in futures]


class GroupPanel(object):
    """A DB whose accession columns are spread over the GPUs of a ``Group`` driven by this process: member i holds
    columns [a0_i, a1_i) of every SNP row as its own panel (``Panel`` or ``StreamedPanel``).  ``query`` / ``run`` have
    the signatures of ``Panel`` / ``Query`` and return full-length results."""

    def __init__(self, group, members, n_acc):
        assert group.rank0 == 0 and group.world == group.n_local == len(members), "GroupPanel: one process drives the whole group"
        self.group, self.members, self.n_acc = group, members, int(n_acc)
        self.bounds = group.local_shards(n_acc)
        assert all(m.n_acc == a1 - a0 for m, (a0, a1) in zip(members, self.bounds))
        self.n_snp = members[0].n_snp
        self.packed = members[0].packed
        self.ctx = members[0].ctx

    @staticmethod
    def usable_members(n_acc, n_devices):
        """how many members a DB of n_acc accessions can feed (every member needs at least one accession quad)"""
        n = max(1, min(int(n_devices), (int(n_acc) + 3) // 4))
        while n > 1 and (n - 1) * (((n_acc + n - 1) // n + 3) // 4 * 4) >= n_acc:
            n -= 1                      # the last shard would be empty
        return n

    @classmethod
    def build(cls, group, n_acc, make_member):
        """``make_member(ctx, a0, a1)`` -> the member's panel of columns [a0, a1); members are built side by side"""
        bounds = group.local_shards(n_acc)
        assert all(a1 > a0 for a0, a1 in bounds), "more GPUs than accession quads: use GroupPanel.usable_members"
        members = _on_members([(lambda c=c, b=b: make_member(c, b[0], b[1])) for c, b in zip(group.contexts, bounds)])
        return cls(group, members, n_acc)

    @classmethod
    def from_host(cls, group, snps, packed=False):
        return cls.build(group, snps.shape[1], lambda ctx, a0, a1: Panel.from_host(ctx, snps, packed=packed, cols=(a0, a1)))

    def query(self, row_idx, wei, row0=0):
        return GroupQuery(self, row_idx, wei, row0)

    def segregating_rows(self, cols):
        """Genotype.identify_segregating_snps over accessions that live on different GPUs: every member scans the listed
        accessions it holds (local mask + first informative call per row); a row segregates when some member saw two
        different calls or two members saw different ones"""
        cols = np.asarray(cols)
        parts = _on_members([(lambda m=m, b=b: m.segregating_first(cols[(cols >= b[0]) & (cols < b[1])] - b[0]))
                             for m, b in zip(self.members, self.bounds)])
        masks = np.stack([p[0] for p in parts])
        firsts = np.stack([p[1] for p in parts])
        seen = firsts != 0xFF
        lo = np.where(seen, firsts, 255).min(axis=0)
        hi = np.where(seen, firsts, 0).max(axis=0)
        return (masks.any(axis=0) | (seen.any(axis=0) & (lo != hi))).astype(np.uint8)

    def score_batch(self, samples, chunk=1000, skip_hets=False, mode=MODE_EXACT, likelihoods=True, device=None, table=None):
        """``score_batch`` with the accession axis spread over the members: every member scores all samples against its
        columns; the likelihood rows need the minimum over all accessions and are taken on the joined arrays"""
        assert device is None, "device-resident batch inputs belong to one GPU"
        parts = _on_members([(lambda m=m: score_batch(m, samples, chunk, skip_hets, mode, False, None, table)) for m in self.members])
        out = {"score": np.concatenate([p["score"] for p in parts], axis=1),
               "ninfo": np.concatenate([p["ninfo"] for p in parts], axis=1),
               "pairs_reeval": sum(p["pairs_reeval"] for p in parts),
               "strict_fallback": any(p["strict_fallback"] for p in parts)}
        if likelihoods:
            out["lik"], out["lrt"] = self.ctx.likelihood(out["score"], out["ninfo"], truncate=True)
        return out

    def free(self):
        for m in self.members:
            m.free()

    @property
    def h(self):
        """not None while the members' panels are alive (the test ``Genotype.panel`` uses to decide whether to rebuild)"""
        return self.members[0].h


class GroupQuery(object):
    """``Query`` against a ``GroupPanel``: the sample's rows and weights are replicated on every member, every member
    scores its accession shard, one all-gather (RCCL) brings the per-accession results together."""

    def __init__(self, gpanel, row_idx, wei, row0=0):
        self.panel = gpanel
        self.parts = _on_members([(lambda m=m: m.query(row_idx, wei, row0=row0)) for m in gpanel.members])
        self.n = self.parts[0].n
        self._wei = wei

    def run(self, chunk=1000, skip_hets=False, mode=MODE_EXACT, return_info=False):
        gp = self.panel
        ptrs = _on_members([(lambda q=q: q.run_device(chunk, skip_hets, mode)) for q in self.parts])
        out = gp.group.gather([p[0] for p in ptrs], [p[1] for p in ptrs], gp.n_acc)
        if return_info:
            n_re = sum(q.last_reeval() for q in self.parts) if mode == MODE_EXACT else 0
            return out["score"], out["ninfo"], {"n_strict_reeval": int(n_re), "members": len(self.parts),
                                                "transport": gp.group.transport}
        return out["score"], out["ninfo"]

    def run_windows(self, win_off, skip_hets=False, totals=True, fast=False):
        """per-window rows of every member side by side (host arrays: the window table is built on the host), totals
        through the all-gather"""
        res = _on_members([(lambda q=q: q.run_windows(win_off, skip_hets, totals=totals, fast=fast)) for q in self.parts])
        score = np.concatenate([r[0] for r in res], axis=1)
        ninfo = np.concatenate([r[1] for r in res], axis=1)
        tot_s = np.concatenate([r[2] for r in res])
        tot_n = np.concatenate([r[3] for r in res])
        infos = [getattr(q, "last_windows_info", None) for q in self.parts]
        if fast and all(i is not None for i in infos):
            self.last_windows_info = {"pairs_reeval": sum(i["pairs_reeval"] for i in infos),
                                      "totals_reeval": sum(i["totals_reeval"] for i in infos),
                                      "strict_fallback": any(i["strict_fallback"] for i in infos)}
        return score, ninfo, tot_s, tot_n

    def gather_columns(self, acc_idx):
        """calls of the accessions ``acc_idx`` (global indices, on whichever member holds them) at the matched rows"""
        acc_idx = np.asarray(acc_idx)
        out = np.full((len(acc_idx), self.n), 0xFF, dtype=np.uint8)
        for q, (a0, a1) in zip(self.parts, self.panel.bounds):
            mine = np.flatnonzero((acc_idx >= a0) & (acc_idx < a1))
            if len(mine):
                out[mine] = q.gather_columns(acc_idx[mine] - a0)
        return out

    def f1_pairs(self, acc_idx):
        """the in-silico crosses need the listed columns side by side: they are read where they live and crossed on a
        small panel of member 0 (core/csmatch.py:115-125)"""
        acc_idx = np.asarray(acc_idx)
        codes = self.gather_columns(acc_idx)
        ctx = self.panel.members[0].ctx
        small = Panel.from_host(ctx, np.ascontiguousarray(codes.T).view(np.int8))
        q = Query(small, None, self._wei)
        try:
            return q.f1_pairs(np.arange(len(acc_idx)))
        finally:
            q.free()
            small.free()

    def free(self):
        for q in self.parts:
            q.free()


# ----------------------------------------------------------------------------------------------------------
# DBs larger than the HBM budget: the reference streams any size through g.g.snps[idx, :] (core/snpmatch.py:218-225,
# pygwas/genotype.py:548-550).  Here two half-buffers alternate: while the matched rows of piece k are scored, the host
# threads read piece k + 1 from the file and the copy stream brings it in.
class StreamedPanel(object):
    """Columns ``cols`` of a DB (``RowStore``) that does not fit the HBM budget.  ``query`` has ``Panel.query``'s signature; a
    run walks the sample's matched rows in pieces that fit one half-buffer.  The pass is bound by the host link (PCIe
    ~50 GB/s against 5-6 TB/s of HBM), so every piece is scored in the reference's summation order (k_strict4): the fp64
    totals carry the reference's bits after ONE pass over the file, no certificate and no second pass."""

    def __init__(self, ctx, store, cols=None, packed=False, budget_bytes=None):
        self.ctx, self.store = ctx, store
        self.cols = (0, store.n_acc) if cols is None else (int(cols[0]), int(cols[1]))
        self.n_snp, self.n_acc, self.packed = store.n_snp, self.cols[1] - self.cols[0], bool(packed)
        pitch = ctx.row_pitch(self.n_acc, packed)
        if budget_bytes is None:
            budget_bytes = int(0.85 * ctx.mem_info()[0])
        self.rows_cap = int(budget_bytes // 2 // pitch) - 32             # the panel keeps 32 prefetch rows of its own
        assert self.rows_cap >= 1, "HBM budget too small for a single row of this DB"
        self.rows_cap = min(self.rows_cap, max(self.n_snp, 1))
        self.halves = [Panel(ctx, self.rows_cap, self.n_acc, packed=packed, n_acc_total=store.n_acc) for _ in range(2)]
        ctx._children.add(self)
        self._pinned = {}
        self.loads = 0              # pieces loaded so far (tests / timing)

    @property
    def h(self):
        return self.halves[0].h if self.halves else None

    def query(self, row_idx, wei, row0=0):
        return StreamedQuery(self, row_idx, wei, row0)

    def _load(self, which, rows):
        self.store.load(self.halves[which], self.cols, rows, 0)
        self.loads += 1

    def _pieces_of_all_rows(self):
        return [(r, min(r + self.rows_cap, self.n_snp)) for r in range(0, self.n_snp, self.rows_cap)]

    def segregating_first(self, cols):
        """(mask, first) over ALL DB rows (--refine scans the whole DB, core/snp_genotype.py:188-211): piece by piece"""
        masks, firsts = [], []
        for k, (r0, r1) in enumerate(self._pieces_of_all_rows()):
            self._load(k % 2, (r0, r1 - r0))
            m, f = self.halves[k % 2].segregating_first(cols)
            masks.append(m[:r1 - r0])
            firsts.append(f[:r1 - r0])
        return np.concatenate(masks), np.concatenate(firsts)

    def segregating_rows(self, cols):
        out = []
        for k, (r0, r1) in enumerate(self._pieces_of_all_rows()):
            self._load(k % 2, (r0, r1 - r0))
            out.append(self.halves[k % 2].segregating_rows(cols)[:r1 - r0])
        return np.concatenate(out)

    def pinned(self, key, shape, dtype):
        """grow-only pinned host arrays for results that arrive while the next piece is loaded"""
        need = int(np.prod(shape))
        have = self._pinned.get(key)
        if have is None or have.size < need:
            have = self._pinned[key] = self.ctx.pinned_empty((need,), dtype)
        return have[:need].reshape(shape)

    def free(self):
        for p in self.halves:
            p.free()
        self.halves = []
        self._pinned = {}


class StreamedQuery(object):
    """``Query`` against a ``StreamedPanel``: same calls, same results (reference order, fp64 bit for bit)."""

    def __init__(self, spanel, row_idx, wei, row0=0):
        wei = np.asarray(wei)
        assert wei.ndim == 2 and wei.shape[1] == 3, "SNP weights should be a np.array with  shape == n,3"
        self.panel, self.wei = spanel, np.ascontiguousarray(wei, dtype=np.float64)
        self.n = self.wei.shape[0]
        if row_idx is None:
            self.rows = np.arange(int(row0), int(row0) + self.n, dtype=np.int64)
        else:
            self.rows = np.ascontiguousarray(row_idx, dtype=np.int64)
            assert self.rows.shape == (self.n,), "please provide same number of positions for both sample and db"
        assert self.n == 0 or (self.rows.min() >= 0 and self.rows.max() < spanel.n_snp), "row index outside the panel"

    def _walk(self, bounds, score_piece):
        """load the matched rows [i0, i1) of every piece into alternating half-buffers and hand (half, i0, i1, k) to
        ``score_piece``, which only ENQUEUES device work: the next piece is read and copied in while it runs"""
        sp = self.panel
        if not bounds:
            return
        import os
        import time
        trace = [] if os.environ.get("SNPM_STREAM_TRACE") else None
        t0 = time.perf_counter()
        sp._load(0, self.rows[bounds[0][0]:bounds[0][1]])
        if trace is not None:
            trace.append(("load", 0, time.perf_counter() - t0))
        for k, (i0, i1) in enumerate(bounds):
            t0 = time.perf_counter()
            score_piece(sp.halves[k % 2], i0, i1, k)
            t1 = time.perf_counter()
            if k + 1 < len(bounds):
                j0, j1 = bounds[k + 1]
                sp._load((k + 1) % 2, self.rows[j0:j1])
            if trace is not None:
                trace.append(("enqueue", k, t1 - t0))
                trace.append(("load", k + 1, time.perf_counter() - t1))
        if trace is not None:
            import sys
            sys.stderr.write("stream trace: " + ", ".join("%s[%d] %.3f s" % t for t in trace) + "\n")

    def _run_carry(self, chunk, skip_hets, mode):
        sp = self.panel
        chunk = int(chunk)
        per = sp.rows_cap // chunk * chunk
        assert per >= chunk, "HBM budget too small for one %d-row chunk of this DB" % chunk
        bounds = [(i, min(i + per, self.n)) for i in range(0, self.n, per)]     # pieces of the MATCHED list, cut at chunk boundaries
        carry = Carry(sp.ctx, sp.n_acc)
        m = MODE_FAST if mode == MODE_FAST else MODE_STRICT
        live = []

        def score_piece(half, i0, i1, k):
            q = Query(half, None, self.wei[i0:i1])
            after = sum(-(-(b1 - b0) // chunk) for b0, b1 in bounds[k + 1:])
            q.run_carry(carry, chunk, skip_hets, m, after)
            live.append(q)
            if len(live) > 2:                   # the query of piece k - 2 has long finished
                live.pop(0).free()

        self._walk(bounds, score_piece)
        carry.finish(want_results=False)            # waits for the last piece
        for q in live:
            q.free()
        return carry, len(bounds)

    def run(self, chunk=1000, skip_hets=False, mode=MODE_EXACT, return_info=False):
        carry, n_pieces = self._run_carry(chunk, skip_hets, mode)
        score, ninfo, _ = carry.finish()
        carry.free()
        if return_info:
            return score, ninfo, {"n_strict_reeval": 0, "all_integer_weights": False, "reeval_path": 0, "pieces": n_pieces}
        return score, ninfo

    def run_device(self, chunk=1000, skip_hets=False, mode=MODE_EXACT):
        """results left on the device (raw pointers, valid until the next run / free): what a GroupQuery gathers"""
        if getattr(self, "_carry", None) is not None:
            self._carry.free()
        self._carry, _ = self._run_carry(chunk, skip_hets, mode)
        return self._carry.device_ptrs()

    def last_reeval(self):
        return 0

    def _window_pieces(self, win_off):
        sp = self.panel
        pieces, w0 = [], 0
        n_win = len(win_off) - 1
        while w0 < n_win:
            w1 = w0 + 1
            assert win_off[w1] - win_off[w0] <= sp.rows_cap, "HBM budget too small for the SNPs of one window"
            while w1 < n_win and win_off[w1 + 1] - win_off[w0] <= sp.rows_cap:
                w1 += 1
            pieces.append((w0, w1))
            w0 = w1
        return pieces

    def run_windows(self, win_off, skip_hets=False, totals=True, fast=False):
        """per-window matchGTsAccs over a streamed DB: pieces hold whole windows, the totals' chain continues in a carry"""
        sp = self.panel
        ctx = sp.ctx
        win_off = np.ascontiguousarray(win_off, dtype=np.int64)
        n_win = len(win_off) - 1
        p_score = sp.pinned("w_score", (max(n_win, 1), sp.n_acc), np.float64)
        p_ninfo = sp.pinned("w_ninfo", (max(n_win, 1), sp.n_acc), np.int64)
        carry = Carry(ctx, sp.n_acc)
        pieces = self._window_pieces(win_off)
        live = []

        def score_piece(half, i0, i1, k):
            w0, w1 = pieces[k]
            q = Query(half, None, self.wei[i0:i1])
            off = np.ascontiguousarray(win_off[w0:w1 + 1] - i0)
            check(ctx.lib.snpm_query_run_windows_carry(q.h, ptr(off), w1 - w0, int(bool(skip_hets)), ptr(p_score[w0:w1]),
                                                       ptr(p_ninfo[w0:w1]), carry.h), ctx.h)
            live.append(q)
            if len(live) > 2:
                live.pop(0).free()

        self._walk([(int(win_off[w0]), int(win_off[w1])) for w0, w1 in pieces], score_piece)
        tot_s, tot_n, _ = carry.finish()            # waits for the device: the pinned rows are complete
        for q in live:
            q.free()
        carry.free()
        return np.array(p_score[:n_win]), np.array(p_ninfo[:n_win]), tot_s, tot_n

    def gather_columns(self, acc_idx):
        sp = self.panel
        per = sp.rows_cap
        out = []

        def piece(half, i0, i1, k):
            q = Query(half, None, self.wei[i0:i1])
            out.append(q.gather_columns(acc_idx))
            q.free()

        self._walk([(i, min(i + per, self.n)) for i in range(0, self.n, per)], piece)
        return np.concatenate(out, axis=1) if out else np.zeros((len(acc_idx), 0), dtype=np.uint8)

    def f1_pairs(self, acc_idx):
        """in-silico crosses: the listed columns at the matched rows are read piece by piece, then crossed on a small panel"""
        acc_idx = np.asarray(acc_idx)
        codes = self.gather_columns(acc_idx)
        small = Panel.from_host(self.panel.ctx, np.ascontiguousarray(codes.T).view(np.int8))
        q = Query(small, None, self.wei)
        try:
            return q.f1_pairs(np.arange(len(acc_idx)))
        finally:
            q.free()
            small.free()

    def free(self):
        if getattr(self, "_carry", None) is not None:
            self._carry.free()
            self._carry = None
