"""
The reference's HDF5 DB files without h5py: a small read-only object layer over the library's native reader
(csrc/snpm_h5.cpp, include/snpmatch_hip.h ``snpm_h5_*``) with the handful of h5py idioms the reference uses on its DBs
(pygwas/genotype.py:534-673): ``f = File(path)``, ``f['snps']``, ``.shape``, ``.dtype``, ``.chunks``, ``ds[:]``,
``ds[a:b]``, ``ds[idx_array, :]``, ``ds[:, i]``, ``ds.attrs['chrs']``.  Host only -- no GPU is touched.
"""
import ctypes as C

import numpy as np

from . import _lib


def _check(rc, handle):
    if rc == _lib.SNPM_OK:
        return
    msg = _lib.load().snpm_h5_last_error(handle)
    msg = msg.decode("utf-8", "replace") if msg else ""
    if rc == _lib.SNPM_ERR_OOM:
        raise MemoryError(msg)
    raise IOError(msg)


def _dtype(type_class, elem_size, is_signed):
    if type_class == 0:
        return np.dtype("%s%d" % ("i" if is_signed else "u", elem_size))
    if type_class == 1:
        return np.dtype("f%d" % elem_size)
    if type_class == 3:
        return np.dtype("S%d" % max(elem_size, 1))
    raise IOError("HDF5 datatype class %d is not supported" % type_class)


class _Info(object):
    def __init__(self, f, path, attr=None):
        lib = f.lib
        kind, rank, tc, es, sg, na = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
        dims, chunk = (C.c_int64 * 8)(), (C.c_int64 * 8)()
        _check(lib.snpm_h5_info(f.h, path.encode(), attr.encode() if attr is not None else None, C.byref(kind), C.byref(rank), dims,
                                C.byref(tc), C.byref(es), C.byref(sg), chunk, C.byref(na)), f.h)
        self.is_data, self.rank, self.n_attrs = bool(kind.value), rank.value, na.value
        self.shape = tuple(int(dims[i]) for i in range(rank.value))
        self.chunks = tuple(int(chunk[i]) for i in range(rank.value)) if any(chunk[i] for i in range(rank.value)) else None
        self.dtype = _dtype(tc.value, es.value, sg.value) if self.is_data else None


class Attrs(object):
    """``dataset.attrs``: mapping attribute name -> numpy array (scalars for scalar attributes, bytes for single strings)"""

    def __init__(self, f, path):
        self._f, self._path = f, path

    def keys(self):
        n = _Info(self._f, self._path).n_attrs
        out = []
        buf = C.create_string_buffer(1024)
        for i in range(n):
            _check(self._f.lib.snpm_h5_attr_name(self._f.h, self._path.encode(), i, buf, len(buf)), self._f.h)
            out.append(buf.value.decode())
        return out

    def __contains__(self, name):
        return name in self.keys()

    def __iter__(self):
        return iter(self.keys())

    def __getitem__(self, name):
        if name not in self.keys():
            raise KeyError(name)
        info = _Info(self._f, self._path, name)
        out = np.empty(info.shape, dtype=info.dtype)
        _check(self._f.lib.snpm_h5_read(self._f.h, self._path.encode(), name.encode(), _lib.ptr(out), out.nbytes), self._f.h)
        return out[()] if out.ndim == 0 else out


class Dataset(object):
    def __init__(self, f, path):
        self._f, self.name = f, path
        info = _Info(f, path)
        self.shape, self.dtype, self.chunks = info.shape, info.dtype, info.chunks
        self.ndim = len(self.shape)
        self.attrs = Attrs(f, path)

    def __len__(self):
        return self.shape[0]

    @property
    def size(self):
        return int(np.prod(self.shape)) if self.shape else 1

    def read_rows(self, rows, col0=0, ncols=None):
        """rows ``rows`` (int array, or (first, n)), columns [col0, col0 + ncols) -> array [n, ncols]"""
        n_cols = self.shape[1] if self.ndim == 2 else 1
        ncols = n_cols - col0 if ncols is None else ncols
        es = self.dtype.itemsize
        if isinstance(rows, tuple):
            idx, first, n = None, int(rows[0]), int(rows[1])
        else:
            idx = np.ascontiguousarray(rows, dtype=np.int64)
            first, n = 0, len(idx)
            if n and (idx.min() < 0 or idx.max() >= self.shape[0]):
                raise IndexError("row index out of range")
        out = np.empty((n, ncols), dtype=self.dtype)
        _check(self._f.lib.snpm_h5_read_rows(self._f.h, self.name.encode(), _lib.ptr(idx), first, n, int(col0), int(ncols), _lib.ptr(out),
                                             ncols * es), self._f.h)
        return out

    def _axis(self, sel, length):
        """-> ('range', first, n) | ('list', int64 array) | ('one', i)"""
        if isinstance(sel, slice):
            a, b, step = sel.indices(length)
            if step != 1:
                return ("list", np.arange(a, b, step, dtype=np.int64))
            return ("range", a, max(b - a, 0))
        if isinstance(sel, (int, np.integer)):
            i = int(sel) + (length if sel < 0 else 0)
            if not 0 <= i < length:
                raise IndexError("index out of range")
            return ("one", i)
        arr = np.asarray(sel)
        if arr.dtype == bool:
            arr = np.flatnonzero(arr)
        arr = arr.astype(np.int64)
        arr = np.where(arr < 0, arr + length, arr)
        return ("list", arr)

    def __getitem__(self, key):
        if self.ndim == 0:
            out = np.empty((), dtype=self.dtype)
            _check(self._f.lib.snpm_h5_read(self._f.h, self.name.encode(), None, _lib.ptr(out), out.nbytes), self._f.h)
            return out[()]
        if self.dtype.kind == "S" and self.ndim <= 2:
            # string datasets (accession names) are small and may be variable-length in the file: read whole, index in numpy
            out = np.empty(self.shape, dtype=self.dtype)
            _check(self._f.lib.snpm_h5_read(self._f.h, self.name.encode(), None, _lib.ptr(out), out.nbytes), self._f.h)
            return out[key]
        if self.ndim > 2:
            if key is Ellipsis or key == slice(None):
                out = np.empty(self.shape, dtype=self.dtype)
                _check(self._f.lib.snpm_h5_read(self._f.h, self.name.encode(), None, _lib.ptr(out), out.nbytes), self._f.h)
                return out
            raise IOError("only [:] is supported on datasets of rank %d" % self.ndim)
        if key is Ellipsis:
            key = slice(None)
        if not isinstance(key, tuple):
            key = (key,)
        key = key + (slice(None),) * (self.ndim - len(key))
        rsel = self._axis(key[0], self.shape[0])
        if self.ndim == 1:
            rows = (rsel[1], rsel[2]) if rsel[0] == "range" else (np.array([rsel[1]]) if rsel[0] == "one" else rsel[1])
            out = self.read_rows(rows)[:, 0]
            return out[0] if rsel[0] == "one" else out
        csel = self._axis(key[1], self.shape[1])
        rows = (rsel[1], rsel[2]) if rsel[0] == "range" else (np.array([rsel[1]]) if rsel[0] == "one" else rsel[1])
        if csel[0] == "range":
            out = self.read_rows(rows, csel[1], csel[2])
        elif csel[0] == "one":
            out = self.read_rows(rows, csel[1], 1)
        else:
            cols = csel[1]
            lo, hi = (int(cols.min()), int(cols.max()) + 1) if len(cols) else (0, 0)
            out = self.read_rows(rows, lo, hi - lo)[:, cols - lo]
        if csel[0] == "one":
            out = out[:, 0]
        if rsel[0] == "one":
            out = out[0]
        return out

    def __array__(self, dtype=None, copy=None):
        a = self[:] if self.ndim else np.asarray(self[()])
        return a.astype(dtype) if dtype is not None else a


class File(object):
    """``h5py.File(path, 'r')`` for the reference's DB files"""

    def __init__(self, path, mode="r"):
        assert mode == "r", "the native HDF5 layer is read-only"
        self.lib = _lib.load()
        h = C.c_void_p()
        _check(self.lib.snpm_h5_open(str(path).encode(), C.byref(h)), None)
        self.h, self.filename = h, str(path)

    def keys(self, group=""):
        need = C.c_int64(0)
        _check(self.lib.snpm_h5_list(self.h, group.encode(), None, 0, C.byref(need)), self.h)
        buf = C.create_string_buffer(max(need.value, 1))
        _check(self.lib.snpm_h5_list(self.h, group.encode(), buf, len(buf), None), self.h)
        return [s for s in buf.value.decode().split("\n") if s]

    def __contains__(self, name):
        try:
            _Info(self, name)
            return True
        except IOError:
            return False

    def __getitem__(self, name):
        info = _Info(self, name)
        return Dataset(self, name) if info.is_data else Group(self, name)

    def close(self):
        if self.h:
            self.lib.snpm_h5_close(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Group(object):
    def __init__(self, f, path):
        self._f, self.name = f, path.strip("/")
        self.attrs = Attrs(f, path)

    def keys(self):
        return self._f.keys(self.name)

    def __getitem__(self, name):
        return self._f[self.name + "/" + name]
