"""
ctypes binding of oracle/liboracle.so (the plain-C CPU oracle) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
See oracle/snpmatch_oracle.c for the reference lines each entry point follows.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "snpmatch_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        _LIB = C.CDLL(so)
        i64, p = C.c_int64, C.c_void_p
        _LIB.oracle_match.argtypes = [p, p, i64, i64, i64, C.c_int, p, p]
        _LIB.oracle_genotyper.argtypes = [p, i64, i64, p, p, i64, i64, C.c_int, p, p]
        _LIB.oracle_windows.argtypes = [p, i64, i64, p, p, p, i64, C.c_int, p, p, p, p]
        _LIB.oracle_likelihood.argtypes = [p, p, i64, p, p]
    return _LIB


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def match(wei, db, skip_hets_db=False):
    """matchGTsAccs on a dense C-contiguous int8 [n, n_acc] block."""
    wei = np.ascontiguousarray(wei, dtype=np.float64)
    db = np.ascontiguousarray(db, dtype=np.int8)
    n, n_acc = db.shape
    assert wei.shape == (n, 3)
    score = np.empty(n_acc, dtype=np.float64)
    ninfo = np.empty(n_acc, dtype=np.int64)
    rc = lib().oracle_match(_ptr(wei), _ptr(db), n, n_acc, n_acc, int(skip_hets_db), _ptr(score), _ptr(ninfo))
    assert rc == 0
    return score, ninfo


def genotyper(panel, row_idx, wei, chunk=1000, skip_hets_db=False):
    """Chunked scoring of panel rows row_idx (None = all rows, dense)."""
    panel = np.ascontiguousarray(panel, dtype=np.int8)
    wei = np.ascontiguousarray(wei, dtype=np.float64)
    n_acc = panel.shape[1]
    if row_idx is not None:
        row_idx = np.ascontiguousarray(row_idx, dtype=np.int64)
        n = len(row_idx)
    else:
        n = panel.shape[0]
    assert wei.shape == (n, 3)
    score = np.empty(n_acc, dtype=np.float64)
    ninfo = np.empty(n_acc, dtype=np.int64)
    rc = lib().oracle_genotyper(_ptr(panel), n_acc, n_acc, _ptr(row_idx), _ptr(wei), n, chunk,
                                int(skip_hets_db), _ptr(score), _ptr(ninfo))
    assert rc == 0
    return score, ninfo


def windows(panel, row_idx, wei, win_off, skip_hets_db=False):
    panel = np.ascontiguousarray(panel, dtype=np.int8)
    wei = np.ascontiguousarray(wei, dtype=np.float64)
    win_off = np.ascontiguousarray(win_off, dtype=np.int64)
    n_acc = panel.shape[1]
    n_win = len(win_off) - 1
    if row_idx is not None:
        row_idx = np.ascontiguousarray(row_idx, dtype=np.int64)
    score = np.empty((n_win, n_acc), dtype=np.float64)
    ninfo = np.empty((n_win, n_acc), dtype=np.int64)
    tot_s = np.empty(n_acc, dtype=np.float64)
    tot_n = np.empty(n_acc, dtype=np.int64)
    rc = lib().oracle_windows(_ptr(panel), n_acc, n_acc, _ptr(row_idx), _ptr(wei), _ptr(win_off), n_win,
                              int(skip_hets_db), _ptr(score), _ptr(ninfo), _ptr(tot_s), _ptr(tot_n))
    assert rc == 0
    return score, ninfo, tot_s, tot_n


def likelihood(y, n):
    y = np.ascontiguousarray(y, dtype=np.float64)
    n = np.ascontiguousarray(n, dtype=np.int64)
    out = np.empty(len(y), dtype=np.float64)
    bad = C.c_int(0)
    lib().oracle_likelihood(_ptr(y), _ptr(n), len(y), _ptr(out), C.byref(bad))
    return out, bool(bad.value)
