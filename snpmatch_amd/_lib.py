"""
ctypes binding of libsnpmatch_hip.so (include/snpmatch_hip.h).

The product path has NO CPU fallback: if the shared library is missing or no MI355X is
usable, the calls below raise.  (The CPU oracle lives in oracle/ and is test infrastructure.)
"""
import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SNPMATCH_HIP_LIB: load another build of the library (kernel experiments, system-wide installs)
LIB_PATH = os.environ.get("SNPMATCH_HIP_LIB") or os.path.join(_HERE, "libsnpmatch_hip.so")

SNPM_OK = 0
SNPM_ERR_BADARG = -1
SNPM_ERR_HIP = -2
SNPM_ERR_OOM = -3
SNPM_ERR_STATE = -4
SNPM_ERR_DOMAIN = -5
SNPM_ERR_RCCL = -6
GROUP_ID_BYTES = 128
GROUP_LOOPBACK = 1

MODE_EXACT = 0
MODE_STRICT = 1
MODE_FAST = 2

# every symbol include/snpmatch_hip.h declares (tests check that the .so exports all of them)
SYMBOLS = [
    "snpm_version", "snpm_hip_build_version", "snpm_build_id", "snpm_device_count", "snpm_init", "snpm_destroy", "snpm_last_error", "snpm_set_stream",
    "snpm_synchronize", "snpm_device_mem_info", "snpm_panel_create", "snpm_panel_create_packed", "snpm_panel_is_packed", "snpm_panel_set_total_accessions", "snpm_panel_row_pitch", "snpm_panel_free", "snpm_panel_info", "snpm_panel_upload_rows", "snpm_panel_load_file", "snpm_panel_load_file_rows", "snpm_panel_load_file_rows_packed", "snpm_pack_rows_host",
    "snpm_panel_upload_wait", "snpm_panel_download_rows", "snpm_panel_fill_synthetic", "snpm_query_create",
    "snpm_query_free", "snpm_query_bind_outputs", "snpm_query_run", "snpm_query_run_device", "snpm_query_error_bound",
    "snpm_query_run_windows", "snpm_query_run_windows_carry", "snpm_score_dense_host", "snpm_likelihood", "snpm_likelihood_device",
    "snpm_binom_identity", "snpm_binom_sf_host", "snpm_intersect_sorted", "snpm_panel_segregating",
    "snpm_query_f1_pairs", "snpm_intersect_sorted_search",
    "snpm_vcf_parse", "snpm_vcf_dims", "snpm_vcf_fill", "snpm_vcf_fill_u32", "snpm_vcf_sample_name", "snpm_vcf_free",
    "snpm_debug_stream_read", "snpm_profile_enable", "snpm_profile_reset", "snpm_profile_read",
    "snpm_panel_fill_synthetic_rows", "snpm_sample_synthetic", "snpm_query_create_device", "snpm_query_last_reeval",
    "snpm_query_last_kernel", "snpm_carry_create", "snpm_carry_reset", "snpm_carry_free", "snpm_carry_set_columns", "snpm_carry_bind_outputs", "snpm_panel_segregating_first", "snpm_query_gather_columns", "snpm_query_run_windows_fast", "snpm_score_batch", "snpm_score_batch_coded", "snpm_batch_configure", "snpm_batch_last_stats", "snpm_genotype_once", "snpm_genotype_once_coded", "snpm_host_alloc", "snpm_host_free",
    "snpm_query_run_carry", "snpm_carry_finish", "snpm_carry_patch", "snpm_carry_device_ptrs", "snpm_carry_error_bound",
    "snpm_group_unique_id", "snpm_group_create_rank", "snpm_group_create_local", "snpm_group_free", "snpm_group_last_error",
    "snpm_group_info", "snpm_group_ctx", "snpm_group_shard", "snpm_group_gather_scores", "snpm_group_gathered_ptrs",
    "snpm_group_transport",
    "snpm_h5_open", "snpm_h5_close", "snpm_h5_last_error", "snpm_h5_list", "snpm_h5_info", "snpm_h5_attr_name", "snpm_h5_read",
    "snpm_h5_read_rows", "snpm_panel_load_h5",
]

_lib = None


class SnpmError(RuntimeError):
    def __init__(self, code, msg):
        RuntimeError.__init__(self, "libsnpmatch_hip error %d: %s" % (code, msg))
        self.code = code
        self.msg = msg


HIP_RUNTIME = "system"          # which HIP runtime serves the library in this process: "system", "torch (already imported)" or "torch (preloaded)"


def _build_hip_major():
    """major HIP version libsnpmatch_hip.so was built against, read from the file without loading its dependencies"""
    import re
    try:
        with open(LIB_PATH, "rb") as fh:
            m = re.search(rb"libamdhip64\.so\.(\d+)", fh.read())
        return int(m.group(1)) if m else None
    except OSError:
        return None


def _torch_hip_major():
    """major HIP version of the installed PyTorch wheel, from torch/version.py (torch is NOT imported)"""
    import importlib.util
    import re
    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.origin:
        return None, None
    base = os.path.dirname(spec.origin)
    try:
        m = re.search(r"^hip\b[^=]*=\s*['\"](\d+)\.", open(os.path.join(base, "version.py")).read(), flags=re.M)
    except OSError:
        m = None
    return (int(m.group(1)) if m else None), os.path.join(base, "lib", "libamdhip64.so")


def _share_hip_runtime_with_torch():
    """PyTorch wheels bundle their own libamdhip64.so and load it by file name.  Imported first, torch's runtime also serves
    this library; imported AFTER this library was loaded, torch would bring a second HIP runtime into the process and
    find no GPU.  Preloading torch's copy (without importing torch) makes the order irrelevant -- but it also means the
    library, built with /opt/rocm's hipcc, runs on the wheel's runtime.  So the preload happens only when that runtime has
    the MAJOR version the library was linked against (its DT_NEEDED soname libamdhip64.so.N against ``hip`` in
    torch/version.py); otherwise, and with SNPMATCH_HIP_RUNTIME=system (what the CLI sets when it is not started by
    torch.distributed.run: it never imports torch), the system runtime is used.  SNPMATCH_HIP_RUNTIME=torch forces the
    preload.  The choice is kept in ``HIP_RUNTIME`` and logged."""
    global HIP_RUNTIME
    import logging
    log = logging.getLogger(__name__)
    choice = os.environ.get("SNPMATCH_HIP_RUNTIME", "").lower()
    if "torch" in sys.modules:
        HIP_RUNTIME = "torch (already imported)"
    elif choice != "system":
        try:
            major, path = _torch_hip_major()
            built = _build_hip_major()
            if path and os.path.exists(path) and (choice == "torch" or (major is not None and major == built)):
                C.CDLL(path, mode=C.RTLD_GLOBAL)
                HIP_RUNTIME = "torch (preloaded)"
            elif path and os.path.exists(path):
                log.warning("PyTorch bundles HIP %s, libsnpmatch_hip.so was built for HIP %s: using the system runtime; import "
                            "torch BEFORE snpmatch_amd if both are needed in one process", major, built)
        except Exception as e:          # noqa: BLE001
            log.warning("could not inspect PyTorch's HIP runtime (%s): using the system runtime", e)
    log.debug("HIP runtime for libsnpmatch_hip.so: %s", HIP_RUNTIME)


def load():
    """dlopen the HIP library; raises (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    _share_hip_runtime_with_torch()
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s not found: build it with ./build_lib.sh (hipcc --offload-arch=gfx950). "
            "snpmatch_amd has no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    i64, p, dbl, ci = C.c_int64, C.c_void_p, C.c_double, C.c_int
    pp = C.POINTER(C.c_void_p)
    lib.snpm_version.restype = ci
    lib.snpm_build_id.restype = C.c_char_p
    lib.snpm_device_count.argtypes = [C.POINTER(ci)]
    lib.snpm_init.argtypes = [ci, pp]
    lib.snpm_destroy.argtypes = [p]
    lib.snpm_last_error.argtypes = [p]
    lib.snpm_last_error.restype = C.c_char_p
    lib.snpm_set_stream.argtypes = [p, p]
    lib.snpm_synchronize.argtypes = [p]
    lib.snpm_device_mem_info.argtypes = [p, C.POINTER(i64), C.POINTER(i64)]
    lib.snpm_panel_create.argtypes = [p, i64, i64, pp]
    lib.snpm_panel_create_packed.argtypes = [p, i64, i64, pp]
    lib.snpm_panel_row_pitch.argtypes = [p, i64, C.c_int, C.POINTER(C.c_int64)]
    lib.snpm_panel_is_packed.argtypes = [p, C.POINTER(ci)]
    lib.snpm_panel_set_total_accessions.argtypes = [p, i64]
    lib.snpm_panel_free.argtypes = [p]
    lib.snpm_panel_info.argtypes = [p, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64), pp]
    lib.snpm_panel_upload_rows.argtypes = [p, i64, i64, p, i64]
    lib.snpm_panel_load_file.argtypes = [p, C.c_char_p, i64, i64, i64]
    lib.snpm_panel_load_file_rows.argtypes = [p, C.c_char_p, i64, i64, i64, p, i64, i64, i64]
    lib.snpm_panel_load_file_rows_packed.argtypes = [p, C.c_char_p, i64, i64, i64, p, i64, i64, i64]
    lib.snpm_pack_rows_host.argtypes = [p, i64, i64, i64, p, i64, ci, C.POINTER(ci)]
    lib.snpm_panel_upload_wait.argtypes = [p]
    lib.snpm_panel_download_rows.argtypes = [p, i64, i64, p, i64]
    lib.snpm_panel_fill_synthetic.argtypes = [p, C.c_uint64, i64, i64]
    lib.snpm_panel_fill_synthetic_rows.argtypes = [p, C.c_uint64, i64, i64, i64, i64]
    lib.snpm_sample_synthetic.argtypes = [p, C.c_uint64, i64, i64, i64, ci, ci, p, p]
    lib.snpm_query_create_device.argtypes = [p, p, i64, i64, p, pp]
    lib.snpm_query_last_reeval.argtypes = [p, C.POINTER(i64)]
    lib.snpm_query_last_kernel.argtypes = [p]
    lib.snpm_query_last_kernel.restype = C.c_char_p
    lib.snpm_carry_create.argtypes = [p, i64, pp]
    lib.snpm_carry_reset.argtypes = [p]
    lib.snpm_carry_free.argtypes = [p]
    lib.snpm_carry_set_columns.argtypes = [p, p, i64]
    lib.snpm_carry_bind_outputs.argtypes = [p, p, p]
    lib.snpm_query_run_carry.argtypes = [p, i64, ci, ci, i64, p]
    lib.snpm_carry_finish.argtypes = [p, p, p, p, i64, C.POINTER(i64)]
    lib.snpm_carry_patch.argtypes = [p, p]
    lib.snpm_carry_device_ptrs.argtypes = [p, pp, pp]
    lib.snpm_carry_error_bound.argtypes = [p, C.POINTER(dbl)]
    lib.snpm_group_unique_id.argtypes = [p]
    lib.snpm_group_create_rank.argtypes = [p, p, ci, ci, pp]
    lib.snpm_group_create_local.argtypes = [C.POINTER(ci), ci, ci, pp]
    lib.snpm_group_free.argtypes = [p]
    lib.snpm_group_last_error.argtypes = [p]
    lib.snpm_group_last_error.restype = C.c_char_p
    lib.snpm_group_info.argtypes = [p, C.POINTER(ci), C.POINTER(ci), C.POINTER(ci)]
    lib.snpm_group_ctx.argtypes = [p, ci, pp]
    lib.snpm_group_shard.argtypes = [p, i64, ci, C.POINTER(i64), C.POINTER(i64)]
    lib.snpm_group_gather_scores.argtypes = [p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), i64, i64, i64, ci, p, p, p, p]
    lib.snpm_group_gathered_ptrs.argtypes = [p, ci, pp, pp]
    lib.snpm_group_transport.argtypes = [p]
    lib.snpm_group_transport.restype = C.c_char_p
    lib.snpm_h5_open.argtypes = [C.c_char_p, pp]
    lib.snpm_h5_close.argtypes = [p]
    lib.snpm_h5_last_error.argtypes = [p]
    lib.snpm_h5_last_error.restype = C.c_char_p
    lib.snpm_h5_list.argtypes = [p, C.c_char_p, p, i64, C.POINTER(i64)]
    lib.snpm_h5_info.argtypes = [p, C.c_char_p, C.c_char_p, C.POINTER(ci), C.POINTER(ci), C.POINTER(i64), C.POINTER(ci),
                                 C.POINTER(ci), C.POINTER(ci), C.POINTER(i64), C.POINTER(ci)]
    lib.snpm_h5_attr_name.argtypes = [p, C.c_char_p, ci, p, i64]
    lib.snpm_h5_read.argtypes = [p, C.c_char_p, C.c_char_p, p, i64]
    lib.snpm_h5_read_rows.argtypes = [p, C.c_char_p, p, i64, i64, i64, i64, p, i64]
    lib.snpm_panel_load_h5.argtypes = [p, p, C.c_char_p, i64, p, i64, i64, i64]
    lib.snpm_query_create.argtypes = [p, p, i64, i64, p, pp]
    lib.snpm_query_free.argtypes = [p]
    lib.snpm_query_bind_outputs.argtypes = [p, p, p]
    lib.snpm_query_run.argtypes = [p, i64, ci, ci, p, p, p]
    lib.snpm_query_run_device.argtypes = [p, i64, ci, ci, pp, pp, p]
    lib.snpm_query_error_bound.argtypes = [p, i64, C.POINTER(dbl)]
    lib.snpm_query_run_windows.argtypes = [p, p, i64, ci, p, p, p, p]
    lib.snpm_query_run_windows_carry.argtypes = [p, p, i64, ci, p, p, p]
    lib.snpm_query_run_windows_fast.argtypes = [p, p, i64, ci, p, p, p, p, p]
    lib.snpm_score_batch_coded.argtypes = [p, i64, p, p, p, p, i64, i64, ci, ci, p, p, p, p, p]
    lib.snpm_genotype_once.argtypes = [p, p, p, p, i64, i64, i64, ci, ci, p, p, p, p, p]
    lib.snpm_genotype_once_coded.argtypes = [p, p, p, p, i64, p, i64, i64, i64, ci, ci, p, p, p, p, p]
    lib.snpm_host_alloc.argtypes = [p, i64, pp]
    lib.snpm_host_free.argtypes = [p, p]
    lib.snpm_score_batch.argtypes = [p, i64, p, p, p, ci, i64, ci, ci, p, p, p, p, p]
    lib.snpm_batch_configure.argtypes = [p, ci, ci, dbl]
    lib.snpm_batch_last_stats.argtypes = [p, p]
    lib.snpm_score_dense_host.argtypes = [p, p, i64, i64, i64, p, ci, p, p]
    lib.snpm_likelihood.argtypes = [p, p, p, i64, i64, ci, dbl, p, p]
    lib.snpm_likelihood_device.argtypes = [p, p, p, i64, i64, ci, dbl, p, p, C.POINTER(ci)]
    lib.snpm_binom_identity.argtypes = [p, p, p, i64, dbl, dbl, p, p]
    lib.snpm_binom_sf_host.argtypes = [p, p, i64, dbl, p]
    lib.snpm_intersect_sorted.argtypes = [p, i64, p, i64, p, p, C.POINTER(i64)]
    lib.snpm_panel_segregating.argtypes = [p, p, i64, p]
    lib.snpm_query_f1_pairs.argtypes = [p, p, ci, p, p]
    lib.snpm_panel_segregating_first.argtypes = [p, p, i64, p, p]
    lib.snpm_query_gather_columns.argtypes = [p, p, ci, p]
    lib.snpm_intersect_sorted_search.argtypes = [p, i64, p, i64, p, p, C.POINTER(i64)]
    lib.snpm_vcf_parse.argtypes = [C.c_char_p, ci, pp]
    lib.snpm_vcf_dims.argtypes = [p, C.POINTER(i64), C.POINTER(ci), C.POINTER(ci), C.POINTER(ci), C.POINTER(ci)]
    lib.snpm_vcf_fill.argtypes = [p, p, p, p, p, p]
    lib.snpm_vcf_fill_u32.argtypes = [p, p, p, p, p, p, p]
    lib.snpm_vcf_sample_name.argtypes = [p, ci]
    lib.snpm_vcf_sample_name.restype = C.c_char_p
    lib.snpm_vcf_free.argtypes = [p]
    lib.snpm_debug_stream_read.argtypes = [p, C.POINTER(i64)]
    lib.snpm_profile_enable.argtypes = [p, ci]
    lib.snpm_profile_reset.argtypes = [p]
    lib.snpm_profile_read.argtypes = [p, C.c_char_p, C.POINTER(i64), C.POINTER(dbl)]
    for name in SYMBOLS:          # fail at load time, not at first use, if the .so is stale
        getattr(lib, name)
    _lib = lib
    return lib


def build_id():
    """hash of the sources the loaded library was built from (12 hex digits)"""
    return load().snpm_build_id().decode()


def check_group(rc, group_handle=None):
    """status of a snpm_group_* call -> exception with the group's message"""
    if rc == SNPM_OK:
        return
    msg = load().snpm_group_last_error(group_handle)
    msg = msg.decode("utf-8", "replace") if msg else ""
    if rc == SNPM_ERR_BADARG:
        raise AssertionError(msg)
    if rc == SNPM_ERR_DOMAIN:
        raise AssertionError(msg or "provided y is greater than n")
    if rc == SNPM_ERR_OOM:
        raise MemoryError(msg)
    raise SnpmError(rc, msg)


def ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def check(rc, ctx_handle=None):
    if rc == SNPM_OK:
        return
    msg = load().snpm_last_error(ctx_handle)
    msg = msg.decode("utf-8", "replace") if msg else ""
    if rc == SNPM_ERR_BADARG:
        # the reference validates arguments with `assert` (core/snpmatch.py:75-76)
        raise AssertionError(msg)
    if rc == SNPM_ERR_DOMAIN:
        raise AssertionError(msg or "provided y is greater than n")     # core/snpmatch.py:43
    if rc == SNPM_ERR_OOM:
        raise MemoryError(msg)
    raise SnpmError(rc, msg)


def intersect_sorted(a, b, a_verified=False):
    """indices (ia, ib) of the common values of two strictly increasing integer arrays (native sorted merge, or
    a galloping search when ``a`` is known to be strictly increasing and much longer than ``b``);
    None when an input is not strictly increasing."""
    a = np.ascontiguousarray(a, dtype=np.int64)
    b = np.ascontiguousarray(b, dtype=np.int64)
    m = min(len(a), len(b))
    ia = np.empty(m, dtype=np.int64)
    ib = np.empty(m, dtype=np.int64)
    n = C.c_int64(0)
    fn = load().snpm_intersect_sorted_search if (a_verified and len(a) > 8 * len(b)) else load().snpm_intersect_sorted
    rc = fn(ptr(a), len(a), ptr(b), len(b), ptr(ia), ptr(ib), C.byref(n))
    if rc == SNPM_ERR_STATE:
        return None
    if rc != SNPM_OK:
        raise AssertionError("snpm_intersect_sorted: bad arguments")
    return ia[:n.value], ib[:n.value]


def vcf_parse(path, sample_index=0):
    """native VCF reader (snpm_vcf_*: blocks of lines parsed by a team of threads): dict with chr, pos, gt, pl, dp, flags, sample
    names; None when the library declines the file (the caller then uses the Python reader).  ``chr`` / ``gt`` arrive as numpy
    unicode arrays filled by the library (UTF-32, no conversion pass), ``called`` marks the records whose genotype is not './.' / '.|.'."""
    lib = load()
    h = C.c_void_p()
    rc = lib.snpm_vcf_parse(os.fsencode(path), int(sample_index), C.byref(h))
    if rc == SNPM_ERR_STATE:
        return None
    if rc != SNPM_OK:
        raise IOError("cannot read %s" % path)
    try:
        n, cw, gw, flags, ns = C.c_int64(0), C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
        lib.snpm_vcf_dims(h, C.byref(n), C.byref(cw), C.byref(gw), C.byref(flags), C.byref(ns))
        n = n.value
        pos = np.empty(n, dtype=np.int64)
        dp = np.empty(n, dtype=np.int64)
        pl = np.empty((n, 3), dtype=np.float64)
        called = None
        if flags.value & 8:
            chrom = np.empty(n, dtype="<U%d" % cw.value)
            gt = np.empty(n, dtype="<U%d" % gw.value)
            called = np.empty(n, dtype=np.uint8)
            rc = lib.snpm_vcf_fill_u32(h, ptr(chrom), ptr(pos), ptr(gt), ptr(pl), ptr(dp), ptr(called))
            assert rc == SNPM_OK
            called = called.view(np.bool_)
        else:
            chrom = np.zeros(n, dtype="S%d" % cw.value)
            gt = np.zeros(n, dtype="S%d" % gw.value)
            lib.snpm_vcf_fill(h, ptr(chrom), ptr(pos), ptr(gt), ptr(pl), ptr(dp))
            chrom, gt = chrom.astype("U"), gt.astype("U")
        names = [lib.snpm_vcf_sample_name(h, i).decode() for i in range(ns.value)]
    finally:
        lib.snpm_vcf_free(h)
    return {"chr": chrom, "pos": pos, "gt": gt, "pl": pl, "dp": dp, "names": names, "called": called,
            "has_gt": bool(flags.value & 1), "has_pl": bool(flags.value & 2), "has_dp": bool(flags.value & 4)}


def pack_rows_host(snps):
    """int8 calls [n, n_acc] -> uint8 [n, (n_acc + 3) // 4], 2 bits per call (the loader's host packer); AssertionError when a
    call code other than -1 (any negative), 0, 1, 2 is met"""
    snps = np.ascontiguousarray(snps, dtype=np.int8)
    n, n_acc = snps.shape
    out = np.empty((n, (n_acc + 3) // 4), dtype=np.uint8)
    bad = C.c_int(0)
    rc = load().snpm_pack_rows_host(ptr(snps), n_acc, n, n_acc, ptr(out), out.shape[1], 0, C.byref(bad))
    assert rc == SNPM_OK
    assert not bad.value, "a packed panel holds only the codes -1 (any negative), 0, 1, 2; use the int8 format for other values"
    return out


def binom_sf_host(k, n, p):
    k = np.ascontiguousarray(k, dtype=np.float64)
    n = np.ascontiguousarray(n, dtype=np.float64)
    out = np.empty(len(k), dtype=np.float64)
    rc = load().snpm_binom_sf_host(ptr(k), ptr(n), len(k), float(p), ptr(out))
    assert rc == SNPM_OK
    return out


def as_f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        assert a.shape == shape
    return a
