"""A/B of k_fast_bits (hard calls on a packed panel) between two builds on the SAME box:
usage: python tools/ab/ab_bits.py LIB.so [N_ACC=10000] [N_SNP=50000000]"""
import ctypes as C
import sys
import numpy as np
lib = C.CDLL(sys.argv[1])
n_acc = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
n_snp = int(sys.argv[3]) if len(sys.argv) > 3 else 50_000_000
p, i64 = C.c_void_p, C.c_int64
lib.snpm_init.argtypes = [C.c_int, C.POINTER(p)]
lib.snpm_panel_create_packed.argtypes = [p, i64, i64, C.POINTER(p)]
lib.snpm_panel_fill_synthetic.argtypes = [p, C.c_uint64, i64, i64]
lib.snpm_query_create.argtypes = [p, p, i64, i64, p, C.POINTER(p)]
lib.snpm_query_run_device.argtypes = [p, i64, C.c_int, C.c_int, C.POINTER(p), C.POINTER(p), p]
lib.snpm_profile_enable.argtypes = [p, C.c_int]
lib.snpm_profile_reset.argtypes = [p]
lib.snpm_profile_read.argtypes = [p, C.c_char_p, C.POINTER(i64), C.POINTER(C.c_double)]
lib.snpm_synchronize.argtypes = [p]
ctx, panel, q = p(), p(), p()
assert lib.snpm_init(0, C.byref(ctx)) == 0
assert lib.snpm_panel_create_packed(ctx, n_snp, n_acc, C.byref(panel)) == 0
assert lib.snpm_panel_fill_synthetic(panel, 10050, 0, 0) == 0
rng = np.random.default_rng(1)
hard = len(sys.argv) <= 4 or sys.argv[4] != "pl"
if hard:
    wei = np.zeros((n_snp, 3))
    wei[np.arange(n_snp), rng.integers(0, 3, size=n_snp)] = 1.0
else:
    wei = np.exp(-rng.integers(0, 256, size=(n_snp, 3)) / 10.0)
assert lib.snpm_query_create(panel, None, 0, n_snp, wei.ctypes.data_as(p), C.byref(q)) == 0
ds, dn = p(), p()
for rep in range(3):
    for _ in range(2):
        lib.snpm_query_run_device(q, 1000, 0, 2, C.byref(ds), C.byref(dn), None)
    lib.snpm_synchronize(ctx)
    lib.snpm_profile_enable(ctx, 1)
    lib.snpm_profile_reset(ctx)
    for _ in range(6):
        lib.snpm_query_run_device(q, 1000, 0, 2, C.byref(ds), C.byref(dn), None)
    n, ms = i64(0), C.c_double(0)
    lib.snpm_profile_read(ctx, b"fast", C.byref(n), C.byref(ms))
    t = ms.value / n.value
    print("%s  packed %d x %d %s  fast pass %.3f ms  %.0f GB/s (%.3f of 8 TB/s)  %.3g comparisons/s" % (
        sys.argv[1].split("/")[-1], n_acc, n_snp, "hard" if hard else "PL", t, n_snp * (n_acc / 4 + 24.0) / t / 1e6,
        n_snp * (n_acc / 4 + 24.0) / t / 1e6 / 8000, n_snp * n_acc / t * 1e3), flush=True)
