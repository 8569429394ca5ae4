"""the fuzzer's own sequence of GPU calls up to one case, then the one-call forms there, repeated"""
import os, sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from oracle import c_oracle
from snpmatch_amd import engine
import test_gpu_parity as tp
target, seed = int(sys.argv[1]), 20260101
first_once = int(sys.argv[2]) if len(sys.argv) > 2 else 0          # cases before this one skip their one-call part
rng = np.random.default_rng(seed)
ctx = engine.Context(0)
for case in range(target + 1):
    n_snp = int(rng.integers(1, 7000)); n_acc = int(rng.choice([1, 2, 3, 5, 63, 64, 65, 255, 256, 257, 700, 1135, 2049]))
    packed = bool(rng.integers(0, 2)); skip = bool(rng.integers(0, 2)); chunk = int(rng.choice([1, 7, 128, 1000, 1001, 5000]))
    db = tp.rand_db(rng, n_snp, n_acc)
    if not packed and rng.random() < 0.3:
        db[rng.integers(0, n_snp), :] = 3
    kind = rng.integers(0, 3)
    if kind == 0:
        rows, n = None, n_snp
    elif kind == 1:
        n = int(rng.integers(0, n_snp + 1)); rows = np.sort(rng.choice(n_snp, size=n, replace=False)).astype(np.int64)
    else:
        n = int(rng.integers(1, 2 * n_snp + 2)); rows = rng.integers(0, n_snp, size=n).astype(np.int64)
    wei = tp.rand_wei(rng, n, frac_pl=float(rng.choice([0.0, 0.5, 1.0])))
    if n > 3:
        wei[rng.integers(0, n)] = 0.0
    panel = engine.Panel.from_host(ctx, db, packed=packed)
    q = engine.Query(panel, rows, wei)
    want_s, want_n = c_oracle.genotyper(db, rows, wei, chunk, skip)
    q.run(chunk, skip, engine.MODE_STRICT)
    s, ni = q.run(chunk, skip, engine.MODE_EXACT)
    q.error_bound(chunk)
    if n > 0:
        cuts = np.sort(rng.integers(0, n + 1, size=int(rng.integers(0, 12))))
        off = np.concatenate([[0], cuts, [n]]).astype(np.int64)
        q.run_windows(off, skip)
    if rows is not None and np.all(wei <= 1.0) and case >= first_once:
        print("case %d: %dx%d packed=%s skip=%s chunk=%d kind=%d n=%d" % (case, n_snp, n_acc, packed, skip, chunk, kind, n), flush=True)
        table, inv = np.unique(wei.ravel(), return_inverse=True)
        for rep in range(3 if case == target else 1):
            for coded in (False, True):
                for mode in (engine.MODE_STRICT, engine.MODE_EXACT):
                    if coded:
                        o = panel.genotype_once(rows, inv.reshape(wei.shape).astype(np.uint16), None, chunk, skip, mode, table=table)
                    else:
                        o = panel.genotype_once(rows, wei, None, chunk, skip, mode)
                    bad = np.flatnonzero(o["score"].astype(int) != want_s.astype(int))
                    if len(bad) or case == target:
                        print("   rep %d coded %d mode %d: wrong counts %d of %d, max |diff| %.3g, ninfo ok %s, reeval %d" % (rep, coded, mode, len(bad), n_acc, np.max(np.abs(o["score"] - want_s), initial=0), np.array_equal(o["ninfo"], want_n), o["n_strict_reeval"]), flush=True)
    q.free(); panel.free()
