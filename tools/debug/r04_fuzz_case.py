"""replay tests/test_gpu_parity.py::_fuzz_random_configurations up to one case and look at the one-call forms there"""
import os, sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from oracle import c_oracle
from snpmatch_amd import engine, synth
import test_gpu_parity as tp
target, seed = int(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 20260101
rng = np.random.default_rng(seed)
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
    if n > 0:
        rng.integers(0, n + 1, size=int(rng.integers(0, 12)))
print("case", target, n_snp, n_acc, packed, skip, chunk, kind, n)
want_s, want_n = c_oracle.genotyper(db, rows, wei, chunk, skip)
for name, env in (("fused", {}), ("unfused", {"SNPM_ONCE_FUSED": "0"}), ("fused zero-copy", {"SNPM_ONCE_ZEROCOPY": "1"})):
    os.environ.update(env); ctx = engine.Context(0)
    for k in env: del os.environ[k]
    panel = engine.Panel.from_host(ctx, db, packed=packed)
    for ch in (chunk, 7, 1000):
        ws, wn = c_oracle.genotyper(db, rows, wei, ch, skip)
        for mode in (engine.MODE_STRICT, engine.MODE_EXACT, engine.MODE_FAST):
            o = panel.genotype_once(rows, wei, None, ch, skip, mode)
            bad = np.flatnonzero(o["score"].astype(int) != ws.astype(int))
            print("%-16s chunk %4d mode %d: wrong counts %d of %d, max |diff| %.3g, ninfo ok %s, reeval %d" % (name, ch, mode, len(bad), n_acc, np.max(np.abs(o["score"] - ws)), np.array_equal(o["ninfo"], wn), o["n_strict_reeval"]))
    panel.free(); ctx.close()
