#!/usr/bin/env python3
"""
One-off differential fuzz on MID-SIZE shapes (the seeded fuzzers of the test suite stay below 7000 SNPs x 2049 accessions): random
panels of up to 20 000 accessions x 600 000 SNPs, int8 / packed, dense / gathered, PL / hard-call samples, against the C oracle --
strict mode = fp64 bits, default mode = counts (+ the bound), and the one-call path for gathered samples.

    python tools/fuzz_large.py [n_cases [seed]] > gpurun_out/fuzz_large.txt
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 4401
    from oracle import c_oracle
    from snpmatch_amd import engine, synth
    rng = np.random.default_rng(seed)
    ctx = engine.Context(0)
    t0 = time.time()

    def bits(a):
        return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)

    for case in range(n_cases):
        n_acc = int(rng.choice([1, 2, 65, 257, 700, 1135, 1500, 2049, 3000, 4097, 5000, 8192, 10000, 12345, 20000]))
        n_snp = int(rng.integers(50_000, max(60_000, min(600_000, 4_000_000_000 // (n_acc * 8) + 60_000))))
        packed = bool(rng.integers(0, 2))
        skip = bool(rng.integers(0, 2))
        chunk = int(rng.choice([1000, 777, 4096, 10_000]))
        db = rng.choice(np.array([-1, 0, 1, 2], dtype=np.int8), size=(n_snp, n_acc), p=[0.05, 0.60, 0.33, 0.02])
        if rng.integers(0, 2):
            rows, n = None, n_snp
        else:
            n = int(rng.integers(1000, n_snp))
            rows = np.sort(rng.choice(n_snp, size=n, replace=False)).astype(np.int64)
        calls = rng.choice(np.array([0, 1, 2], dtype=np.int8), size=n, p=[0.6, 0.35, 0.05])
        wei = synth.sample_weights(rng, calls, float(rng.choice([0.0, 0.8, 1.0])))
        tag = "case %d: %d x %d packed=%s skip=%s chunk=%d %s n=%d" % (case, n_snp, n_acc, packed, skip, chunk, "dense" if rows is None else "gathered", n)
        panel = engine.Panel.from_host(ctx, db, packed=packed)
        q = engine.Query(panel, rows, wei)
        want_s, want_n = c_oracle.genotyper(db, rows, wei, chunk, skip)
        s, ni = q.run(chunk, skip, engine.MODE_STRICT)
        assert np.array_equal(bits(s), bits(want_s)) and np.array_equal(ni, want_n), tag
        s, ni, info = q.run(chunk, skip, engine.MODE_EXACT, return_info=True)
        assert np.array_equal(ni, want_n) and np.array_equal(s.astype(int), want_s.astype(int)), tag
        assert np.max(np.abs(s - want_s), initial=0.0) <= q.error_bound(chunk), tag
        if rows is not None:
            once = panel.genotype_once(rows, wei, None, chunk, skip, engine.MODE_EXACT)
            assert np.array_equal(once["ninfo"], want_n) and np.array_equal(once["score"].astype(int), want_s.astype(int)), tag
            table, inv = np.unique(wei.ravel(), return_inverse=True)
            if 1 <= len(table) <= 65536:
                once = panel.genotype_once(rows, inv.reshape(wei.shape).astype(np.uint16), None, chunk, skip, engine.MODE_STRICT, table=table)
                assert np.array_equal(bits(once["score"]), bits(want_s)) and np.array_equal(once["ninfo"], want_n), tag
        q.free()
        panel.free()
        print("ok %s  (re-evaluated %d, %.0f s)" % (tag, info["n_strict_reeval"], time.time() - t0), flush=True)
    print("done")


if __name__ == "__main__":
    main()
