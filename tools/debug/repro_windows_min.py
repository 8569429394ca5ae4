#!/usr/bin/env python3
"""smallest dense window layouts on a 1135-wide packed panel whose certified fast pass disagrees with the strict one"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from snpmatch_amd import engine, synth  # noqa: E402

ctx = engine.Context(0)
rng = np.random.default_rng(5)
n_acc = int(os.environ.get("N_ACC", "1135"))
for n_snp in (32, 64, 96, 200, 256, 257, 300, 512, 513, 1000, 3000):
    db = rng.choice(np.array([-1, 0, 1, 2], dtype=np.int8), size=(n_snp, n_acc), p=[0.05, 0.60, 0.33, 0.02])
    codes = rng.choice(np.array([0, 1, 2], dtype=np.int8), size=n_snp, p=[0.6, 0.35, 0.05])
    wei = synth.sample_weights(rng, codes, 0.8)
    panel = engine.Panel.from_host(ctx, db, packed=True)
    q = engine.Query(panel, None, wei)
    for off in ([0, n_snp], [0, n_snp // 2, n_snp], [0, 1, n_snp], [0, n_snp // 3, 2 * n_snp // 3, n_snp]):
        off = np.array(off, dtype=np.int64)
        ws, wn, ts, tn = q.run_windows(off, False)
        fs, fn, fts, ftn = q.run_windows(off, False, fast=True)
        d = np.argwhere(fn != wn)
        msg = "ok"
        if len(d):
            cols = sorted(set(d[:, 1].tolist()))
            msg = "WRONG windows %s cols %d..%d (%d) e.g. w%d c%d fast %d strict %d" % (
                sorted(set(d[:, 0].tolist())), cols[0], cols[-1], len(cols), d[0][0], d[0][1], fn[d[0][0], d[0][1]], wn[d[0][0], d[0][1]])
        print("n_snp %5d off %-28s %s" % (n_snp, off.tolist(), msg), flush=True)
    q.free()
    panel.free()
