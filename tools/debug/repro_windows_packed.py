#!/usr/bin/env python3
"""certified fast windows against strict windows on narrow packed panels: where do the informative counts differ?"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from snpmatch_amd import engine, synth  # noqa: E402

ctx = engine.Context(0)
rng = np.random.default_rng(5)
bad = 0
for case in range(40):
    n_snp = int(rng.choice([300, 2500, 9000, 20_000]))
    n_acc = int(rng.choice([64, 257, 1135, 1300, 2400]))
    db = rng.choice(np.array([-1, 0, 1, 2], dtype=np.int8), size=(n_snp, n_acc), p=[0.05, 0.60, 0.33, 0.02])
    n = n_snp
    rows = None
    if case % 2:
        n = int(rng.integers(1, n_snp + 1))
        rows = np.sort(rng.choice(n_snp, size=n, replace=False)).astype(np.int64)
    codes = rng.choice(np.array([0, 1, 2], dtype=np.int8), size=n, p=[0.6, 0.35, 0.05])
    wei = synth.sample_weights(rng, codes, 0.8)
    panel = engine.Panel.from_host(ctx, db, packed=True)
    q = engine.Query(panel, rows, wei)
    cuts = np.sort(rng.integers(0, n + 1, size=int(rng.integers(1, 9))))
    off = np.concatenate([[0], cuts, [n]]).astype(np.int64)
    ws, wn, ts, tn = q.run_windows(off, False)
    fs, fn, fts, ftn = q.run_windows(off, False, fast=True)
    d = np.argwhere(fn != wn)
    if len(d):
        bad += 1
        print("case %d: %d x %d n=%d gathered=%s off=%s" % (case, n_snp, n_acc, n, rows is not None, off.tolist()))
        print("  windows", sorted(set(d[:, 0].tolist())), "columns", sorted(set(d[:, 1].tolist()))[:40], "count", len(d))
        for w, c in d[:6]:
            print("   window %d (rows %d..%d) col %d: fast %d strict %d" % (w, off[w], off[w + 1], c, fn[w, c], wn[w, c]))
    q.free()
    panel.free()
print("bad cases:", bad)
