#!/usr/bin/env python3
"""The shared-row scan of a batch on the 1001-Genomes shape (1135 accessions x 11M SNPs resident): B samples on ONE marker set
(each lacking `drop` of it), inputs in device memory.  Prints wall time per call for the per-sample pass and the shared-row scan
and the library's per-stage event times; run it under `rocprofv3 --kernel-trace --stats` for the per-kernel breakdown.
usage: tools/time_shared.py [B=64] [n_match=200000] [reps=10] [digits=0] [packed=0] [drop=0.03]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snpmatch_amd import engine, synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n_match = int(sys.argv[2]) if len(sys.argv) > 2 else 200_000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
digits = int(sys.argv[4]) if len(sys.argv) > 4 else 0
packed = len(sys.argv) > 5 and sys.argv[5] == "1"
drop = float(sys.argv[6]) if len(sys.argv) > 6 else 0.03
n_snp, n_acc, seed = 11_000_000, 1135, 1001
import torch  # noqa: E402

ctx = engine.Context(0)
panel = engine.Panel(ctx, n_snp, n_acc, packed=packed)
panel.fill_synthetic(seed)
rng = np.random.default_rng(3)
base = np.sort(rng.choice(n_snp, size=n_match, replace=False)).astype(np.int64)
rows_l, wei_l, accs = [], [], []
for b in range(B):
    rows = base[rng.random(n_match) >= drop]
    acc = (b * 11 + 417) % n_acc
    col = synth.panel_rows(seed, rows, acc // 4 * 4, 4)[:, acc % 4]
    rows_l.append(rows)
    wei_l.append(synth.planted_sample(rng, col, 0.02)[1])
    accs.append(acc)
off = np.concatenate([[0], np.cumsum([len(r) for r in rows_l])]).astype(np.int64)
d_rows = torch.as_tensor(np.concatenate(rows_l), device="cuda:0")
d_wei = torch.as_tensor(np.concatenate(wei_l), device="cuda:0")
torch.cuda.synchronize()
dev = (d_rows.data_ptr(), d_wei.data_ptr(), off)
ref = None
for policy, name in ((0, "per-sample pass"), (1, "shared-row scan")):
    engine.batch_configure(ctx, shared_rows=policy, digits=digits)
    out = engine.score_batch(panel, None, device=dev)
    ctx.synchronize()
    each = []
    reuse = os.environ.get("TIME_SHARED_REUSE_OUTPUTS", "1") != "0"      # the result arrays of the previous call are written again (0: fresh arrays per call)
    for _ in range(reps):
        t0 = time.perf_counter()
        out = engine.score_batch(panel, None, device=dev, out=out if reuse else None)
        each.append(time.perf_counter() - t0)
    ctx.profile(True)
    ctx.profile_reset()
    for _ in range(reps):
        engine.score_batch(panel, None, device=dev)
    ctx.synchronize()
    parts = {k: ctx.profile_read(k) for k in ("lut", "fast", "reduce", "strict", "scan", "likelihood")}
    ctx.profile(False)
    assert [int(np.nanargmin(out["lik"][b])) for b in range(B)] == accs
    if ref is None:
        ref = out
    else:
        assert np.array_equal(ref["ninfo"], out["ninfo"]) and np.array_equal(ref["score"].astype(np.int64), out["score"].astype(np.int64))
    print("%-16s wall median %.3f ms (min %.3f) = %.0f samples/s; events per call: %s; pairs re-scored %d; %s"
          % (name, np.median(each) * 1e3, min(each) * 1e3, B / np.median(each),
             " ".join("%s %.3f" % (k, v[1] / reps) for k, v in parts.items() if v[0]), out["pairs_reeval"],
             engine.batch_last_stats(ctx) if policy else ""), flush=True)
ctx.close()
