#!/usr/bin/env python3
"""samples/s against a resident 1001-Genomes-shaped panel (1135 accessions x 11M SNPs, 200k matched SNPs per sample):
one sample per call (round 1's path), batches through snpm_score_batch from host memory, and batches whose inputs
already are in device memory.  usage: tools/time_batch.py [B=32] [n_batches=8] [n_match=200000]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snpmatch_amd import engine, synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n_match = int(sys.argv[3]) if len(sys.argv) > 3 else 200_000
n_snp, n_acc, seed = 11_000_000, 1135, 1001
ctx = engine.Context(0)
panel = engine.Panel(ctx, n_snp, n_acc, packed=os.environ.get("PACKED", "0") == "1")     # PACKED=1: 2-bit packed panel
panel.fill_synthetic(seed)
rng = np.random.default_rng(1)
samples = []
for b in range(B):
    rows = np.sort(rng.choice(n_snp, size=n_match, replace=False)).astype(np.int64)
    col = synth.panel_rows(seed, rows, (b * 7 % n_acc) // 4 * 4, 4)[:, (b * 7 % n_acc) % 4]
    samples.append((rows, synth.planted_sample(rng, col, 0.02)[1]))
ctx.synchronize()

t0 = time.perf_counter()
for rows, wei in samples:
    q = engine.Query(panel, rows, wei)
    s, n = q.run(1000, False, engine.MODE_EXACT)
    lik, lrt = ctx.likelihood(s, n, truncate=True)
    q.free()
dt = time.perf_counter() - t0
print("one sample per call : %7.0f samples/s (%.3f ms per sample)" % (B / dt, dt / B * 1e3), flush=True)

off = np.concatenate([[0], np.cumsum([len(r) for r, _ in samples])]).astype(np.int64)
cat = (np.concatenate([r for r, _ in samples]), np.concatenate([w for _, w in samples]), off)
engine.score_batch(panel, cat)
t0 = time.perf_counter()
for _ in range(NB):
    out = engine.score_batch(panel, cat)
dt = time.perf_counter() - t0
print("batches of %3d, host : %7.0f samples/s (%.3f ms per sample), pairs re-evaluated per batch %d"
      % (B, B * NB / dt, dt / B / NB * 1e3, out["pairs_reeval"]), flush=True)
pin = (ctx.pinned_empty(cat[0].shape, np.int64), ctx.pinned_empty(cat[1].shape, np.float64), off)
pin[0][:] = cat[0]
pin[1][:] = cat[1]
engine.score_batch(panel, pin)
t0 = time.perf_counter()
for _ in range(NB):
    out = engine.score_batch(panel, pin)
dt = time.perf_counter() - t0
print("batches of %3d, pinned: %7.0f samples/s (%.3f ms per sample)" % (B, B * NB / dt, dt / B / NB * 1e3), flush=True)
tab = engine.pl_table()
codes = engine.weight_codes(cat[1], tab)
assert codes is not None
coded = (cat[0], codes, off)
outc = engine.score_batch(panel, coded, table=tab)
t0 = time.perf_counter()
for _ in range(NB):
    outc = engine.score_batch(panel, coded, table=tab)
dt = time.perf_counter() - t0
print("batches of %3d, coded : %7.0f samples/s (%.3f ms per sample)  [6 B of weight codes + 4 B of row index per SNP over PCIe]"
      % (B, B * NB / dt, dt / B / NB * 1e3), flush=True)
for k in ("score", "ninfo", "lik", "lrt"):
    assert np.array_equal(outc[k].view(np.uint64), out[k].view(np.uint64)), k
top = [int(np.nanargmin(out["lik"][b])) for b in range(B)]
assert top == [b * 7 % n_acc for b in range(B)], top

import torch  # noqa: E402
d_rows = torch.as_tensor(cat[0], device="cuda:0")
d_wei = torch.as_tensor(cat[1], device="cuda:0")
torch.cuda.synchronize()
dev = (d_rows.data_ptr(), d_wei.data_ptr(), off)
engine.score_batch(panel, None, device=dev)
ctx.profile(True)
ctx.profile_reset()
t0 = time.perf_counter()
for _ in range(NB):
    out2 = engine.score_batch(panel, None, device=dev)
dt = time.perf_counter() - t0
kf = ctx.profile_read("fast")
k_ms = kf[1] / NB                      # a batch may take several launches (runs)
row_bytes = n_acc / 4.0 if panel.packed else float(n_acc)
moved = B * n_match * (row_bytes + 32.0) / (k_ms * 1e-3) / 1e9
print("batches of %3d, HBM  : %7.0f samples/s (%.3f ms per sample); %s %.3f ms per batch = %.0f GB/s of %s bytes (row + 24 B weights + "
      "8 B index)%s"
      % (B, B * NB / dt, dt / B / NB * 1e3, "k_fast_packed_q4<GATHER, SEG>" if panel.packed else "k_fast<GATHER, SEG>", k_ms, moved,
         "packed" if panel.packed else "int8",
         "; int8-equivalent %.0f GB/s (not a bandwidth: what an int8 panel would have streamed)"
         % (B * n_match * (n_acc + 32.0) / (k_ms * 1e-3) / 1e9) if panel.packed else ""), flush=True)
assert np.array_equal(out2["ninfo"], out["ninfo"]) and np.array_equal(out2["score"].astype(np.int64), out["score"].astype(np.int64))
