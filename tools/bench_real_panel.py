#!/usr/bin/env python3
"""
The real-panel configurations of BASELINE.json as bench legs: configs[1] (1001 Genomes shape, 1135 accessions x 11M SNPs
resident, one 200k-SNP sample: the GATHERED access shape of Genotyper.genotyper, core/snpmatch.py:218-225), the same for a
batch of 64 samples whose inputs already are in HBM, and configs[2] (the 399 windows of `snpmatch cross` at 300 kb,
core/csmatch.py:80-95) -- on the int8 panel and on the 2-bit packed panel (the drop-in classes' default residency).

bench.py runs this as a child process after the headline job and puts the JSON it prints under "real_panel"; run alone
(optionally under rocprofv3) it is the workload of the profiles/r04_real_panel_* summaries.

Per leg: the scoring kernel's name, its average duration from HIP events on the library's stream (snpm_profile_*), the
ALGORITHMIC bytes of one launch = matched rows x (row bytes + 24 B of fp64 weights + 8 B of row index), where a row is
n_acc bytes (int8) or n_acc / 4 bytes (packed), and the fraction of the 8 TB/s HBM peak those bytes per second are.  Packed
legs also state the int8-equivalent rate (what an int8 panel would have had to stream for the same comparisons): that
figure may exceed the chip's bandwidth and is labelled as such.

usage: tools/bench_real_panel.py [--formats int8,packed] [--reps 20] [--batch 64] [--json-only]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from snpmatch_amd import engine, synth  # noqa: E402
from snpmatch_amd.core import genomes  # noqa: E402

HBM_PEAK_GBS = 8000.0
N_SNP, N_ACC, SEED, PLANTED, N_MATCH = 11_000_000, 1135, 1001, 417, 200_000


def tair10_layout(n_snp):
    """DB positions spread over the five TAIR10 chromosomes in proportion to their lengths"""
    g0 = genomes.Genome("athaliana_tair10")
    frac = np.cumsum(g0.chrlen) / g0.chrlen.sum()
    bounds = np.concatenate([[0], np.round(frac * n_snp).astype(np.int64)])
    positions = np.concatenate([1 + (np.arange(bounds[c + 1] - bounds[c]) * int(g0.chrlen[c] - 1)) // int(bounds[c + 1] - bounds[c])
                                for c in range(5)])
    return g0, bounds, positions


def window_offsets(g0, bounds, positions, rows, bin_len=300000):
    """offsets of the 300-kb windows (core/genomes.py:111-116: [1 + k b, (k + 1) b]) inside the sorted matched-row list"""
    off = [0]
    for c in range(5):
        lo, hi = np.searchsorted(rows, bounds[c]), np.searchsorted(rows, bounds[c + 1])
        pos = positions[rows[lo:hi]]
        nb = len(range(1, int(g0.chrlen[c]), bin_len))
        edges = np.searchsorted(pos, 1 + bin_len * np.arange(1, nb + 1), side="left")
        off.extend((lo + edges).tolist())
    return np.asarray(off, dtype=np.int64)


def kernel_leg(ctx, name, run, n_launch_rows, row_bytes, reps, kernel="fast"):
    """time `run` reps times (wall clock, the library's event profiling off), then reps times more with it on: kernel = average
    HIP-event duration of the scoring kernel per launch (the events cost a short call 0.03-0.06 ms of wall time)"""
    run()
    ctx.synchronize()
    each = []
    t0 = time.perf_counter()
    for _ in range(reps):
        t1 = time.perf_counter()
        run()
        each.append(time.perf_counter() - t1)
    ctx.synchronize()
    mean_wall = (time.perf_counter() - t0) / reps
    wall = float(np.median(each))          # the median: once per few hundred launches the HIP runtime stalls a call for tens of ms
    if os.environ.get("BENCH_REAL_PANEL_TRACE"):
        sys.stderr.write("%s: %s ms\n" % (name, " ".join("%.2f" % (t * 1e3) for t in each)))
    ctx.profile(True)
    ctx.profile_reset()
    for _ in range(reps):
        run()
    ctx.synchronize()
    launches, ms = ctx.profile_read(kernel)
    parts = {k: ctx.profile_read(k) for k in ("lut", "fast", "reduce", "strict", "scan", "likelihood")}
    ctx.profile(False)
    per_call_ms = ms / reps                       # a call may take several launches (runs of a batch)
    alg = float(n_launch_rows) * (row_bytes + 24.0 + 8.0)
    rate = alg / (per_call_ms * 1e-3) / 1e9 if per_call_ms > 0 else 0.0
    return {"leg": name, "wall_ms_per_call": wall * 1e3, "wall_ms_mean": mean_wall * 1e3, "wall_ms_max": max(each) * 1e3, "kernel_ms_per_call": per_call_ms, "kernel_launches_per_call": launches / reps,
            "algorithmic_bytes_per_call": alg, "achieved_GBs": rate, "frac_of_hbm_peak": rate / HBM_PEAK_GBS,
            "other_kernels_ms_per_call": {k: v[1] / reps for k, v in parts.items() if v[0] and k != kernel}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--formats", default="int8,packed")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--n-snp", type=int, default=N_SNP)
    ap.add_argument("--n-match", type=int, default=N_MATCH)
    args = ap.parse_args()
    import torch

    n_snp, n_match, B = args.n_snp, args.n_match, args.batch
    ctx = engine.Context(0)
    g0, bounds, positions = tair10_layout(n_snp)
    rng = np.random.default_rng(5)
    samples = []
    for b in range(B):
        rows = np.sort(rng.choice(n_snp, size=n_match, replace=False)).astype(np.int64)
        acc = PLANTED if b == 0 else (b * 7) % N_ACC
        col = synth.panel_rows(SEED, rows, acc // 4 * 4, 4)[:, acc % 4]
        samples.append((rows, synth.planted_sample(rng, col, 0.02)[1], acc))
    # the same number of samples on ONE marker set of n_match rows, each lacking 3 % of it (a SNP chip / a fixed capture panel)
    base = np.sort(rng.choice(n_snp, size=n_match, replace=False)).astype(np.int64)
    c_rows, c_wei, c_accs = [], [], []
    for b in range(B):
        rows = base[rng.random(n_match) >= 0.03]
        acc = PLANTED if b == 0 else (b * 11) % N_ACC
        col = synth.panel_rows(SEED, rows, acc // 4 * 4, 4)[:, acc % 4]
        c_rows.append(rows)
        c_wei.append(synth.planted_sample(rng, col, 0.02)[1])
        c_accs.append(acc)
    chip_batches = [{"rows": np.concatenate(c_rows), "wei": np.concatenate(c_wei), "accs": c_accs,
                     "off": np.concatenate([[0], np.cumsum([len(r) for r in c_rows])]).astype(np.int64)}]
    rows0, wei0, _ = samples[0]
    win_off = window_offsets(g0, bounds, positions, rows0)
    off = np.concatenate([[0], np.cumsum([len(r) for r, _, _ in samples])]).astype(np.int64)
    cat_rows = np.concatenate([r for r, _, _ in samples])
    cat_wei = np.concatenate([w for _, w, _ in samples])
    out = {"workload": "configs[1] / configs[2]: synthetic %d accessions x %d SNPs resident (seed %d), %d-SNP samples (planted accession, 2%% "
                       "error, 80%% PL weights), %d windows of 300 kb" % (N_ACC, n_snp, SEED, n_match, len(win_off) - 1),
           "hbm_peak_GBs": HBM_PEAK_GBS, "formats": {}}
    for fmt in args.formats.split(","):
        packed = fmt == "packed"
        t0 = time.perf_counter()
        panel = engine.Panel(ctx, n_snp, N_ACC, packed=packed)
        panel.fill_synthetic(SEED)
        ctx.synchronize()
        row_bytes = N_ACC / 4.0 if packed else float(N_ACC)
        legs = []
        # (1) ONE gathered 200k-row sample per call, inputs from host memory: query + run + likelihood (what Genotyper.genotyper does)
        lik_top = [None]

        def one_sample():
            q = engine.Query(panel, rows0, wei0)
            s, n = q.run(1000, False, engine.MODE_EXACT)
            lik, _ = ctx.likelihood(s, n, truncate=True)
            lik_top[0] = int(np.nanargmin(lik))
            q.free()

        leg = kernel_leg(ctx, "single_sample_200k_three_calls", one_sample, n_match, row_bytes, args.reps)
        assert lik_top[0] == PLANTED
        leg["kernel"] = "k_fast_packed_q4<GATHER>" if packed else "k_fast<GATHER>"
        leg["note"] = "round 3's path: snpm_query_create + snpm_query_run + snpm_likelihood, weights gathered by the caller"
        legs.append(leg)
        # (1b) the same sample through snpm_genotype_once: the library gathers wei_all[sample_idx] itself (what Genotyper.genotyper
        # hands over: the whole sample's weights + the matched positions), counts and likelihoods come back in one copy
        n_in = n_match + n_match // 4                            # a sample with 25 % of its SNPs absent from the DB
        sidx = np.sort(rng.choice(n_in, size=n_match, replace=False)).astype(np.int64)
        wei_all = np.zeros((n_in, 3))
        wei_all[sidx] = wei0

        def one_call():
            lik_top[0] = int(np.nanargmin(panel.genotype_once(rows0, wei_all, sidx, 1000, False, engine.MODE_EXACT)["lik"]))

        leg = kernel_leg(ctx, "single_sample_200k_one_call", one_call, n_match, row_bytes, args.reps)
        assert lik_top[0] == PLANTED
        leg["kernel"] = "k_fast_packed_q4<GATHER>" if packed else "k_fast<GATHER>"
        leg["note"] = "snpm_genotype_once: host-thread gather into a pinned slab, one upload, one copy back (Genotyper.genotyper's path)"
        legs.append(leg)
        # (1c) ... and with the weights as dictionary codes, as ParseInputs keeps them for a parsed VCF (exp(-PL / 10) of integer PLs):
        # 10 instead of 32 bytes per matched SNP cross PCIe (snpm_genotype_once_coded)
        tab = engine.pl_table(256)
        tab = np.concatenate([tab, [0.0]])
        codes_all = engine.weight_codes(wei_all, tab)
        if codes_all is not None:
            def one_call_coded():
                lik_top[0] = int(np.nanargmin(panel.genotype_once(rows0, codes_all, sidx, 1000, False, engine.MODE_EXACT, table=tab)["lik"]))

            leg = kernel_leg(ctx, "single_sample_200k_one_call_coded_weights", one_call_coded, n_match, row_bytes, args.reps)
            assert lik_top[0] == PLANTED
            leg["kernel"] = "k_fast_packed_q4<GATHER>" if packed else "k_fast<GATHER>"
            leg["note"] = "snpm_genotype_once_coded: what Genotyper.genotyper calls for a sample parsed from a VCF"
            legs.append(leg)
        # (2) the same sample, query kept: the scoring alone (kernel + reduce + certificate), results to the host
        q = engine.Query(panel, rows0, wei0)
        leg = kernel_leg(ctx, "single_sample_200k_rerun_resident_query", lambda: q.run(1000, False, engine.MODE_EXACT), n_match, row_bytes, args.reps)
        leg["kernel"] = q.last_kernel()
        legs.append(leg)
        # (3) configs[2]: the 399 windows of that sample in one segmented pass (fast + per-(window, accession) certificate)
        leg = kernel_leg(ctx, "cross_399_windows_certified", lambda: q.run_windows(win_off, False, totals=True, fast=True), n_match,
                         row_bytes, args.reps)
        leg["kernel"] = "k_fast_packed_q4<GATHER, SEG>" if packed else "k_fast<GATHER, SEG>"
        leg["windows"] = int(len(win_off) - 1)
        legs.append(leg)
        leg = kernel_leg(ctx, "cross_399_windows_reference_order", lambda: q.run_windows(win_off, False, totals=True, fast=False), n_match,
                         row_bytes, max(2, args.reps // 4), kernel="strict")
        leg["kernel"] = "k_strict4<GATHER>"
        legs.append(leg)
        q.free()
        # (4) B samples per call, inputs already in HBM
        d_rows = torch.as_tensor(cat_rows, device="cuda:0")
        d_wei = torch.as_tensor(cat_wei, device="cuda:0")
        torch.cuda.synchronize()
        dev = (d_rows.data_ptr(), d_wei.data_ptr(), off)
        res = [None]

        def batch():
            res[0] = engine.score_batch(panel, None, device=dev)

        leg = kernel_leg(ctx, "batch_%d_samples_inputs_in_hbm" % B, batch, B * n_match, row_bytes, max(2, args.reps // 2))
        leg["kernel"] = "k_fast_packed_q4<GATHER, SEG>" if packed else "k_fast<GATHER, SEG>"
        leg["samples_per_s"] = B / (leg["wall_ms_per_call"] * 1e-3)
        assert [int(np.nanargmin(res[0]["lik"][b])) for b in range(B)] == [s[2] for s in samples]
        legs.append(leg)
        # (5) the same number of samples genotyped on ONE marker set (each lacks 3 % of it): the per-sample pass reads every sample's
        # rows again (B x 200k gathered rows); the shared-row scan reads each DB row once and scores it against all samples as an int8
        # MFMA contraction (snpm_k_shared.hpp).  Then the shared-row scan forced onto the random-marker batch of (4): almost no
        # (sample, union row) slot holds a call there, the contraction computes zeros -- the automatic policy keeps the per-sample pass.
        chip = chip_batches[0]
        d_rows_c = torch.as_tensor(chip["rows"], device="cuda:0")
        d_wei_c = torch.as_tensor(chip["wei"], device="cuda:0")
        torch.cuda.synchronize()
        dev_c = (d_rows_c.data_ptr(), d_wei_c.data_ptr(), chip["off"])

        def shared_leg(name, device, n_entries, accs, policy, reps):
            engine.batch_configure(ctx, shared_rows=policy)

            def run():
                res[0] = engine.score_batch(panel, None, device=device)

            leg = kernel_leg(ctx, name, run, n_entries, row_bytes, reps)
            st = engine.batch_last_stats(ctx)
            engine.batch_configure(ctx, shared_rows=-1)
            assert [int(np.nanargmin(res[0]["lik"][b])) for b in range(B)] == accs
            leg["samples_per_s"] = B / (leg["wall_ms_per_call"] * 1e-3)
            leg["pairs_reeval"] = res[0]["pairs_reeval"]
            leg["shared_rows"] = st
            if st["taken"]:
                leg["kernel"] = "k_sh_mfma<%s>" % ("packed" if packed else "int8")
                u, rps = st["union_rows"], st["digits"] + 1
                steps = -(-(-(-u // 8)) // 8) * 8                 # steps of 8 union rows, padded to two rounds of 32
                m_rows = st["groups"] * 128
                n_cols = -(-N_ACC // 128) * 128
                macs = float(m_rows) * n_cols * steps * 24           # K = (row, class) over the three informative classes
                # bytes that must come from HBM once: the union's panel rows, the samples' rows + weights, the digit matrix written and read
                hbm = u * (row_bytes + 4.0) + n_entries * 32.0 + 2.0 * steps * 3072.0 * st["groups"]
                leg["algorithmic_bytes_per_call"] = hbm
                leg["hbm_bytes_per_sample"] = hbm / B
                leg["achieved_GBs"] = hbm / (leg["kernel_ms_per_call"] * 1e-3) / 1e9
                leg["frac_of_hbm_peak"] = leg["achieved_GBs"] / HBM_PEAK_GBS
                leg["int8_macs_per_call"] = macs
                peak_macs = 256 * 4 * 1024 * 2.4e9          # 32x32x32 int8 MFMA: 32768 MACs per 32 cycles per SIMD
                leg["mfma_TMACs"] = macs / (leg["kernel_ms_per_call"] * 1e-3) / 1e12
                leg["frac_of_int8_mfma_peak"] = macs / (leg["kernel_ms_per_call"] * 1e-3) / peak_macs
                leg["ceilings"] = {"hbm_ms": hbm / (HBM_PEAK_GBS * 1e9) * 1e3, "int8_mfma_ms": macs / peak_macs * 1e3,
                                   "fp64_valu_ms_if_lut_form": float(n_entries) * N_ACC / (256 * 4 * 16 * 2.4e9) * 1e3,
                                   "note": "fp64 VALU: one v_add_f64 per (sample, row, accession) at 16 lanes per clock and SIMD -- what a shared-row LUT "
                                           "form (or v_mfma_f64 at the same rate and 3x the flops) would be bound by"}
                leg["comparisons_per_s"] = float(n_entries) * N_ACC / (leg["wall_ms_per_call"] * 1e-3)
            else:
                leg["kernel"] = "k_fast_packed_q4<GATHER, SEG>" if packed else "k_fast<GATHER, SEG>"
                leg["hbm_bytes_per_sample"] = leg["algorithmic_bytes_per_call"] / B
            return leg

        reps_b = max(2, args.reps // 2)
        legs.append(shared_leg("batch_%d_per_sample_pass_same_markers" % B, dev_c, int(chip["off"][-1]), chip["accs"], 0, reps_b))
        legs.append(shared_leg("batch_%d_shared_rows_same_markers" % B, dev_c, int(chip["off"][-1]), chip["accs"], 1, reps_b))
        legs.append(shared_leg("batch_%d_shared_rows_random_markers" % B, dev, B * n_match, [s[2] for s in samples], 1, 2))
        del d_rows_c, d_wei_c
        for leg in legs:
            if leg.get("kernel", "").startswith("k_sh_mfma"):
                continue
            if packed:
                leg["bytes_counted"] = "packed bytes (n_acc / 4 per row): the bytes the kernel moves"
                leg["int8_equivalent_GBs"] = leg["achieved_GBs"] * (N_ACC + 32.0) / (row_bytes + 32.0)
                leg["int8_equivalent_note"] = "what an int8 panel would stream for the same comparisons; NOT a bandwidth (may exceed 8 TB/s)"
            else:
                leg["bytes_counted"] = "int8 bytes (n_acc per row)"
            leg["comparisons_per_s"] = leg["algorithmic_bytes_per_call"] / (row_bytes + 32.0) * N_ACC / (leg["kernel_ms_per_call"] * 1e-3) \
                if leg["kernel_ms_per_call"] > 0 else None
        out["formats"][fmt] = {"panel_gb": n_snp * panel.pitch / 1e9, "row_pitch": panel.pitch, "setup_s": time.perf_counter() - t0, "legs": legs}
        del d_rows, d_wei
        panel.free()
        torch.cuda.empty_cache()
    ctx.close()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
