#!/usr/bin/env python3
"""
bench.py -- accession x SNP comparisons/s of the Genotyper hot path on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W      (N>1 via torch.distributed.run)
One JSON line on rank 0.  A "step" is one full pass of the hot path over the resident panel shard:
fast scoring kernel + ordered reduce + exactness check (strict re-evaluation when needed) +
(N>1) RCCL all-gather of per-accession score/ninfo + likelihood / nanmin / LRT on device.

Workload (BASELINE.json configs[3], "Synthetic 10k accessions x 50M SNPs int8, acc-sharded"):
  weak scaling, 62.5 GB of panel per GPU at every N: the job covers 10 000 accessions x
  (6.25 M x N) SNPs, accession-sharded; N=8 is the full 10k x 50M panel (1250 acc x 50M SNPs per
  GPU), N=1 is a 6.25M-SNP slab of it at full 10k-accession row width (the 500 GB panel cannot be
  resident in one 288 GB GPU).  Values P(-1,0,1,2) ~ (0.05,0.60,0.33,0.02) from a counter-based RNG
  generated on the device; sample = planted accession 417 with 2 % error, 80 % PL-derived weights.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable
SEED = 10050
PLANTED = 417
N_ACC_TOTAL = 10000
SNPS_PER_GPU_UNIT = 6_250_000    # x N ranks


def baseline_metric():
    """the metric string of BASELINE.json (falls back to its text when the file is not shipped)"""
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:
        return "accession\u00d7SNP comparisons/sec (whole node); achieved HBM GB/s vs peak"


def make_sample(n_snp, seed, planted, err=0.02, frac_pl=0.8, block=2_000_000):
    """weights [n_snp,3] of a sample planted on accession `planted` (numpy twin of the device panel)."""
    from snpmatch_amd import synth
    rng = np.random.default_rng(seed + 1)
    wei = np.empty((n_snp, 3))
    q0 = planted // 4 * 4
    for r0 in range(0, n_snp, block):
        nr = min(block, n_snp - r0)
        col = synth.panel_values(seed, r0, nr, q0, 4)[:, planted - q0]
        _, w = synth.planted_sample(rng, col, err, frac_pl)
        wei[r0:r0 + nr] = w
    return wei


def cpu_baseline(panel, wei, n_acc, seconds_target=20.0):
    """Reference CPU path (numpy, same expression graph as matchGTsAccs) on a bounded sample of the
    same workload: 1000-row chunks of the resident panel, single thread as the reference runs."""
    from oracle import snpmatch_oracle as orc
    chunk = 1000
    db = panel.download_rows(0, chunk)
    t0 = time.perf_counter()
    orc.match_gts_accs_graph(wei[:chunk], db)
    t1 = time.perf_counter() - t0
    n_chunks = int(max(2, min(100, seconds_target / max(t1, 1e-3))))
    db = panel.download_rows(0, chunk * n_chunks)
    t0 = time.perf_counter()
    s, n = orc.genotyper_scores(wei[:chunk * n_chunks], db, chunk, False, match=orc.match_gts_accs_graph)
    dt = time.perf_counter() - t0
    # for scale: the plain-C restatement of the same arithmetic (oracle/snpmatch_oracle.c, reference summation
    # order) on the host cores of a one-GPU share, one thread per accession block (ctypes releases the GIL; a
    # GPU-initialised process must not fork).  The reference itself is single-threaded numpy: that is `value`.
    from concurrent.futures import ThreadPoolExecutor
    from oracle import c_oracle
    cores = max(1, min(len(os.sched_getaffinity(0)), 16))
    edges = np.linspace(0, n_acc, cores + 1).astype(int)
    blocks = [np.ascontiguousarray(db[:, a:b]) for a, b in zip(edges[:-1], edges[1:]) if b > a]
    wsub = np.ascontiguousarray(wei[:chunk * n_chunks])
    c_oracle.genotyper(blocks[0][:chunk], None, wsub[:chunk], chunk, False)        # load the library
    reps = 3
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as pool:
        for _ in range(reps):
            parts = list(pool.map(lambda blk: c_oracle.genotyper(blk, None, wsub, chunk, False), blocks))
    dt_all = (time.perf_counter() - t0) / reps
    c_counts_ok = bool(np.array_equal(np.concatenate([p_[1] for p_ in parts]), n) and
                       np.array_equal(np.concatenate([p_[0] for p_ in parts]).astype(np.int64), np.asarray(s).astype(np.int64)))
    return {"value": chunk * n_chunks * n_acc / dt, "unit": "comparisons/s", "cores": 1, "kind": "port",
            "sample": "%d x 1000-SNP chunks x %d accessions of the same panel, numpy restatement of "
                      "matchGTsAccs (oracle.match_gts_accs_graph), %.1f s" % (n_chunks, n_acc, dt),
            "c_port_all_cores": {"value": chunk * n_chunks * n_acc / dt_all, "cores": cores, "counts_equal_numpy_port": c_counts_ok,
                                 "how": "C restatement (oracle/snpmatch_oracle.c), same sample, one thread per accession block"}},             (s, n, chunk * n_chunks)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n-acc", type=int, default=N_ACC_TOTAL)
    ap.add_argument("--snps-per-gpu-unit", type=int, default=SNPS_PER_GPU_UNIT)
    ap.add_argument("--mode", default="exact", choices=["exact", "strict", "fast"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--chunk", type=int, default=1000)
    ap.add_argument("--hard-calls", action="store_true",
                    help="sample with hard genotype calls only (all weights 0 or 1, as from a BED file or a VCF without PL)")
    ap.add_argument("--packed", action="store_true",
                    help="2-bit packed panel (4 accessions per byte) instead of the int8 panel BASELINE.json names")
    # rehearsal of the N>1 code path on a one-GPU box (never used by the driver): gloo instead of
    # RCCL and every rank on device 0
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--all-ranks-on-device0", action="store_true")
    ap.add_argument("--force-dist", action="store_true", help="init the process group and all-gather even at N=1")
    args = ap.parse_args()

    # Contract: ONE JSON line on stdout.  RCCL prints a version banner to stdout when its first communicator
    # is created, so file descriptor 1 is pointed at stderr until the result line is ready.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run (WORLD_SIZE=%d)" % (args.gpus, world))
        args.gpus = world

    import torch
    import torch.distributed as dist
    from snpmatch_amd import engine
    from snpmatch_amd.dist import AccessionShards

    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X (no GPU visible); there is no CPU fallback")
    if args.all_ranks_on_device0:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    ctx = engine.Context(local_rank)
    # One dedicated (non-default) stream for everything: the library's kernels are launched on it and
    # torch.distributed orders its collectives against torch's CURRENT stream, so the all-gather
    # cannot start before the scores it gathers are written, with no host synchronisation.
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)

    n_acc = args.n_acc
    n_snp = args.snps_per_gpu_unit * world
    shards = AccessionShards(n_acc, world, rank, dev, force_collective=args.force_dist)
    a0, n_loc, per = shards.a0, shards.n_local, shards.per
    mode = {"exact": engine.MODE_EXACT, "strict": engine.MODE_STRICT, "fast": engine.MODE_FAST}[args.mode]

    t_setup = time.perf_counter()
    panel = engine.Panel(ctx, n_snp, n_loc, packed=args.packed)
    panel.fill_synthetic(SEED, 0, a0)
    wei = make_sample(n_snp, SEED, PLANTED, frac_pl=0.0 if args.hard_calls else 0.8)
    query = engine.Query(panel, None, wei)
    # results land in torch tensors (plumbing for the all-gather); padded to the common shard size
    query.bind_outputs(shards.score_loc.data_ptr(), shards.ninfo_loc.data_ptr())
    lik = torch.zeros(per * world, dtype=torch.float64, device=dev)
    lrt = torch.zeros(per * world, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    t_setup = time.perf_counter() - t_setup

    n_reeval = [0]

    def step():
        _, _, nre = query.run_device(args.chunk, False, mode)
        n_reeval[0] += nre
        src_s, src_n = shards.gather()          # the one collective of the path (no-op at N=1)
        # padded tail entries are (0, 0) -> NaN likelihood, ignored by nanmin
        ctx.likelihood_device(src_s.data_ptr(), src_n.data_ptr(), 1, per * world, lik.data_ptr(), lrt.data_ptr(),
                              truncate=True)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    ctx.profile(True)
    ctx.profile_reset()
    n_reeval[0] = 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    ctx.profile(False)
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    kernel = {"fast": "fast", "exact": "fast", "strict": "strict"}[args.mode]
    launches, k_ms = ctx.profile_read(kernel)
    k_avg_ms = k_ms / max(launches, 1)
    # algorithmic bytes: 1 B per element (0.25 B on a packed panel) + 24 B of fp64 weights per SNP row
    alg_bytes = float(n_snp) * ((n_loc / 4.0 if args.packed else n_loc) + 24.0)
    achieved = alg_bytes / (k_avg_ms * 1e-3) / 1e9 if k_avg_ms > 0 else 0.0
    traffic = None           # PMC-measured HBM bytes per launch, only for the shape they were collected on
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc) and kernel == "fast" and not args.packed:
        try:
            rec = json.load(open(pmc)).get("%d" % world, {})
            if rec.get("n_acc") == n_loc and rec.get("n_snp") == n_snp:
                traffic = rec.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None

    # correctness of what was timed: top hit is the planted accession, counts agree with the CPU path
    top = int(np.nanargmin(lik.cpu().numpy()))
    result_ok = (shards.to_global(top) == PLANTED)

    cpu = None
    parity = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu, (cs, cn, nrows) = cpu_baseline(panel, wei, n_loc)
        q2 = engine.Query(panel, None, wei[:nrows])
        gs, gn = q2.run(args.chunk, False, engine.MODE_EXACT)
        parity = bool(np.array_equal(gn, cn) and np.array_equal(np.array(gs, dtype=int), np.array(cs, dtype=int)))
        q2.free()

    if rank == 0:
        comparisons = float(n_snp) * n_acc * args.steps
        out = {
            "metric": baseline_metric(),
            "value": comparisons / dt,
            "unit": "comparisons/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",          # weighted match sums accumulate in fp64 (panel elements: int8 / 2-bit codes)
            "data": "synthetic",
            "config": {
                "workload": "configs[3]: synthetic 10k accessions x 50M SNPs int8, accession-sharded; "
                            "per GPU %d accessions x %d SNPs (%.1f GB resident), job = %d x %d"
                            % (n_loc, n_snp, n_snp * panel.pitch / 1e9, n_acc, n_snp),
                "n_acc": n_acc, "n_snp": n_snp, "acc_per_gpu": n_loc, "mode": args.mode, "chunk": args.chunk,
                "panel_format": "packed2" if args.packed else "int8",
                "sample": "planted accession %d, 2%% error, 80%% PL weights" % PLANTED,
                "parallelism": "acc-shard x%d + all-gather (%s)" % (world, args.backend if world > 1 else "none"),
            },
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": ("k_fast_packed16" if args.packed else "k_fast") if kernel == "fast" else "k_strict",
                         "launches": launches, "avg_ms": k_avg_ms, "algorithmic_bytes_per_launch": alg_bytes},
            "cpu_baseline": cpu,
            "checks": {"top_hit_is_planted": result_ok, "counts_match_cpu_port": parity,
                       "strict_reevaluations": n_reeval[0]},
            "setup_s": t_setup,
        }
        if args.packed:      # 0.25 B per comparison: the pass is VALU/LDS-issue-bound, the HBM fraction is informative only
            out["roofline"]["note"] = "packed panel: bound by VALU + LDS issue (DESIGN.md), not by HBM"
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
