#!/usr/bin/env python3
"""
bench.py -- accession x SNP comparisons/s of the Genotyper hot path on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W      (N>1 via torch.distributed.run)
One JSON line on rank 0.

Workload = BASELINE.json configs[3]: the synthetic 10 000 accessions x 50M SNPs int8 panel, the WHOLE job at
every N (strong scaling), accession-sharded over the N ranks (10 000 / N accessions x 50M SNPs per GPU) with one
all-gather of the per-accession results and the likelihood step on the full vector.

A "step" is one full pass of the hot path over the job: fast scoring kernel + ordered reduce + certificate
(reference-order re-evaluation of the accessions it flags) + (N>1) RCCL all-gather of per-accession score/ninfo
+ likelihood / nanmin / LRT on the device.

A rank's shard that does not fit in HBM (N=1: 512 GB, N=2: 256 GB) is scored SNP slab after SNP slab with a
carry (snpm_query_run_carry; DESIGN.md "Slabs"): slabs of <= --slab-gb (default 205 GB: 20M SNPs at 10 000
accessions -> 20M + 20M + 10M).  `value` keeps the contract's definition -- inputs resident in HBM when the timed
region starts: for every slab in turn the slab is regenerated on the device (untimed, a stand-in for loading it),
then exactly K scoring steps of that slab are timed between barriers; the timed regions of all slabs (and of the
final certificate / gather / likelihood, and of the second pass over the slabs for flagged accessions) add up to K
steps of the whole job.  The job INCLUDING regeneration is timed once more and reported as `end_to_end`.

Values P(-1,0,1,2) ~ (0.05,0.60,0.33,0.02) from a counter-based RNG generated on the device; sample = planted
accession 417 with 2 % error, 80 % PL-derived weights, generated on the device as well (numpy twins in
snpmatch_amd.synth check both in the tests).
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable
SEED = 10050
PLANTED = 417
N_ACC_TOTAL = 10000
N_SNP_TOTAL = 50_000_000


def baseline_metric():
    """the metric string of BASELINE.json (falls back to its text when the file is not shipped)"""
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:
        return "accession×SNP comparisons/sec (whole node); achieved HBM GB/s vs peak"


def workload_name(n_acc, n_snp):
    if (n_acc, n_snp) == (N_ACC_TOTAL, N_SNP_TOTAL):
        return "configs[3]"
    if (n_acc, n_snp) == (12500, 100_000_000):
        return "configs[4] per-GPU share (12 500 of 100k accessions x 100M SNPs, looped through one resident slab buffer)"
    return "configs[3]-shaped"


def make_sample(n_snp, seed, planted, err=0.02, frac_pl=0.8, block=2_000_000):
    """host weights [n_snp,3] of the bench sample (numpy twin of the device generator; tools and tests)"""
    from snpmatch_amd import synth
    return np.concatenate([synth.sample_weights_twin(seed, r0, min(block, n_snp - r0), planted, err, frac_pl)
                           for r0 in range(0, n_snp, block)])


def cpu_baseline(db, wei, n_acc, seconds_target=20.0):
    """Reference CPU path (numpy, same expression graph as matchGTsAccs) on a bounded sample of the
    same workload: 1000-row chunks of the panel, single thread as the reference runs."""
    from oracle import snpmatch_oracle as orc
    chunk = 1000
    t0 = time.perf_counter()
    orc.match_gts_accs_graph(wei[:chunk], db[:chunk])
    t1 = time.perf_counter() - t0
    n_chunks = int(max(2, min(len(db) // chunk, seconds_target / max(t1, 1e-3))))
    db = db[:chunk * n_chunks]
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


def _numpy_worker(args):
    """one process of the all-cores numpy baseline: an accession block of a panel with the workload's value mix, the numpy
    restatement of matchGTsAccs over 1000-row chunks (the reference's expression graph), again and again for `seconds`"""
    block, n_rows, n_acc_block, seconds, seed = args
    from oracle import snpmatch_oracle as orc
    from snpmatch_amd import synth
    rng = np.random.default_rng(seed + block)
    db = rng.choice(np.array([-1, 0, 1, 2], dtype=np.int8), size=(n_rows, n_acc_block), p=[0.05, 0.60, 0.33, 0.02])
    wei = synth.sample_weights_twin(seed, 0, n_rows, 417)
    orc.genotyper_scores(wei[:1000], db[:1000], 1000, False, match=orc.match_gts_accs_graph)      # imports, first touch
    done = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        orc.genotyper_scores(wei, db, 1000, False, match=orc.match_gts_accs_graph)
        done += n_rows * n_acc_block
    return done, time.perf_counter() - t0


def cpu_baseline_numpy_all_cores(n_acc, seconds=10.0):
    """the numpy path on ALL host cores (multiprocessing over accession blocks), taken BEFORE this process
    touches the GPU -- a GPU-initialised process must not fork.  The reference itself has no threading: this is what a user
    gets by running one reference process per accession block."""
    import multiprocessing as mp
    cores = max(1, min(len(os.sched_getaffinity(0)), 16))
    per = max(1, n_acc // cores)
    with mp.get_context("fork").Pool(cores) as pool:
        res = pool.map(_numpy_worker, [(b, 20_000, per, seconds, 10050) for b in range(cores)])
    total = sum(r[0] for r in res)
    wall = max(r[1] for r in res)
    return {"value": total / wall, "unit": "comparisons/s", "cores": cores,
            "how": "numpy restatement of matchGTsAccs (oracle.match_gts_accs_graph, the reference's expression graph), %d processes, each "
                   "an accession block of %d x 20000 SNPs with the workload's value mix, 1000-row chunks, %.1f s" % (cores, per, wall)}


def child_legs(out, args, world):
    """The legs reported beside the headline, each measured by a CHILD process after this one has released its memory (a failure
    there only drops the extra field): the same job on the 2-bit packed panel, the real-panel configurations
    (tools/bench_real_panel.py), the DB staging path (tools/bench_staging.py)."""
    # Beside the headline (the int8 panel BASELINE.json names): the same job on the 2-bit packed panel, which fits one GPU
    # whole (125 GB).  A child process after this one has released its memory; a failure there only drops the extra field.
    whole_job = (args.n_acc == N_ACC_TOTAL and args.n_snp == N_SNP_TOTAL and args.mode == "exact" and not args.hard_calls)
    # (not under a profiler: its preloaded tool would follow the child)
    profiled = any("rocprof" in os.environ.get(k, "").lower() for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB"))
    if world == 1 and not args.packed and not args.no_alternatives and whole_job and not profiled:
        try:
            import subprocess
            child = subprocess.run([sys.executable, os.path.abspath(__file__), "--packed", "--steps", "8", "--warmup", "2",
                                    "--no-cpu-baseline", "--no-alternatives"], capture_output=True, text=True, timeout=300)
            alt = json.loads(child.stdout.strip().splitlines()[-1])
            out["alternatives"] = {"packed2_panel_resident": {
                "value": alt["value"], "unit": alt["unit"], "ms_per_step": alt["ms_per_step"], "steps": alt["steps"],
                "kernel": alt["roofline"]["kernel"], "kernel_avg_ms": alt["roofline"]["avg_ms"],
                "panel_gb": N_SNP_TOTAL * ((N_ACC_TOTAL // 4 + 255) // 256 * 256) / 1e9, "checks": alt["checks"],
                "note": "same job, same sample, 2 bits per call: results identical (tests), not the format the metric names"}}
        except Exception as e:          # noqa: BLE001
            out["alternatives"] = {"packed2_panel_resident": {"error": str(e)[:200]}}
    # The real-panel configurations (configs[1]: one 200k-SNP sample against 1135 x 11M resident -- gathered rows; a batch of 64;
    # configs[2]: the 399 windows of `cross`), int8 and packed, as a child process: tools/bench_real_panel.py.  Informative
    # legs beside the headline, each with its kernel, its HIP-event duration, algorithmic bytes and fraction of the HBM peak.
    if world == 1 and not args.no_real_panel and whole_job and not profiled and not args.packed:
        try:
            import subprocess
            child = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_real_panel.py"), "--reps", "10"],
                                   capture_output=True, text=True, timeout=300)
            rp = json.loads(child.stdout.strip().splitlines()[-1])
            keep = ("leg", "kernel", "wall_ms_per_call", "kernel_ms_per_call", "algorithmic_bytes_per_call", "achieved_GBs",
                    "frac_of_hbm_peak", "bytes_counted", "int8_equivalent_GBs", "samples_per_s", "windows", "shared_rows", "hbm_bytes_per_sample",
                    "frac_of_int8_mfma_peak", "mfma_TMACs", "ceilings", "pairs_reeval", "other_kernels_ms_per_call")
            out["real_panel"] = {"workload": rp["workload"], "traffic": "profiles/r04_pmc_split_real_*.json (request-size split, separate --pmc passes; per-sample legs)",
                                 "formats": {f: {"row_pitch": v["row_pitch"], "panel_gb": v["panel_gb"],
                                                 "legs": [{k: leg[k] for k in keep if k in leg} for leg in v["legs"]]}
                                             for f, v in rp["formats"].items()}}
        except Exception as e:          # noqa: BLE001
            out["real_panel"] = {"error": str(e)[:200]}
    # The DB staging path north_star names (pinned-host slabs + hipMemcpyAsync on a side stream, in place of the reference's
    # h5py read, core/snpmatch.py:222): a 20 GB int8 host panel into int8 / packed panels from memory, from a flat file and
    # from the reference's lzf HDF5 layout, against the hipMemcpyAsync ceiling of the same run, and an upload while a
    # resident query is being scored.  A child process: tools/bench_staging.py.
    if world == 1 and not args.no_staging and whole_job and not profiled and not args.packed:
        try:
            import subprocess
            child = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_staging.py")], capture_output=True, text=True,
                                   timeout=420)
            out["staging"] = json.loads(child.stdout.strip().splitlines()[-1])
        except Exception as e:          # noqa: BLE001
            out["staging"] = {"error": str(e)[:200]}


def join_library_group(args, ctx, dev, rank, world, n_acc, shards, use_dist, side_group, real_stdout):
    """The collective of the path behind the C ABI (snpm_group_*): rank 0 makes the RCCL id, torch.distributed (already up for
    the barriers of the timing contract) carries its 128 bytes to the other ranks, every rank joins with its context.  Joined
    BEFORE the panel is allocated, so that a process whose join hangs holds next to no HBM.
      every rank joined           -> collective.transport = "c-abi-rccl"
      a rank failed cleanly       -> every rank gathers through torch.distributed in this process: "torch-nccl-fallback" + reason
      a rank is STUCK in the join -> nothing is measured in this process (a thread inside ncclCommInitRank holds the context):
                                     every rank starts a fresh child with --collective torch, rank 0's child prints the line
                                     (transport "torch-nccl-fallback", the reason, "degraded": true) and the parents leave with
                                     exit code 3, so that the run is recorded as degraded, not as a normal one.
    The ranks agree on the outcome over `side_group` (gloo, with a timeout): the NCCL process group may be exactly what hangs, and a
    rank that hears nothing within the timeout treats the join as stuck too.
    Returns (group or None, collective record, seconds spent joining)."""
    import torch
    import torch.distributed as dist
    from snpmatch_amd import engine
    group = None
    coll = {"transport": "none", "reason": None}
    join_s = 0.0
    if not use_dist:
        return group, coll, join_s
    coll["transport"] = "torch-%s" % args.backend
    if args.degraded_reason:
        coll = {"transport": "torch-nccl-fallback", "reason": args.degraded_reason, "degraded": True}
    if not (args.collective == "c-abi" and args.backend == "nccl"):
        return group, coll, join_s
    status, why = 2, ""                 # 2 joined, 1 failed cleanly, 0 stuck
    t_join = time.perf_counter()
    try:
        box = [engine.Group.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0, group=side_group)
        made = {}

        def join_group():
            try:
                if os.environ.get("SNPM_BENCH_SIMULATE_STUCK_JOIN") == str(rank):      # rehearsal of the stuck branch
                    time.sleep(3600)
                made["group"] = engine.Group.from_rank(ctx, box[0], world, rank)
            except Exception as exc:          # noqa: BLE001
                made["error"] = exc

        th = threading.Thread(target=join_group, daemon=True)
        th.start()
        th.join(args.group_timeout)
        if th.is_alive():
            status, why = 0, "rank %d: snpm_group_create_rank (ncclCommInitRank) did not return within %.0f s" % (rank, args.group_timeout)
        elif "error" in made:
            status, why = 1, "rank %d: snpm_group_create_rank failed: %s" % (rank, str(made["error"])[:200])
        else:
            group = made["group"]
            assert group.shard(n_acc, rank) == (shards.a0, shards.a1), "the library and bench.py disagree about the shards"
    except Exception as e:          # noqa: BLE001
        status, why = 1, "rank %d: %s" % (rank, str(e)[:200])
    join_s = time.perf_counter() - t_join
    if why:
        sys.stderr.write(why + "\n")
    # the verdict of all ranks, over gloo (CPU tensors): a hang of the NCCL side cannot block it, and it times out
    try:
        t = torch.tensor([status], dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=side_group)
        worst = int(t.item())
        reasons = [None] * world
        dist.all_gather_object(reasons, why, group=side_group)
        reason = "; ".join(r for r in reasons if r) or None
    except Exception as e:          # noqa: BLE001
        worst, reason = 0, "rank %d: no agreement with the other ranks within the side group's timeout (%s)" % (rank, str(e)[:120])
    if worst == 2:
        coll = {"transport": "c-abi-rccl", "reason": None,
                "how": "snpm_group_gather_scores: one ncclAllGather inside libsnpmatch_hip.so (%s)" % group.transport}
    elif worst == 1:
        if group is not None:
            group.free()
        group = None
        coll = {"transport": "torch-nccl-fallback", "reason": reason}
    else:
        # stuck: leave this process alone.  torch's process group is abandoned, not torn down (a destroy could hang on the same
        # communicator); the children form their own on the next port
        try:
            dist.barrier(group=side_group)
        except Exception:          # noqa: BLE001
            pass
        import subprocess
        env = dict(os.environ, MASTER_PORT=str(int(os.environ.get("MASTER_PORT", "29533")) + 1))
        env.pop("SNPM_BENCH_SIMULATE_STUCK_JOIN", None)
        cmd = [sys.executable, os.path.abspath(__file__)] + [a for a in sys.argv[1:]] + \
              ["--collective", "torch", "--degraded-reason", "library RCCL group join hung (%s); measured in a fresh process over torch.distributed" % reason]
        child = subprocess.run(cmd, env=env, stdout=real_stdout if rank == 0 else subprocess.DEVNULL)
        sys.stderr.write("rank %d: degraded run, child exit code %d\n" % (rank, child.returncode))
        os._exit(3 if child.returncode == 0 else 4)
    return group, coll, join_s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n-acc", type=int, default=N_ACC_TOTAL)
    ap.add_argument("--n-snp", type=int, default=N_SNP_TOTAL, help="SNPs of the whole job")
    ap.add_argument("--slab-gb", type=float, default=205.0, help="largest resident SNP slab per GPU (10^9 bytes)")
    ap.add_argument("--slabs", type=int, default=0, help="score the shard in this many equal slabs (default: as few as fit --slab-gb)")
    ap.add_argument("--mode", default="exact", choices=["exact", "strict", "fast"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the extra untimed-by-K pass that includes slab regeneration")
    ap.add_argument("--no-alternatives", action="store_true", help="skip the extra packed-panel run reported beside the headline")
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
    ap.add_argument("--group-timeout", type=float, default=120.0,
                    help="seconds the ranks may take to join the library's RCCL group before all of them fall back to torch.distributed")
    ap.add_argument("--degraded-reason", default="", help=argparse.SUPPRESS)      # set by a parent whose group join hung (see join_library_group)
    ap.add_argument("--no-real-panel", action="store_true", help="skip the configs[1] / configs[2] legs (tools/bench_real_panel.py) after the headline")
    ap.add_argument("--no-staging", action="store_true", help="skip the DB staging legs (tools/bench_staging.py) after the headline")
    ap.add_argument("--collective", default="c-abi", choices=["c-abi", "torch"],
                    help="who runs the all-gather of the per-accession results at N>1: the library itself (snpm_group_*: RCCL "
                         "communicator inside libsnpmatch_hip.so, one packed all-gather) or torch.distributed")
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

    numpy_all_cores = None
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        try:
            numpy_all_cores = cpu_baseline_numpy_all_cores(args.n_acc)       # before anything touches the GPU (fork)
        except Exception as e:          # noqa: BLE001
            numpy_all_cores = {"error": str(e)[:200]}

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
    side_group = None
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        # a gloo side group with a timeout for what must not depend on the NCCL communicator (the verdict about the library's
        # group join: join_library_group)
        import datetime
        side_group = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=max(60.0, args.group_timeout + 30.0)))

    ctx = engine.Context(local_rank)
    # One dedicated (non-default) stream for everything: the library's kernels are launched on it and
    # torch.distributed orders its collectives against torch's CURRENT stream, so the all-gather
    # cannot start before the scores it gathers are written, with no host synchronisation.
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)

    n_acc, n_snp, chunk = args.n_acc, args.n_snp, args.chunk
    planted = PLANTED if n_acc > PLANTED else n_acc - 1      # narrow test panels have no accession 417
    shards = AccessionShards(n_acc, world, rank, dev, force_collective=args.force_dist)
    a0, n_loc, per = shards.a0, shards.n_local, shards.per
    mode = {"exact": engine.MODE_EXACT, "strict": engine.MODE_STRICT, "fast": engine.MODE_FAST}[args.mode]

    group, coll, join_s = join_library_group(args, ctx, dev, rank, world, n_acc, shards, use_dist, side_group, real_stdout)

    # ---- slabs of this rank's shard (the same on every rank: sized for the widest shard)
    pitch_max = ctx.row_pitch(per, args.packed)          # the library's rule (128-B rows for narrow int8 panels, + 256 B at multiples of 8 KiB)
    if args.slabs > 0:
        rows_per_slab = -(-(-(-n_snp // args.slabs)) // chunk) * chunk
    else:
        rows_per_slab = max(chunk, int(args.slab_gb * 1e9 / pitch_max) // chunk * chunk)
    rows_per_slab = min(rows_per_slab, n_snp)
    slabs = [min(rows_per_slab, n_snp - r0) for r0 in range(0, n_snp, rows_per_slab)]
    starts = [sum(slabs[:k]) for k in range(len(slabs))]
    S = len(slabs)

    def chunks_after(k):
        return sum(-(-s // chunk) for s in slabs[k + 1:])

    t_setup = time.perf_counter()
    panel = engine.Panel(ctx, rows_per_slab, n_loc, packed=args.packed)
    # the sample, generated on the device (every rank makes the same one; no host pass over 50M rows)
    wei_dev = torch.empty((n_snp, 3), dtype=torch.float64, device=dev)
    ctx.sample_synthetic(SEED, 0, n_snp, planted, wei_dev.data_ptr(), err=0.02, frac_pl=0.0 if args.hard_calls else 0.8)
    queries = [engine.Query.from_device(panel, None, wei_dev[starts[k]:].data_ptr(), slabs[k]) for k in range(S)]
    lik = torch.zeros(per * world, dtype=torch.float64, device=dev)
    lrt = torch.zeros(per * world, dtype=torch.float64, device=dev)

    def load_slab(k):
        panel.fill_synthetic(SEED, snp0=starts[k], acc0=a0, row0=0, nrows=slabs[k])

    load_slab(0)
    torch.cuda.synchronize()
    t_setup = time.perf_counter() - t_setup

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def gather_and_likelihood():
        if group is not None:
            # pack + ONE ncclAllGather + unpack on the library's stream; the full-length vectors stay on the device
            group.gather([shards.score_loc.data_ptr()], [shards.ninfo_loc.data_ptr()], n_acc, host=False)
            src_s, src_n = group.gathered_ptrs(0)
            ctx.likelihood_device(src_s, src_n, 1, n_acc, lik.data_ptr(), lrt.data_ptr(), truncate=True)
            return
        src_s, src_n = shards.gather()          # the one collective of the path (no-op at N=1)
        # padded tail entries are (0, 0) -> NaN likelihood, ignored by nanmin
        ctx.likelihood_device(src_s.data_ptr(), src_n.data_ptr(), 1, per * world, lik.data_ptr(), lrt.data_ptr(),
                              truncate=True)

    def timed(fn, warm):
        """W untimed + K timed calls of fn between barriers -> seconds of the K calls on this rank"""
        for _ in range(warm):
            fn()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            fn()
        barrier()
        return time.perf_counter() - t0

    prof = {}           # rows of the slab -> [passes over the slab, launches, ms] of the scoring kernel

    def collect_profile(rows):
        kern = "strict" if args.mode == "strict" else "fast"
        n, ms = ctx.profile_read(kern)
        e = prof.setdefault(rows, [0, 0, 0.0])
        e[0] += args.steps       # the reference-order kernel takes several launches per pass (bounded workspace)
        e[1] += n
        e[2] += ms

    n_reeval = 0
    second_pass = False
    dt = 0.0
    ctx.profile(True)
    if S == 1:
        # the whole shard is resident: a step is the complete hot path, results land in the torch tensors
        queries[0].bind_outputs(shards.score_loc.data_ptr(), shards.ninfo_loc.data_ptr())

        def step():
            queries[0].run_device(chunk, False, mode)
            gather_and_likelihood()

        for _ in range(args.warmup):
            step()
        barrier()
        ctx.profile_reset()
        dt = timed(step, 0)
        collect_profile(slabs[0])
        n_reeval = queries[0].last_reeval() * args.steps if args.mode == "exact" else 0
    else:
        carry = engine.Carry(ctx, n_loc)
        carry.bind_outputs(shards.score_loc.data_ptr(), shards.ninfo_loc.data_ptr())
        scratch = engine.Carry(ctx, n_loc)      # takes the K timed repetitions of a slab (same work, totals unused)
        for k in range(S):
            if k:
                load_slab(k)
            queries[k].run_carry(carry, chunk, False, mode, chunks_after(k))       # the one that counts
            scratch.reset()
            fn = lambda: queries[k].run_carry(scratch, chunk, False, mode, 0 if mode == engine.MODE_STRICT else chunks_after(k))  # noqa: E731
            for _ in range(args.warmup):
                fn()
            scratch.reset()
            barrier()
            ctx.profile_reset()
            dt += timed(fn, 0)
            collect_profile(slabs[k])
        # certificate over the totals, (N>1) all-gather, likelihood: the tail of every step
        flagged = []

        def tail():
            nonlocal flagged
            _, _, flagged = carry.finish(want_results=False)
            gather_and_likelihood()

        dt += timed(tail, args.warmup)
        n_fl = carry.n_flagged
        if use_dist:        # a second pass is a collective decision (every rank regenerates the slabs again)
            t = torch.tensor([n_fl], dtype=torch.int64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            any_flagged = int(t.item()) > 0
        else:
            any_flagged = n_fl > 0
        n_reeval = n_fl * args.steps
        if any_flagged and mode == engine.MODE_EXACT:
            # second pass over the slabs: the flagged accessions in reference order, chain carried across slabs
            second_pass = True
            assert n_fl <= 64, "more than 64 accessions flagged: run the job in strict mode"
            cols = engine.Carry(ctx, n_loc)
            cols_scratch = engine.Carry(ctx, n_loc)
            mine = flagged if n_fl > 0 else np.array([0], dtype=np.int32)      # ranks with nothing flagged keep step
            cols.set_columns(mine)
            for k in range(S):
                load_slab(k)
                queries[k].run_carry(cols, chunk, False, engine.MODE_STRICT, chunks_after(k))
                cols_scratch.reset()
                cols_scratch.set_columns(mine)
                fn = lambda: queries[k].run_carry(cols_scratch, chunk, False, engine.MODE_STRICT, 0)  # noqa: E731
                dt += timed(fn, min(args.warmup, 1))
            if n_fl > 0:
                carry.patch_from(cols)
            dt += timed(gather_and_likelihood, 0)
        else:
            gather_and_likelihood()
    ctx.profile(False)
    barrier()
    per_rank = None
    ranks_agree = None
    if use_dist:
        mine = {"rank": rank, "ms_per_step": dt / args.steps * 1e3, "group_join_s": join_s, "acc": [int(a0), int(a0 + n_loc)]}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # every rank holds the full-length gathered vectors and the likelihoods computed from them: their bit patterns must
        # be the same on all ranks (an all-gather that dropped or misplaced a shard shows up here, not only in the top hit)
        torch.cuda.synchronize()
        ck = (lik[:n_acc].view(torch.int64).sum() + lrt[:n_acc].view(torch.int64).sum() * 31).reshape(1).clone()
        lo, hi = ck.clone(), ck.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        ranks_agree = bool(int(lo.item()) == int(hi.item()))

    # ---- the job once more INCLUDING the regeneration of every slab (what a panel larger than HBM really costs)
    end_to_end = None
    if S > 1 and not args.no_end_to_end:
        c2 = engine.Carry(ctx, n_loc)
        barrier()
        t0 = time.perf_counter()
        for k in range(S):
            load_slab(k)
            queries[k].run_carry(c2, chunk, False, mode, chunks_after(k))
        _, _, fl = c2.finish(want_results=False)
        if c2.n_flagged > 0 and mode == engine.MODE_EXACT and c2.n_flagged <= 64:
            cc = engine.Carry(ctx, n_loc)
            cc.set_columns(fl)
            for k in range(S):
                load_slab(k)
                queries[k].run_carry(cc, chunk, False, engine.MODE_STRICT, chunks_after(k))
            c2.patch_from(cc)
        barrier()
        e2e = time.perf_counter() - t0
        synth_n, synth_ms = 0, 0.0
        end_to_end = {"job_ms_incl_slab_regeneration": e2e * 1e3, "value_incl_slab_regeneration": float(n_snp) * n_acc / e2e,
                      "note": "slabs are regenerated on the device (k_synth) between passes, a stand-in for loading them; "
                              "a second regeneration pass serves the flagged accessions"}

    kernel = "strict" if args.mode == "strict" else "fast"
    # roofline of the dominant kernel on the dominant slab shape
    rows_dom = max(prof, key=lambda r: r)
    passes, launches, k_ms = prof[rows_dom]
    k_avg_ms = k_ms / max(passes, 1)              # kernel time of ONE pass over the slab
    # algorithmic bytes: 1 B per element (0.25 B on a packed panel) + 24 B of fp64 weights per SNP row
    row_bytes = (n_loc / 4.0 if args.packed else n_loc) + 24.0
    alg_bytes = float(rows_dom) * row_bytes
    achieved = alg_bytes / (k_avg_ms * 1e-3) / 1e9 if k_avg_ms > 0 else 0.0
    all_ms = sum(v[2] for v in prof.values())
    all_bytes = sum(v[0] * float(r) * row_bytes for r, v in prof.items())
    traffic, traffic_source = None, None          # PMC-measured HBM bytes per launch (separate --pmc passes)
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc) and kernel == "fast" and not args.packed:
        try:
            from snpmatch_amd import _lib
            for rec in json.load(open(pmc)).values():
                if isinstance(rec, dict) and rec.get("n_acc") == n_loc and rec.get("n_snp") == rows_dom:
                    if rec.get("build_id") == _lib.build_id():
                        traffic = rec.get("hbm_bytes_per_launch")
                        traffic_source = "profiles/pmc_traffic.json (separate rocprofv3 --pmc passes on this shape and this build, not this run)"
                    else:       # a figure collected on other kernels says nothing about these
                        traffic_source = "profiles/pmc_traffic.json holds this shape for build %s, the library is build %s: not reported" \
                                         % (rec.get("build_id"), _lib.build_id())
        except Exception:
            traffic = None

    # correctness of what was timed: top hit is the planted accession, counts agree with the CPU path
    top = int(np.nanargmin(lik.cpu().numpy()[:n_acc] if group is not None else lik.cpu().numpy()))
    result_ok = ((top if group is not None else shards.to_global(top)) == planted)

    cpu = None
    parity = None
    if rank == 0 and not args.no_cpu_baseline:
        # rank 0 at every N: the reference's CPU path on a bounded sample of ITS shard, and the counts of the GPU
        # path on the same rows against it (the other ranks wait at the barrier below)
        load_slab(0)
        nrows = min(100_000, slabs[0])
        db = panel.download_rows(0, nrows)
        wei_host = wei_dev[:nrows].cpu().numpy()
        cpu, (cs, cn, nrows) = cpu_baseline(db, wei_host, n_loc)
        if numpy_all_cores is not None:
            cpu["numpy_all_cores"] = numpy_all_cores
        q2 = engine.Query(panel, None, wei_host[:nrows])
        gs, gn = q2.run(chunk, False, engine.MODE_EXACT)
        parity = bool(np.array_equal(gn, cn) and np.array_equal(np.array(gs, dtype=int), np.array(cs, dtype=int)))
        q2.free()

    # what a kernel that only READS this slab reaches on this GPU (k_calib_read: 4 B per lane, non-temporal, no
    # arithmetic): the practical ceiling next to the 8 TB/s of the data sheet
    ceiling = None
    if rank == 0:
        ts = []
        for i in range(5):
            t0 = time.perf_counter()
            nbytes = panel.stream_read()          # synchronises
            if i >= 2:
                ts.append(time.perf_counter() - t0)
        ceiling = nbytes / min(ts) / 1e9

    if rank == 0:
        comparisons = float(n_snp) * n_acc * args.steps
        kname = queries[0].last_kernel() or ("k_strict4" if kernel == "strict" else "k_fast")
        out = {
            "metric": baseline_metric(),
            "value": comparisons / dt,
            "unit": "comparisons/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",          # weighted match sums accumulate in fp64 (panel elements: int8 / 2-bit codes)
            "data": "synthetic",
            "config": {
                "workload": "%s: synthetic %d accessions x %d SNPs %s, the whole job, accession-sharded over %d GPU(s); "
                            "per GPU %d accessions x %d SNPs = %.1f GB, scored in %d resident SNP slab(s) of %s rows "
                            "(largest %.1f GB)"
                            % (workload_name(n_acc, n_snp), n_acc, n_snp, "2-bit packed" if args.packed else "int8", world, n_loc, n_snp,
                               n_snp * panel.pitch / 1e9, S, "+".join(str(s) for s in slabs), rows_per_slab * panel.pitch / 1e9),
                "n_acc": n_acc, "n_snp": n_snp, "acc_per_gpu": n_loc, "mode": args.mode, "chunk": chunk,
                "panel_format": "packed2" if args.packed else "int8", "slabs": slabs,
                "sample": "planted accession %d, 2%% error, %s (generated on the device)"
                          % (planted, "hard 0/1 calls" if args.hard_calls else "80% PL weights"),
                "parallelism": "acc-shard x%d + all-gather (%s)" % (world, coll["transport"]),
                "collective": coll,
                "timing": "per slab: regenerate (untimed), K timed scoring steps between barriers; plus K timed "
                          "certificate/gather/likelihood tails%s; ms_per_step = their sum / K"
                          % (" and K timed second-pass steps per slab for the flagged accessions" if second_pass else "")
                          if S > 1 else "K complete steps between barriers",
            },
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": kname, "launches": launches, "passes": passes, "avg_ms": k_avg_ms,
                         "algorithmic_bytes_per_launch": alg_bytes * passes / max(launches, 1), "algorithmic_bytes_per_pass": alg_bytes,
                         "shape": "%d accessions x %d SNPs" % (n_loc, rows_dom),
                         "kernel_by_shape": [{"shape": "%d x %d" % (n_loc, r), "passes": v[0], "launches": v[1], "avg_ms_per_pass": v[2] / max(v[0], 1),
                                              "frac": float(r) * row_bytes / (v[2] / max(v[0], 1) * 1e-3) / 1e9 / HBM_PEAK_GBS if v[2] > 0 else None}
                                             for r, v in sorted(prof.items(), reverse=True)],
                         "all_slabs_frac": (all_bytes / (all_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if all_ms > 0 else None,
                         "end_to_end_frac": float(n_snp) * row_bytes / (dt / args.steps) / 1e9 / HBM_PEAK_GBS,
                         "read_only_kernel": {"kernel": "k_calib_read", "achieved": ceiling,
                                              "frac": ceiling / HBM_PEAK_GBS if ceiling else None,
                                              "this_kernel_over_read_only": achieved / ceiling if ceiling else None}},
            "cpu_baseline": cpu,
            "end_to_end": end_to_end,
            "checks": {"top_hit_is_planted": result_ok, "counts_match_cpu_port": parity,
                       "strict_reevaluations": int(n_reeval), "second_pass_over_slabs": second_pass, "ranks_agree": ranks_agree},
            "per_rank": per_rank,
            "setup_s": t_setup,
            "library_build_id": __import__("snpmatch_amd._lib", fromlist=["build_id"]).build_id(),
        }
        if args.packed:      # 0.25 B per comparison: the pass is VALU/LDS-issue-bound, the HBM fraction is informative only
            out["roofline"]["note"] = "packed panel: bound by VALU + LDS issue (DESIGN.md), not by HBM"
    if group is not None:
        group.free()
    if use_dist:
        dist.barrier()                  # rank 0 may have spent ~30 s in the CPU baseline
        dist.destroy_process_group()
    ctx.close()
    if rank == 0:
        child_legs(out, args, world)
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if rank == 0 and (not out["checks"]["top_hit_is_planted"] or out["checks"]["ranks_agree"] is False):
        sys.exit(5)                     # a wrong result is a failed run, whatever it measured
    if args.degraded_reason:
        sys.exit(0)                     # the parent turns this into its own exit code 3


if __name__ == "__main__":
    main()
