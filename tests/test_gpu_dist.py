"""
The N>1 code path with the real kernels (-m gpu): two ranks share device 0 and talk over gloo (the driver runs the
real 8-GPU job; this is the same code with RCCL swapped for gloo).  What is under test: accession shards of uneven
size, results bound into the collective's tensors, the library on torch's stream, the all-gather over the padded
vectors, the likelihood over padded (0, 0) tails -- and the product path (Genotyper, --refine, CrossIdentifier)
when Genotype.panel() holds only the rank's shard.  Everything is compared with the unsharded run bit for bit /
with the files of the unmodified reference (core/snpmatch.py:84-88,112: accession columns never interact).
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from snpmatch_amd import engine, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "dist_worker.py")


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_ranks(args, world=2, timeout=600, per_rank_env=None):
    """start `world` copies of tests/dist_worker.py; per_rank_env(rank) -> extra environment (tests/test_gpu_multi.py gives every
    rank its own device and the nccl backend)"""
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
        if per_rank_env is not None:
            env.update(per_rank_env(r))
        procs.append(subprocess.Popen([sys.executable, WORKER] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                                      text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, outs[r][-3000:])


def test_two_ranks_device_flow_equals_unsharded(tmp_path):
    run_ranks(["device", str(tmp_path)])
    n_snp, n_acc, seed, planted = 60_000, 1135, 4242, 417
    ctx = engine.Context(0)
    panel = engine.Panel(ctx, n_snp, n_acc)
    panel.fill_synthetic(seed)
    wei = synth.sample_weights_twin(seed, 0, n_snp, planted)
    q = engine.Query(panel, None, wei)
    want = {"exact": q.run(1000, False, engine.MODE_EXACT), "strict": q.run(1000, False, engine.MODE_STRICT),
            "fast": q.run(1000, False, engine.MODE_FAST)}
    for r in range(2):
        got = np.load(os.path.join(str(tmp_path), "device_rank%d.npz" % r))
        assert got["exact_reeval"][0] >= 2 and got["slab_second_pass"][0] == 1
        for name in ("exact", "strict", "fast"):
            ws, wn = want[name]
            assert np.array_equal(got[name + "_ninfo"], wn), name
            assert np.array_equal(np.array(got[name + "_score"], dtype=np.int64), np.array(ws, dtype=np.int64)), name
            # per-accession scores do not depend on which other accessions share the GPU
            assert np.array_equal(bits(got[name + "_score"]), bits(ws)) or name != "strict"
            lik, lrt = ctx.likelihood(got[name + "_score"], got[name + "_ninfo"], truncate=True)
            assert np.array_equal(bits(got[name + "_lik"]), bits(lik)) and np.array_equal(bits(got[name + "_lrt"]), bits(lrt))
            assert int(np.nanargmin(got[name + "_lik"])) == planted
        # accessions 0, 1 of each shard were re-evaluated in reference order: the strict bits
        for a in (0, 1, 568, 569):
            assert bits(got["exact_score"])[a] == bits(want["strict"][0])[a]
        assert np.array_equal(got["slab_ninfo"], want["exact"][1])
        assert np.array_equal(np.array(got["slab_score"], dtype=np.int64), np.array(want["exact"][0], dtype=np.int64))
    ctx.close()


def test_two_ranks_product_path_writes_the_reference_files(tmp_path, golden_dir):
    from test_gpu_pipeline import cmp_scores_table, cmp_window_table
    run_ranks(["product", str(tmp_path), golden_dir])
    out = str(tmp_path)
    roles = [json.load(open(os.path.join(out, "product_rank%d.json" % r))) for r in range(2)]
    assert roles[0]["writer"] and not roles[1]["writer"]
    gold = json.load(open(os.path.join(golden_dir, "g2_inbred.json")))
    for skip in (0, 1):
        want = gold["inbred_skip%d" % skip]
        cmp_scores_table(open(os.path.join(out, "inbred%d.scores.txt" % skip)).read(), want["scores.txt"])
        assert open(os.path.join(out, "inbred%d.matches.json" % skip)).read() == want["matches.json"]
    gold = json.load(open(os.path.join(golden_dir, "g3_refine.json")))
    assert roles[0]["has_result_fine"] == gold["has_result_fine"] == roles[1]["has_result_fine"]
    cmp_scores_table(open(os.path.join(out, "refine.scores.txt")).read(), gold["scores.txt"])
    cmp_scores_table(open(os.path.join(out, "refine.refined.scores.txt")).read(), gold["refined.scores.txt"])
    assert open(os.path.join(out, "refine.matches.json")).read() == gold["matches.json"]
    gold = json.load(open(os.path.join(golden_dir, "g5_cross.json")))
    for skip in (0, 1):
        want = gold["cross_skip%d" % skip]
        pre = os.path.join(out, "cross%d" % skip)
        cmp_window_table(open(pre + ".windowscore.txt").read(), want[".windowscore.txt"])
        cmp_scores_table(open(pre + ".scores.txt").read(), want[".scores.txt"])
        assert open(pre + ".scores.txt.matches.json").read() == want[".scores.txt.matches.json"]
        assert os.path.exists(pre + ".matches.json") == (".matches.json" in want)


def test_rank_flow_worker_with_one_rank(tmp_path):
    """the worker of tests/test_gpu_multi.py's process-per-GPU test (C ABI group of ranks, id through a file), rehearsed with the
    one rank a one-GPU box can give it: the same code joins, shards, gathers and writes its results"""
    from oracle import c_oracle
    run_ranks(["rank", str(tmp_path)], world=1)
    rng = np.random.default_rng(99)
    db = rng.choice(np.array([-1, 0, 1, 2], dtype=np.int8), size=(30_000, 1135), p=[0.05, 0.60, 0.33, 0.02])
    codes = db[:, 5].copy()
    codes[codes < 0] = 0
    wei = synth.sample_weights(rng, codes, 0.8)
    want_s, want_n = c_oracle.genotyper(db, None, wei, 1000, False)
    got = np.load(os.path.join(str(tmp_path), "rank_flow_rank0.npz"))
    assert tuple(got["bounds"]) == (0, 1135)
    assert np.array_equal(bits(got["strict_score"]), bits(want_s)) and np.array_equal(got["strict_ninfo"], want_n)
    assert np.array_equal(got["exact_score"].astype(np.int64), want_s.astype(np.int64)) and int(np.nanargmin(got["exact_lik"])) == 5
