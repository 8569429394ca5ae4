"""
More than one GPU (-m gpu).  These tests SWITCH THEMSELVES ON when the box they run on has at least two GPUs and skip with that
reason on a one-GPU box (every box this suite has met so far): the first multi-GPU lease checks the correctness of the accession
shards + ONE all-gather design by itself, before anybody draws a scaling curve.

Reference: accession columns never interact (core/snpmatch.py:84-88); the likelihood step needs the minimum over ALL accessions
(core/snpmatch.py:106-117: nanmin over the gathered vector, :112) -- so member r scores columns [a0_r, a1_r) and the only exchange
is the gather of (score, ninfo).  Three ways to form the job, all compared bit for bit with the unsharded run and the files of
the unmodified reference (G2 / G3 / G5):
  * ONE process drives n GPUs through the C ABI (snpm_group_create_local = ncclCommInitAll), uneven shards included;
  * one process per GPU joins the C ABI's group of ranks (snpm_group_create_rank = ncclCommInitRank), id through a file;
  * one process per GPU under torch.distributed with backend nccl (= RCCL): the bench-shaped flow and the product path.
"""
import json
import os

import numpy as np
import pytest

from snpmatch_amd import engine

pytestmark = pytest.mark.gpu


def gpu_count():
    try:
        return engine.device_count()
    except Exception:               # noqa: BLE001  (no GPU at all: the CPU run collects this module too)
        return 0


N_GPUS = gpu_count()
need_two = pytest.mark.skipif(N_GPUS < 2, reason="needs at least 2 GPUs, this box has %d: the test switches itself on at the first "
                                                 "multi-GPU lease" % N_GPUS)


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


@need_two
@pytest.mark.parametrize("n_members,n_acc", [(2, 1135), (2, 257), (0, 1135), (0, 1000)])
def test_one_process_drives_real_devices_over_rccl(n_members, n_acc):
    """snpm_group_create_local over devices 0 .. n-1 (n_members = 0: every GPU of the box): scores, windows, --refine scan, in-silico
    crosses and batches of a GroupPanel against the unsharded run on device 0 -- the checks of the loopback rehearsal, now over
    ncclAllGather between real devices"""
    from test_gpu_group import group_equals_unsharded
    n = n_members or min(N_GPUS, 8)
    group = engine.Group.local(list(range(n)))
    assert "rccl" in group.transport.lower() and (group.world, group.rank0, group.n_local) == (n, 0, n)
    group_equals_unsharded(group, n, n_acc)


@need_two
@pytest.mark.parametrize("world,n_acc", [(2, 1135), (2, 258), (0, 1135)])
def test_rank_processes_join_the_library_group_on_their_own_devices(world, n_acc, tmp_path):
    """one process per GPU, snpm_group_create_rank (ncclCommInitRank) with the id handed over through a file; every rank receives
    the full vectors: reference-order bits, certified counts, likelihoods of the gathered vector (global minimum, :112)"""
    from oracle import c_oracle
    from snpmatch_amd import synth
    from test_gpu_dist import run_ranks
    world = world or min(N_GPUS, 4)
    run_ranks(["rank", str(tmp_path)], world=world, per_rank_env=lambda r: {"SNPM_TEST_DEVICE": str(r), "SNPM_TEST_N_ACC": str(n_acc)})
    rng = np.random.default_rng(99)
    db = rng.choice(np.array([-1, 0, 1, 2], dtype=np.int8), size=(30_000, n_acc), p=[0.05, 0.60, 0.33, 0.02])
    codes = db[:, 5].copy()
    codes[codes < 0] = 0
    wei = synth.sample_weights(rng, codes, 0.8)
    want_s, want_n = c_oracle.genotyper(db, None, wei, 1000, False)
    ctx = engine.Context(0)
    lik, lrt = ctx.likelihood(want_s, want_n, truncate=True)
    covered = []
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), "rank_flow_rank%d.npz" % r))
        covered.append(tuple(got["bounds"]))
        assert np.array_equal(bits(got["strict_score"]), bits(want_s)) and np.array_equal(got["strict_ninfo"], want_n), r
        assert np.array_equal(bits(got["strict_lik"]), bits(lik)) and np.array_equal(bits(got["strict_lrt"]), bits(lrt)), r
        assert np.array_equal(got["exact_ninfo"], want_n) and np.array_equal(got["exact_score"].astype(np.int64), want_s.astype(np.int64)), r
        assert np.array_equal(bits(got["exact_lik"]), bits(lik)) and int(np.nanargmin(got["exact_lik"])) == 5, r
    assert covered[0][0] == 0 and covered[-1][1] == n_acc and all(covered[i][1] == covered[i + 1][0] for i in range(world - 1))
    ctx.close()


@need_two
def test_torch_distributed_ranks_on_their_own_devices_over_rccl(tmp_path, golden_dir):
    """the two flows of tests/test_gpu_dist.py (bench-shaped device flow; Genotyper / --refine / CrossIdentifier writing the
    reference's files) with backend nccl and a GPU per rank instead of gloo on one GPU"""
    from snpmatch_amd import synth
    from test_gpu_dist import run_ranks
    from test_gpu_pipeline import cmp_scores_table, cmp_window_table
    nccl = lambda r: {"SNPM_TEST_BACKEND": "nccl", "SNPM_TEST_DEVICE": str(r), "LOCAL_RANK": str(r), "SNPMATCH_DIST_BACKEND": "nccl"}   # noqa: E731
    run_ranks(["device", str(tmp_path)], world=2, per_rank_env=nccl)
    n_snp, n_acc, seed, planted = 60_000, 1135, 4242, 417
    ctx = engine.Context(0)
    panel = engine.Panel(ctx, n_snp, n_acc)
    panel.fill_synthetic(seed)
    wei = synth.sample_weights_twin(seed, 0, n_snp, planted)
    q = engine.Query(panel, None, wei)
    strict = q.run(1000, False, engine.MODE_STRICT)
    for r in range(2):
        got = np.load(os.path.join(str(tmp_path), "device_rank%d.npz" % r))
        assert np.array_equal(bits(got["strict_score"]), bits(strict[0])) and np.array_equal(got["strict_ninfo"], strict[1])
        for name in ("exact", "fast", "slab"):
            assert np.array_equal(got[name + "_ninfo"], strict[1]), name
            assert np.array_equal(got[name + "_score"].astype(np.int64), strict[0].astype(np.int64)), name
        assert int(np.nanargmin(got["exact_lik"])) == planted
    ctx.close()
    out = str(tmp_path / "product")
    os.makedirs(out)
    run_ranks(["product", out, golden_dir], world=2, per_rank_env=nccl)
    roles = [json.load(open(os.path.join(out, "product_rank%d.json" % r))) for r in range(2)]
    assert roles[0]["writer"] and not roles[1]["writer"]
    gold = json.load(open(os.path.join(golden_dir, "g2_inbred.json")))
    for skip in (0, 1):
        want = gold["inbred_skip%d" % skip]
        cmp_scores_table(open(os.path.join(out, "inbred%d.scores.txt" % skip)).read(), want["scores.txt"])
        assert open(os.path.join(out, "inbred%d.matches.json" % skip)).read() == want["matches.json"]
    gold = json.load(open(os.path.join(golden_dir, "g3_refine.json")))
    cmp_scores_table(open(os.path.join(out, "refine.scores.txt")).read(), gold["scores.txt"])
    cmp_scores_table(open(os.path.join(out, "refine.refined.scores.txt")).read(), gold["refined.scores.txt"])
    gold = json.load(open(os.path.join(golden_dir, "g5_cross.json")))
    for skip in (0, 1):
        want = gold["cross_skip%d" % skip]
        pre = os.path.join(out, "cross%d" % skip)
        cmp_window_table(open(pre + ".windowscore.txt").read(), want[".windowscore.txt"])
        cmp_scores_table(open(pre + ".scores.txt").read(), want[".scores.txt"])


def test_the_multi_gpu_tests_know_why_they_did_not_run():
    """on a one-GPU box the module reports what it is waiting for; on a bigger one this is a tautology"""
    assert N_GPUS >= 1
    if N_GPUS < 2:
        assert "switches itself on" in need_two.kwargs["reason"]
