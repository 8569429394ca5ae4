"""
The N>1 path on CPU: world_size 2 and 3 (uneven shards) over gloo.  Each rank scores its accession
shard (the oracle stands in for the GPU kernel here: what is under test is the sharding, padding and
the all-gather), and every rank must end up with the full vectors of the unsharded computation.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import c_oracle
from oracle import snpmatch_oracle as orc
from snpmatch_amd import dist as sdist
from snpmatch_amd import synth


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def worker(rank, world, port, n_snp, n_acc, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(7)
        planted = synth.panel_values(99, 0, n_snp, 416, 4)[:, 1]
        codes, wei = synth.planted_sample(rng, planted, err=0.02)

        def local(a0, a1):
            assert a0 % 4 == 0
            db = synth.panel_values(99, 0, n_snp, a0, a1 - a0)          # this rank's columns only
            return c_oracle.genotyper(db, None, wei, 1000, False)

        score, ninfo = sdist.sharded_genotyper_scores(local, n_acc, world, rank)
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), score=score, ninfo=ninfo)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_acc", [(2, 1135), (3, 1000), (2, 6)])
def test_sharded_scores_equal_unsharded(world, n_acc, tmp_path):
    n_snp = 3000
    port = free_port()
    mp.spawn(worker, args=(world, port, n_snp, n_acc, str(tmp_path)), nprocs=world, join=True)
    rng = np.random.default_rng(7)
    planted = synth.panel_values(99, 0, n_snp, 416, 4)[:, 1]
    codes, wei = synth.planted_sample(rng, planted, err=0.02)
    want_s, want_n = c_oracle.genotyper(synth.panel_values(99, 0, n_snp, 0, n_acc), None, wei, 1000, False)
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        assert np.array_equal(got["score"].view(np.uint64), want_s.view(np.uint64))
        assert np.array_equal(got["ninfo"], want_n)
    if n_acc > 417:
        lik, lrt = orc.calculate_likelihoods(np.array(want_s, dtype=int), want_n)
        assert int(np.nanargmin(lik)) == 417


def window_worker(rank, world, port, n_snp, n_acc, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(11)
        wei = synth.sample_weights(rng, rng.choice(np.array([0, 1, 2], dtype=np.int8), size=n_snp), 0.8)
        off = np.array([0, 0, 400, 401, 1500, n_snp], dtype=np.int64)       # an empty window included

        def local(a0, a1):
            db = synth.panel_values(5, 0, n_snp, a0, a1 - a0)
            return orc.window_scores(wei, db, off)[:2]

        score, ninfo = sdist.sharded_window_scores(local, n_acc, len(off) - 1, world, rank)
        np.savez(os.path.join(out_dir, "win_rank%d.npz" % rank), score=score, ninfo=ninfo)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_acc", [(2, 70), (3, 50)])
def test_sharded_window_scores_equal_unsharded(world, n_acc, tmp_path):
    n_snp = 2000
    port = free_port()
    mp.spawn(window_worker, args=(world, port, n_snp, n_acc, str(tmp_path)), nprocs=world, join=True)
    rng = np.random.default_rng(11)
    wei = synth.sample_weights(rng, rng.choice(np.array([0, 1, 2], dtype=np.int8), size=n_snp), 0.8)
    off = np.array([0, 0, 400, 401, 1500, n_snp], dtype=np.int64)
    want_s, want_n = orc.window_scores(wei, synth.panel_values(5, 0, n_snp, 0, n_acc), off)[:2]
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), "win_rank%d.npz" % r))
        assert np.array_equal(got["score"].view(np.uint64), np.asarray(want_s, dtype=np.float64).view(np.uint64))
        assert np.array_equal(got["ninfo"], want_n)


def test_shard_bounds_and_padding():
    b, per = sdist.shard_bounds(10000, 8)
    assert per == 1252 and b[0] == (0, 1252) and b[7] == (8764, 10000) and all(a % 4 == 0 for a, _ in b)
    b, per = sdist.shard_bounds(10000, 1)
    assert b == [(0, 10000)] and per == 10000
    b, per = sdist.shard_bounds(6, 4)          # more ranks than quads: trailing shards are empty
    assert per == 4 and b == [(0, 4), (4, 6), (6, 6), (6, 6)]
    sh = sdist.AccessionShards(10, world=1)
    assert sh.padded_index().tolist() == list(range(10)) and sh.to_global(7) == 7
    sh = sdist.AccessionShards(10, world=3, rank=2)
    assert sh.per == 4 and sh.bounds == [(0, 4), (4, 8), (8, 10)] and sh.n_local == 2
    assert sh.padded_index().tolist() == [0, 1, 2, 3, 4, 5, 6, 7, 8, 9]
    assert sh.to_global(9) == 9
    sh = sdist.AccessionShards(9, world=2, rank=1)
    assert sh.per == 8 and sh.padded_index().tolist() == [0, 1, 2, 3, 4, 5, 6, 7, 8] and sh.to_global(8) == 8
