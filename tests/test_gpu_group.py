"""
Multi-GPU behind the C ABI (-m gpu): snpm_group_* -- accession shards per member and ONE all-gather of the
per-accession results inside libsnpmatch_hip.so (reference: accession columns never interact,
core/snpmatch.py:84-88, the likelihood step needs the minimum over all accessions, :112).

A one-GPU box can form two kinds of group:
  * a size-1 group through RCCL (ncclCommInitAll / ncclCommInitRank with one rank: the communicator, the
    all-gather call and the pack / unpack kernels are the ones an 8-GPU job runs);
  * several members on device 0 with the LOOPBACK transport (device-to-device copies instead of ncclAllGather):
    the sharding, the padded tails, the pack / unpack indexing and the product path over a GroupPanel.
Everything is compared with the unsharded run bit for bit, and with the files of the unmodified reference.
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

from oracle import c_oracle
from snpmatch_amd import _lib, engine, synth
from snpmatch_amd.core import csmatch, snp_genotype, snpmatch

from test_gpu_pipeline import cmp_scores_table, cmp_window_table, make_g, make_inputs

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def rand_db(rng, n, n_acc):
    return rng.choice(np.array([-1, 0, 1, 2], dtype=np.int8), size=(n, n_acc), p=[0.05, 0.60, 0.33, 0.02])


def make_case(seed, n, n_acc, planted=5):
    rng = np.random.default_rng(seed)
    db = rand_db(rng, n, n_acc)
    codes = db[:, planted].copy()
    codes[codes < 0] = 0
    return db, synth.sample_weights(rng, codes, 0.8)


def test_rccl_group_of_one_equals_plain_run():
    """ncclCommInitAll with one device, then ncclCommInitRank with one rank: the gathered vectors are the plain run's"""
    db, wei = make_case(1, 30_000, 1135)
    want_s, want_n = c_oracle.genotyper(db, None, wei, 1000, False)
    for how in ("local", "rank"):
        if how == "local":
            group = engine.Group.local([0])
            ctx = group.contexts[0]
        else:
            ctx = engine.Context(0)
            group = engine.Group.from_rank(ctx, engine.Group.unique_id(), 1, 0)
        assert (group.world, group.rank0, group.n_local) == (1, 0, 1)
        assert "rccl" in group.transport.lower()
        assert group.shard(1135, 0) == (0, 1135)
        panel = engine.Panel.from_host(ctx, db)
        q = engine.Query(panel, None, wei)
        d_s, d_n = q.run_device(1000, False, engine.MODE_STRICT)
        out = group.gather([d_s], [d_n], 1135, truncate=True, likelihoods=True)
        assert np.array_equal(bits(out["score"]), bits(want_s)) and np.array_equal(out["ninfo"], want_n)
        lik, lrt = ctx.likelihood(want_s, want_n, truncate=True)
        assert np.array_equal(bits(out["lik"]), bits(lik)) and np.array_equal(bits(out["lrt"]), bits(lrt))
        assert int(np.nanargmin(out["lik"])) == 5
        # results left on the device: nothing copied, the gathered vectors are readable through their pointers
        d_s, d_n = q.run_device(1000, False, engine.MODE_EXACT)
        assert group.gather([d_s], [d_n], 1135, host=False) == {}
        gs, gn = group.gathered_ptrs(0)
        import torch
        t_l = torch.empty(1135, dtype=torch.float64, device="cuda:0")
        t_r = torch.empty(1135, dtype=torch.float64, device="cuda:0")
        ctx.likelihood_device(gs, gn, 1, 1135, t_l.data_ptr(), t_r.data_ptr(), truncate=True)
        ctx.synchronize()
        assert np.array_equal(bits(t_l.cpu().numpy()), bits(lik))
        q.free()
        panel.free()
        group.free()
        if how == "rank":
            ctx.close()


def test_group_errors_are_reported():
    lib = _lib.load()
    h = C.c_void_p()
    ids = (C.c_int * 2)(0, 0)
    assert lib.snpm_group_create_local(ids, 2, 0, C.byref(h)) == _lib.SNPM_ERR_BADARG       # one GPU twice needs loopback
    assert b"twice" in lib.snpm_group_last_error(None)
    assert lib.snpm_group_create_local(ids, 0, 0, C.byref(h)) == _lib.SNPM_ERR_BADARG
    with pytest.raises(AssertionError):
        engine.Group.from_rank(engine.default_context(), engine.Group.unique_id(), 2, 5)


@pytest.mark.parametrize("n_members,n_acc", [(3, 1135), (2, 257), (4, 1000)])
def test_loopback_group_equals_unsharded(n_members, n_acc):
    """members on device 0, uneven shards with padded tails (1135 / 3: 380 + 380 + 375; 257 / 2: 132 + 125)"""
    group = engine.Group.local([0] * n_members, loopback=True)
    assert group.transport == "loopback" and group.n_local == n_members
    group_equals_unsharded(group, n_members, n_acc)


def group_equals_unsharded(group, n_members, n_acc):
    """every product of a GroupPanel over `group` (one process, n_members members) against the unsharded run on device 0, bit for
    bit: reference-order and certified scores, windows, the --refine scan, in-silico crosses, batches.  Used with the loopback
    transport here and with RCCL over real devices by tests/test_gpu_multi.py; frees the group."""
    db, wei = make_case(n_acc, 40_000, n_acc)
    ctx = engine.Context(0)
    whole = engine.Panel.from_host(ctx, db)
    q = engine.Query(whole, None, wei)
    ss, sn = q.run(1000, False, engine.MODE_STRICT)
    es, en = q.run(1000, True, engine.MODE_EXACT)
    want_s, want_n = c_oracle.genotyper(db, None, wei, 1000, False)
    assert np.array_equal(bits(ss), bits(want_s)) and np.array_equal(sn, want_n)
    bounds = group.local_shards(n_acc)
    assert bounds[0][0] == 0 and bounds[-1][1] == n_acc and all(b[0] % 4 == 0 for b in bounds)
    assert all(bounds[i][1] == bounds[i + 1][0] for i in range(n_members - 1))
    gp = engine.GroupPanel.from_host(group, db)
    gq = gp.query(None, wei)
    s, n, info = gq.run(1000, False, engine.MODE_STRICT, return_info=True)
    assert np.array_equal(bits(s), bits(ss)) and np.array_equal(n, sn) and info["members"] == n_members
    s, n = gq.run(1000, True, engine.MODE_EXACT)
    assert np.array_equal(n, en) and np.array_equal(s.astype(np.int64), es.astype(np.int64))
    # gathered rows and windows
    rows = np.sort(np.random.default_rng(2).choice(len(db), size=7000, replace=False)).astype(np.int64)
    off = np.array([0, 100, 100, 2500, 7000], dtype=np.int64)
    want = engine.Query(whole, rows, wei[rows]).run_windows(off)
    got = gp.query(rows, wei[rows]).run_windows(off)
    for a, b in zip(got, want):
        assert np.array_equal(np.ascontiguousarray(a).view(np.uint64), np.ascontiguousarray(b).view(np.uint64))
    # --refine scan and in-silico crosses over accessions that live on different members
    cols = np.array([1, bounds[0][1] - 1, bounds[1][0], n_acc - 1])
    assert np.array_equal(gp.segregating_rows(cols), whole.segregating_rows(cols))
    best = np.array([5, n_acc - 2, bounds[1][0] + 1, 0, bounds[0][1] - 2])
    fs, fn = gp.query(rows, wei[rows]).f1_pairs(best)
    ws, wn = engine.Query(whole, rows, wei[rows]).f1_pairs(best)
    assert np.array_equal(bits(fs), bits(ws)) and np.array_equal(fn, wn)
    # many samples per call
    samples = [(rows[:3000], wei[rows[:3000]]), (rows, wei[rows])]
    a, b = engine.score_batch(gp, samples), engine.score_batch(whole, samples)
    for key in ("score", "ninfo", "lik", "lrt"):
        assert np.array_equal(np.ascontiguousarray(a[key]).view(np.uint64), np.ascontiguousarray(b[key]).view(np.uint64)), key
    gp.free()
    group.free()
    ctx.close()


@pytest.fixture
def three_members(monkeypatch):
    monkeypatch.setenv("SNPMATCH_GPUS", "0,0,0")
    monkeypatch.setenv("SNPMATCH_GROUP_LOOPBACK", "1")
    yield
    if engine._default_group is not None:
        engine._default_group.free()
        engine._default_group = None


def test_product_path_over_a_group_matches_reference_files(golden_dir, tmp_path, three_members):
    """Genotyper, --refine and CrossIdentifier when Genotype.panel() is a GroupPanel of three members: the reference's files"""
    toy = np.load(os.path.join(golden_dir, "toy_db.npz"))
    gold = json.load(open(os.path.join(golden_dir, "g2_inbred.json")))
    out = str(tmp_path / "inbred")
    g = make_g(toy)
    snpmatch.Genotyper(make_inputs(toy), g, out, run_genotyper=True)
    assert isinstance(g.panel(), engine.GroupPanel) and len(g.panel().members) == 3
    cmp_scores_table(open(out + ".scores.txt").read(), gold["inbred_skip0"]["scores.txt"])
    assert open(out + ".matches.json").read() == gold["inbred_skip0"]["matches.json"]
    toy = np.load(os.path.join(golden_dir, "toy_db_refine.npz"))
    gold = json.load(open(os.path.join(golden_dir, "g3_refine.json")))
    out = str(tmp_path / "refine")
    gt = snpmatch.Genotyper(make_inputs(toy), make_g(toy), out, run_genotyper=False)
    gt.filter_tophits()
    assert hasattr(gt, "result_fine") == gold["has_result_fine"]
    cmp_scores_table(open(out + ".scores.txt").read(), gold["scores.txt"])
    cmp_scores_table(open(out + ".refined.scores.txt").read(), gold["refined.scores.txt"])
    toy = np.load(os.path.join(golden_dir, "toy_db_cross.npz"))
    gold = json.load(open(os.path.join(golden_dir, "g5_cross.json")))["cross_skip0"]
    out = str(tmp_path / "cross")
    g = make_g(toy)
    csmatch.CrossIdentifier(make_inputs(toy), g, "athaliana_tair10", 300000, out, run_identifier=True)
    assert isinstance(g.panel(), engine.GroupPanel)
    cmp_window_table(open(out + ".windowscore.txt").read(), gold[".windowscore.txt"])
    cmp_scores_table(open(out + ".scores.txt").read(), gold[".scores.txt"])
    assert open(out + ".scores.txt.matches.json").read() == gold[".scores.txt.matches.json"]
    # many samples in one call over the group
    toy = np.load(os.path.join(golden_dir, "toy_db.npz"))
    outs = [str(tmp_path / "b0"), str(tmp_path / "b1")]
    snpmatch.genotype_batch([make_inputs(toy), make_inputs(toy)], make_g(toy), outs)
    gold = json.load(open(os.path.join(golden_dir, "g2_inbred.json")))["inbred_skip0"]
    for o in outs:
        cmp_scores_table(open(o + ".scores.txt").read(), gold["scores.txt"])
        assert open(o + ".matches.json").read() == gold["matches.json"]


def test_group_of_streamed_members_matches_reference_files(golden_dir, tmp_path, monkeypatch):
    """two GPUs' worth of members, each too small for its accession shard: every member streams its columns of the `.snpm` file in
    slabs (column range of a wider file, two half-buffers), the group gathers the carries' totals -- the reference's files"""
    from snpmatch_amd.core import snp_genotype
    monkeypatch.setenv("SNPMATCH_GPUS", "0,0")
    monkeypatch.setenv("SNPMATCH_GROUP_LOOPBACK", "1")
    monkeypatch.setenv("SNPM_HBM_BUDGET_GB", repr(2 * (1000 + 32) * 128 / 1e9))       # int8 shards of 28 / 22 accessions: 128-B rows
    monkeypatch.setenv("SNPMATCH_PACKED", "0")
    # (with the split layout of round 4 a 28-accession packed panel has 8-B rows and would fit this budget whole: the scenario
    # -- nothing fits, int8 slabs are streamed -- needs round 3's 256-B packed rows)
    monkeypatch.setenv("SNPM_PACKED_SPLIT", "0")
    try:
        toy = np.load(os.path.join(golden_dir, "toy_db.npz"))
        gold = json.load(open(os.path.join(golden_dir, "g2_inbred.json")))["inbred_skip0"]
        db = str(tmp_path / "toy.snpm")
        snp_genotype.save_native(db, toy["snps"], toy["accs"], toy["positions"], toy["chrs"], toy["regions"])
        g = snp_genotype.Genotype(db, None)
        out = str(tmp_path / "inbred")
        snpmatch.Genotyper(make_inputs(toy), g, out, run_genotyper=True)
        p = g.panel()
        assert isinstance(p, engine.GroupPanel) and [type(m).__name__ for m in p.members] == ["StreamedPanel"] * 2
        assert [m.n_acc for m in p.members] == [28, 22] and all(m.loads == 3 for m in p.members)
        cmp_scores_table(open(out + ".scores.txt").read(), gold["scores.txt"])
        assert open(out + ".matches.json").read() == gold["matches.json"]
        toy = np.load(os.path.join(golden_dir, "toy_db_cross.npz"))
        gold = json.load(open(os.path.join(golden_dir, "g5_cross.json")))["cross_skip0"]
        out = str(tmp_path / "cross")
        g = make_g(toy)
        csmatch.CrossIdentifier(make_inputs(toy), g, "athaliana_tair10", 300000, out, run_identifier=True)
        assert [type(m).__name__ for m in g.panel().members] == ["StreamedPanel"] * 2
        cmp_window_table(open(out + ".windowscore.txt").read(), gold[".windowscore.txt"])
        cmp_scores_table(open(out + ".scores.txt").read(), gold[".scores.txt"])
        assert open(out + ".scores.txt.matches.json").read() == gold[".scores.txt.matches.json"]
    finally:
        if engine._default_group is not None:
            engine._default_group.free()
            engine._default_group = None


def test_group_handle_survives_its_context():
    """snpm_destroy on the context of a rank-style group releases the member (communicator, buffers); the group handle then
    refuses work with SNPM_ERR_STATE and can still be freed -- the lifetime rule of panels / queries (include/snpmatch_hip.h)"""
    ctx = engine.Context(0)
    group = engine.Group.from_rank(ctx, engine.Group.unique_id(), 1, 0)
    ctx.close()
    with pytest.raises(_lib.SnpmError, match="outlived"):
        group.gather([0], [0], 8)
    group.free()
    # and the other order with live work in between
    ctx = engine.Context(0)
    group = engine.Group.from_rank(ctx, engine.Group.unique_id(), 1, 0)
    db, wei = make_case(3, 5000, 64)
    q = engine.Query(engine.Panel.from_host(ctx, db), None, wei)
    d_s, d_n = q.run_device(1000, False, engine.MODE_STRICT)
    out = group.gather([d_s], [d_n], 64)
    want_s, want_n = c_oracle.genotyper(db, None, wei, 1000, False)
    assert np.array_equal(bits(out["score"]), bits(want_s)) and np.array_equal(out["ninfo"], want_n)
    group.free()
    ctx.close()


def test_back_to_back_gathers_do_not_race():
    """60 gathers in a row without host synchronisation in between (results left on the devices), alternating two different
    shard results: every member's gathered vector is the one of ITS gather -- over the loopback transport (send buffers are
    rewritten while other members may still be copying the previous ones) and over an RCCL group of one rank"""
    import torch
    n_acc = 1135
    for n_members, loopback in ((3, True), (1, False)):
        group = engine.Group.local([0] * n_members, loopback=loopback)
        bounds = group.local_shards(n_acc)
        vals = []
        for v in range(2):
            vals.append([(torch.full((b[1] - b[0],), float(100 * v + i), dtype=torch.float64, device="cuda:0"),
                          torch.full((b[1] - b[0],), 7 * v + i, dtype=torch.int64, device="cuda:0")) for i, b in enumerate(bounds)])
        torch.cuda.synchronize()
        for it in range(60):
            v = it % 2
            group.gather([t[0].data_ptr() for t in vals[v]], [t[1].data_ptr() for t in vals[v]], n_acc, host=False)
        out = group.gather([t[0].data_ptr() for t in vals[1]], [t[1].data_ptr() for t in vals[1]], n_acc)
        want_s = np.concatenate([np.full(b[1] - b[0], 100.0 + i) for i, b in enumerate(bounds)])
        want_n = np.concatenate([np.full(b[1] - b[0], 7 + i) for i, b in enumerate(bounds)])
        assert np.array_equal(out["score"], want_s) and np.array_equal(out["ninfo"], want_n)
        group.free()


def test_unrequested_group_that_cannot_form_falls_back_to_one_gpu(golden_dir, tmp_path, monkeypatch):
    """several GPUs visible, a DB large enough to be spread over them, but the communicator cannot be formed (no RCCL, an
    unreachable peer): a job that did not ask for the GPUs runs on one of them -- the reference's files -- and a job that
    asked (SNPMATCH_GPUS) is told"""
    from snpmatch_amd.core import snp_genotype
    toy = np.load(os.path.join(golden_dir, "toy_db.npz"))
    gold = json.load(open(os.path.join(golden_dir, "g2_inbred.json")))["inbred_skip0"]

    def broken(n_members=None):
        raise RuntimeError("ncclCommInitAll failed: unhandled system error")

    monkeypatch.delenv("SNPMATCH_GPUS", raising=False)
    monkeypatch.setattr(snp_genotype, "GROUP_MIN_BYTES", 0)
    monkeypatch.setattr(engine, "group_devices", lambda: [0, 1])
    monkeypatch.setattr(engine, "default_group", broken)
    g = make_g(toy)
    out = str(tmp_path / "fallback")
    snpmatch.Genotyper(make_inputs(toy), g, out, run_genotyper=True)
    assert isinstance(g.panel(), engine.Panel)
    cmp_scores_table(open(out + ".scores.txt").read(), gold["scores.txt"])
    assert open(out + ".matches.json").read() == gold["matches.json"]
    monkeypatch.setenv("SNPMATCH_GPUS", "2")
    with pytest.raises(RuntimeError, match="ncclCommInitAll"):
        make_g(toy).panel()

