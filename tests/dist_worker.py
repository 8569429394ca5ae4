"""
One rank of a 2-rank job on ONE GPU over gloo (tests/test_gpu_dist.py starts two of these) -- or, with SNPM_TEST_BACKEND=nccl
and SNPM_TEST_DEVICE=<rank>, one rank of a job whose ranks own a GPU each and talk over RCCL (tests/test_gpu_multi.py, which
switches itself on when the box has two GPUs).  Not a test module.

  device  the bench-shaped flow with the real kernels: a shard of a synthetic panel, results bound into torch
          tensors (bind_outputs), the library on torch's stream (set_stream), all_gather_into_tensor over padded,
          uneven shards, likelihood_device on the gathered vector; every mode, plus a slab-streamed carry.
  product the drop-in classes under an accession-sharded job: Genotyper (+ --refine) and CrossIdentifier on the toy
          DBs of the golden fixtures; rank 0 writes the files.
  rank    the C ABI's own group of ranks (snpm_group_create_rank = ncclCommInitRank, no torch.distributed): the rank's context
          on ITS device, the unique id through a file, uneven accession shards of a seeded DB, one snpm_group_gather_scores
          with the likelihoods over the full vector, windows gathered as rows.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def device_flow(out_dir):
    import torch
    import torch.distributed as dist
    from snpmatch_amd import engine, synth
    from snpmatch_amd.dist import AccessionShards
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    backend = os.environ.get("SNPM_TEST_BACKEND", "gloo")
    dev_index = int(os.environ.get("SNPM_TEST_DEVICE", "0"))
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    os.environ["SNPM_DEBUG_REEVAL"] = "2"            # every certified run re-evaluates two accessions per shard
    ctx = engine.Context(dev_index)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)
    n_snp, n_acc, seed, planted = 60_000, 1135, 4242, 417
    sh = AccessionShards(n_acc, world, rank, dev)
    if world == 2:
        assert sh.per == 568 and sh.n_local == (568 if rank == 0 else 567)      # uneven shards, padded to 568
    panel = engine.Panel(ctx, n_snp, sh.n_local)
    panel.fill_synthetic(seed, 0, sh.a0)
    wei = synth.sample_weights_twin(seed, 0, n_snp, planted)
    q = engine.Query(panel, None, wei)
    q.bind_outputs(sh.score_loc.data_ptr(), sh.ninfo_loc.data_ptr())
    lik = torch.zeros(sh.padded_len, dtype=torch.float64, device=dev)
    lrt = torch.zeros(sh.padded_len, dtype=torch.float64, device=dev)
    res = {}
    for name, mode in (("exact", engine.MODE_EXACT), ("strict", engine.MODE_STRICT), ("fast", engine.MODE_FAST)):
        q.run_device(1000, False, mode)
        fs, fn = sh.gather()
        ctx.likelihood_device(fs.data_ptr(), fn.data_ptr(), 1, sh.padded_len, lik.data_ptr(), lrt.data_ptr(), truncate=True)
        torch.cuda.synchronize()
        res[name + "_score"], res[name + "_ninfo"] = sh.unpad(fs), sh.unpad(fn)
        res[name + "_lik"], res[name + "_lrt"] = sh.unpad(lik), sh.unpad(lrt)
        if mode == engine.MODE_EXACT:
            res["exact_reeval"] = np.array([q.last_reeval()])
    # the same shard scored as three SNP slabs with a carry bound to the same tensors
    q.bind_outputs(None, None)
    slabs = [25_000, 25_000, 10_000]
    buf = engine.Panel(ctx, max(slabs), sh.n_local)
    starts = np.concatenate([[0], np.cumsum(slabs)])
    sc = engine.SlabScorer(buf, slabs, lambda k, p: p.fill_synthetic(seed, snp0=int(starts[k]), acc0=sh.a0, row0=0, nrows=slabs[k]),
                           lambda k: wei[starts[k]:starts[k + 1]])
    sc.carry.bind_outputs(sh.score_loc.data_ptr(), sh.ninfo_loc.data_ptr())
    _, _, info = sc.run(engine.MODE_EXACT)
    fs, fn = sh.gather()
    torch.cuda.synchronize()
    res["slab_score"], res["slab_ninfo"] = sh.unpad(fs), sh.unpad(fn)
    res["slab_second_pass"] = np.array([int(info["second_pass"])])
    np.savez(os.path.join(out_dir, "device_rank%d.npz" % rank), **res)
    sc.free()
    ctx.close()
    dist.destroy_process_group()


def product_flow(out_dir, golden):
    from snpmatch_amd import dist as sdist
    from snpmatch_amd.core import csmatch, parsers, snp_genotype, snpmatch

    def make_inputs(toy):
        inp = parsers.ParseInputs("")
        inp.load_snp_info(toy["s_chrs"], toy["s_pos"], toy["s_gt"], toy["s_wei"], toy["s_dp"])
        return inp

    def make_g(toy):
        return snp_genotype.Genotype.from_arrays(toy["snps"], toy["accs"], toy["positions"], toy["chrs"], toy["regions"])

    os.environ.setdefault("SNPMATCH_DIST_BACKEND", "gloo")
    job = sdist.init_from_env()
    assert job is not None and job.world == int(os.environ["WORLD_SIZE"])
    toy = np.load(os.path.join(golden, "toy_db.npz"))
    for skip in (False, True):
        gt = snpmatch.Genotyper(make_inputs(toy), make_g(toy), os.path.join(out_dir, "inbred%d" % skip), run_genotyper=True,
                                skip_db_hets=skip)
        assert len(gt.result.scores) == len(toy["accs"]) and gt.g.panel().n_acc < len(toy["accs"])
    toy = np.load(os.path.join(golden, "toy_db_refine.npz"))
    gt = snpmatch.Genotyper(make_inputs(toy), make_g(toy), os.path.join(out_dir, "refine"), run_genotyper=False)
    gt.filter_tophits()
    toy = np.load(os.path.join(golden, "toy_db_cross.npz"))
    for skip in (False, True):
        ci = csmatch.CrossIdentifier(make_inputs(toy), make_g(toy), "athaliana_tair10", 300000,
                                     os.path.join(out_dir, "cross%d" % skip), run_identifier=True, skip_db_hets=skip)
        assert len(ci.result.accs) == 30 + 45
    job.barrier()
    with open(os.path.join(out_dir, "product_rank%d.json" % job.rank), "w") as fh:
        json.dump({"writer": job.is_writer, "has_result_fine": hasattr(gt, "result_fine")}, fh)
    import torch.distributed as dist
    dist.destroy_process_group()


def rank_flow(out_dir):
    import time
    from snpmatch_amd import engine, synth
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev_index = int(os.environ.get("SNPM_TEST_DEVICE", "0"))
    ctx = engine.Context(dev_index)
    id_path = os.path.join(out_dir, "group_id.bin")
    if rank == 0:
        uid = engine.Group.unique_id()
        with open(id_path + ".tmp", "wb") as fh:
            fh.write(bytes(uid))
        os.replace(id_path + ".tmp", id_path)
    else:
        t0 = time.time()
        while not os.path.exists(id_path):
            assert time.time() - t0 < 120, "rank 0 never published the group id"
            time.sleep(0.05)
        uid = open(id_path, "rb").read()
    group = engine.Group.from_rank(ctx, uid, world, rank)
    assert (group.world, group.rank0, group.n_local) == (world, rank, 1) and "rccl" in group.transport.lower()
    n_snp, n_acc, seed = 30_000, int(os.environ.get("SNPM_TEST_N_ACC", "1135")), 99
    rng = np.random.default_rng(seed)
    db = rng.choice(np.array([-1, 0, 1, 2], dtype=np.int8), size=(n_snp, n_acc), p=[0.05, 0.60, 0.33, 0.02])
    codes = db[:, 5].copy()
    codes[codes < 0] = 0
    wei = synth.sample_weights(rng, codes, 0.8)
    a0, a1 = group.shard(n_acc, rank)
    res = {"bounds": np.array([a0, a1])}
    if a1 > a0:
        panel = engine.Panel.from_host(ctx, db, cols=(a0, a1))           # this rank's columns of the wider DB
        q = engine.Query(panel, None, wei)
    for name, mode in (("strict", engine.MODE_STRICT), ("exact", engine.MODE_EXACT)):
        if a1 > a0:
            d_s, d_n = q.run_device(1000, False, mode)
        else:
            d_s, d_n = 0, 0
        out = group.gather([d_s], [d_n], n_acc, truncate=True, likelihoods=True)
        for k in ("score", "ninfo", "lik", "lrt"):
            res[name + "_" + k] = out[k]
    np.savez(os.path.join(out_dir, "rank_flow_rank%d.npz" % rank), **res)
    group.free()
    ctx.close()


if __name__ == "__main__":
    if sys.argv[1] == "device":
        device_flow(sys.argv[2])
    elif sys.argv[1] == "rank":
        rank_flow(sys.argv[2])
    else:
        product_flow(sys.argv[2], sys.argv[3])
