"""
Accession-axis sharding over the GPUs of one node (one process per GPU, torch.distributed; backend
"nccl" is RCCL over xGMI on ROCm, "gloo" for CPU rehearsals).

Every reduction of the hot path runs over the SNP axis (core/snpmatch.py:84-88), so accession columns
never interact: rank r owns columns [a0, a1) of every SNP row, scores them locally with no data-path
collective, and ONE all-gather of the per-accession results (score fp64, ninfo int64: 16 B per
accession, latency-bound) makes the full vectors available everywhere for the likelihood step, which
needs the global minimum (core/snpmatch.py:112).  Shard boundaries are multiples of 4 accessions (the
synthetic generator hashes accession quads); shards are padded to a common length for the collective.
"""
import numpy as np


def shard_bounds(n_acc, world, align=4):
    """[(a0, a1)] per rank and the common padded shard length."""
    per = ((n_acc + world - 1) // world + align - 1) // align * align
    return [(min(r * per, n_acc), min((r + 1) * per, n_acc)) for r in range(world)], per


class AccessionShards(object):
    """Buffers and the collective for one rank of an accession-sharded job."""

    def __init__(self, n_acc, world=1, rank=0, device="cpu", group=None, force_collective=False):
        import torch
        self.torch = torch
        self.n_acc, self.world, self.rank, self.group = int(n_acc), int(world), int(rank), group
        self.bounds, self.per = shard_bounds(n_acc, world)
        self.a0, self.a1 = self.bounds[rank]
        self.n_local = self.a1 - self.a0
        self.device = device
        # padded tail entries stay (score 0, ninfo 0): NaN likelihood, ignored by nanmin.  Both result vectors live in ONE
        # buffer of 8-byte words [2, per] (row 0: the fp64 scores, row 1: the int64 counts), so that one collective moves them
        self.pack_loc = torch.zeros((2, self.per), dtype=torch.int64, device=device)
        self.score_loc = self.pack_loc[0].view(torch.float64)
        self.ninfo_loc = self.pack_loc[1]
        self.collective = world > 1 or force_collective
        if self.collective:
            self.pack_all = torch.zeros((world, 2, self.per), dtype=torch.int64, device=device)
            self.score_all = torch.zeros(self.per * world, dtype=torch.float64, device=device)
            self.ninfo_all = torch.zeros(self.per * world, dtype=torch.int64, device=device)
        else:
            self.score_all, self.ninfo_all = self.score_loc, self.ninfo_loc

    @property
    def padded_len(self):
        return self.per * self.world

    def set_local(self, score, ninfo):
        """host results of this rank's shard -> the local buffers (CPU rehearsals / host API)."""
        t = self.torch
        self.score_loc[:self.n_local] = t.as_tensor(np.asarray(score, dtype=np.float64), device=self.device)
        self.ninfo_loc[:self.n_local] = t.as_tensor(np.asarray(ninfo, dtype=np.int64), device=self.device)

    def gather(self):
        """the single collective of the path: ONE all-gather of the packed (score, ninfo) words along the accession axis,
        then two strided device copies into the contiguous full-length vectors"""
        if self.collective:
            import torch.distributed as dist
            dist.all_gather_into_tensor(self.pack_all.view(-1), self.pack_loc.view(-1), group=self.group)
            self.score_all.view(self.world, self.per).copy_(self.pack_all[:, 0, :].view(self.torch.float64))
            self.ninfo_all.view(self.world, self.per).copy_(self.pack_all[:, 1, :])
        return self.score_all, self.ninfo_all

    def padded_index(self):
        """positions of accessions 0..n_acc-1 inside the padded gathered vectors"""
        idx = [np.arange(a0, a1) - a0 + r * self.per for r, (a0, a1) in enumerate(self.bounds)]
        return np.concatenate(idx) if idx else np.zeros(0, dtype=int)

    def unpad(self, full):
        """gathered padded tensor -> numpy array of length n_acc in accession order"""
        return full.detach().cpu().numpy()[self.padded_index()]

    def to_global(self, padded_pos):
        r, off = divmod(int(padded_pos), self.per)
        return self.bounds[r][0] + off


def sharded_genotyper_scores(score_local_fn, n_acc, world, rank, device="cpu", group=None):
    """Run `score_local_fn(a0, a1) -> (score, ninfo)` on this rank's accession shard and return the
    full-length (score, ninfo) numpy vectors on every rank."""
    sh = AccessionShards(n_acc, world, rank, device, group)
    s, n = score_local_fn(sh.a0, sh.a1)
    sh.set_local(s, n)
    fs, fn = sh.gather()
    return sh.unpad(fs), sh.unpad(fn)


def sharded_window_scores(score_local_fn, n_acc, n_win, world, rank, device="cpu", group=None):
    """Per-window variant (core/csmatch.py:80-95): `score_local_fn(a0, a1) -> (score [n_win, a1 - a0],
    ninfo [n_win, a1 - a0])` on this rank's shard; ONE all-gather of the packed words [2, n_win, per] along the
    accession axis; returns the full (score [n_win, n_acc], ninfo [n_win, n_acc]) numpy arrays on every rank."""
    import torch
    import torch.distributed as dist
    sh = AccessionShards(n_acc, world, rank, "cpu", group)          # bounds only
    s, n = score_local_fn(sh.a0, sh.a1)
    loc = torch.zeros((2, n_win, sh.per), dtype=torch.int64, device=device)
    loc[0, :, :sh.n_local] = torch.as_tensor(np.ascontiguousarray(np.asarray(s, dtype=np.float64).reshape(n_win, sh.n_local)).view(np.int64),
                                             device=device)
    loc[1, :, :sh.n_local] = torch.as_tensor(np.asarray(n, dtype=np.int64).reshape(n_win, sh.n_local), device=device)
    if world > 1:
        every = torch.zeros((world, 2, n_win, sh.per), dtype=torch.int64, device=device)
        dist.all_gather_into_tensor(every.view(-1), loc.view(-1), group=group)
    else:
        every = loc.reshape(1, 2, n_win, sh.per)
    idx = sh.padded_index()
    host = every.cpu().numpy()          # [world, 2, n_win, per] -> [n_win, world * per] -> accession order

    def full(k, dtype):
        return np.ascontiguousarray(host[:, k].transpose(1, 0, 2).reshape(n_win, world * sh.per)[:, idx]).view(dtype)
    return full(0, np.float64), full(1, np.int64)


# ----------------------------------------------------------------------------------------------------------
# The accession-sharded job as the product path sees it (Genotype.panel, Genotyper, CrossIdentifier).
class Job(object):
    """One process per GPU; this rank holds accessions [a0, a1) of every DB it opens.  Created by ``init_from_env``
    (CLI under torch.distributed.run) or ``attach`` (a caller that initialised torch.distributed itself)."""

    def __init__(self, world, rank, group=None, device="cpu"):
        self.world, self.rank, self.group, self.device = int(world), int(rank), group, device
        self._tmp = None

    @property
    def is_writer(self):
        return self.rank == 0

    def bounds(self, n_acc):
        b, _ = shard_bounds(n_acc, self.world)
        return b[self.rank]

    def barrier(self):
        import torch.distributed as dist
        dist.barrier(group=self.group)

    def output_prefix(self, prefix):
        """rank 0 writes the result files; the other ranks run the same code against a private scratch prefix"""
        if self.is_writer or prefix is None:
            return prefix
        import os
        import tempfile
        if self._tmp is None:
            import atexit
            import shutil
            self._tmp = tempfile.mkdtemp(prefix="snpmatch_rank%d_" % self.rank)
            atexit.register(shutil.rmtree, self._tmp, True)          # the scratch copies of a non-writer rank go with the process
        return os.path.join(self._tmp, os.path.basename(prefix))

    def gather_scores(self, score_loc, ninfo_loc, n_acc):
        """local (score, ninfo) of this rank's accessions -> the full vectors on every rank (ONE all-gather each)"""
        sh = AccessionShards(n_acc, self.world, self.rank, self.device, self.group)
        assert len(score_loc) == sh.n_local, "local results do not match this rank's accession shard"
        sh.set_local(score_loc, ninfo_loc)
        fs, fn = sh.gather()
        return sh.unpad(fs), sh.unpad(fn)

    def gather_windows(self, score_loc, ninfo_loc, n_acc):
        n_win = score_loc.shape[0]
        return sharded_window_scores(lambda a0, a1: (score_loc, ninfo_loc), n_acc, n_win, self.world, self.rank,
                                     self.device, self.group)

    def all_gather_bytes(self, arr):
        """uint8 array of the same shape on every rank -> [world, ...] on every rank"""
        import torch
        import torch.distributed as dist
        arr = np.ascontiguousarray(arr, dtype=np.uint8)
        loc = torch.as_tensor(arr.reshape(-1), device=self.device)
        out = torch.empty(self.world * loc.numel(), dtype=torch.uint8, device=self.device)
        dist.all_gather_into_tensor(out, loc, group=self.group)
        return out.cpu().numpy().reshape((self.world,) + arr.shape)

    def sum_bytes(self, arr):
        """element-wise sum over ranks of a uint8 array (each element is non-zero on at most one rank)"""
        import torch
        import torch.distributed as dist
        t = torch.as_tensor(np.ascontiguousarray(arr, dtype=np.uint8), device=self.device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t.cpu().numpy()


_job = None


def job():
    """the active accession-sharded job, or None in a single-process run"""
    return _job


def attach(group=None, device="cpu"):
    """Use the caller's torch.distributed process group for the product path (every rank must call it)."""
    global _job
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    _job = Job(world, rank, group, device) if world > 1 else None
    return _job


def detach():
    global _job
    _job = None


def init_from_env(backend=None):
    """CLI entry under ``python -m torch.distributed.run``: WORLD_SIZE > 1 starts the process group (backend nccl =
    RCCL when a GPU per rank is visible, else gloo) and pins this rank's default context to LOCAL_RANK."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return None
    import torch
    import torch.distributed as dist
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if backend is None:
        backend = os.environ.get("SNPMATCH_DIST_BACKEND", "nccl" if torch.cuda.device_count() >= world else "gloo")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29544")
    if backend == "nccl":
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
        dist.init_process_group("nccl", device_id=dev)
    else:
        dev = "cpu"
        if torch.cuda.device_count() < world:
            os.environ["SNPMATCH_DEVICE"] = "0"         # rehearsal on a one-GPU box: every rank on device 0
        dist.init_process_group("gloo")
    return attach(None, dev)
