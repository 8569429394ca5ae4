"""
DB genotype container (reference: core/snp_genotype.py:24-68,188-211 and the HDF5Genotype duck type of
pygwas/genotype.py:534-673).

``Genotype(hdf5_file, hdf5_acc_file)`` keeps the reference's constructor and the attributes the hot
path uses -- ``g.g.snps[idx, :]``, ``g.g.accessions``, ``g.g.positions``, ``g.g.chrs``,
``g.g.chr_regions``, ``g.g.chromosomes``, ``g.g_acc.snps[:, i]``, ``g.accessions``, ``g.chrs`` -- and
adds ``g.panel(ctx)``: the int8 matrix resident in HBM (uploaded once through the pinned-host staging
path of libsnpmatch_hip), which is what ``Genotyper`` / ``CrossIdentifier`` score against.

On-disk formats
  * native flat panel ``<name>.snpm/``: ``snps.npy`` = int8 [num_snps, num_accessions] C-order (read natively by the
    library's loader threads, streamed to the GPU slab by slab) or ``snps.p2.npy`` = the same matrix with 2 bits per call
    (``save_native(..., packed=True)`` / ``makedb-native --packed``: a quarter of the disk and of the bytes a load moves),
    + ``meta.npz`` (accessions, positions, chrs, chr_regions).
  * ``.npz`` with the same keys plus ``snps`` (small DBs, tests).
  * the reference's HDF5 layout (pygwas/genotype.py:310-326: ``snps`` in lzf chunks of (1000, num_accessions), ``positions``
    with attrs ``chrs`` / ``chr_regions``, ``accessions``), read by the library's own HDF5 reader (``snpmatch_amd.h5``,
    csrc/snpm_h5.cpp) -- h5py is only a fallback for files in formats that reader refuses.
"""
import logging
import os
import re
from glob import glob

import numpy as np

from . import parsers
from .. import _lib

log = logging.getLogger(__name__)
GROUP_MIN_BYTES = 1 << 30          # DBs of at least this many calls are spread over the visible GPUs without being asked to
chunk_size = 1000


class MemGenotype(object):
    """Array-backed stand-in for pygwas HDF5Genotype (attributes used by inbred / cross only)."""

    def __init__(self, snps, accessions, positions, chrs, chr_regions):
        self.snps = snps
        acc = np.asarray(accessions)
        self.accessions = acc if acc.dtype.kind == "S" else np.char.encode(acc.astype("U"), "utf-8")
        self.positions = np.asarray(positions, dtype="i4")
        self.chrs = np.asarray(chrs).astype("U")
        self.chr_regions = np.asarray(chr_regions, dtype=np.int64).reshape(-1, 2)

    @property
    def chromosomes(self):
        """one chromosome name per SNP row (pygwas/genotype.py:156-161), as an array"""
        reps = (self.chr_regions[:, 1] - self.chr_regions[:, 0]).astype(int)
        return np.repeat(self.chrs, reps)

    @property
    def num_snps(self):
        return self.snps.shape[0]


class PackedRows(object):
    """Host view of the 2-bit matrix of a packed flat panel (``snps.p2.npy``: uint8 [n_snp, ceil(n_acc / 4)], field f of byte b =
    accession 4 b + f, 0 ref / 1 alt / 2 het / 3 missing): behaves like the int8 matrix for the reads the host side makes
    (``snps[idx, :]``, ``snps[:, i]``, slices), unpacking what is asked for."""

    def __init__(self, packed, n_acc):
        self.packed, self.shape, self.dtype, self.ndim = packed, (int(packed.shape[0]), int(n_acc)), np.dtype(np.int8), 2

    def __len__(self):
        return self.shape[0]

    @staticmethod
    def unpack(block, n_acc, a0=0):
        """uint8 rows holding accessions a0 .. (a0 % 4 == 0) -> int8 [n, n_acc]"""
        block = np.ascontiguousarray(block)
        fields = (block[:, :, None] >> np.array([0, 2, 4, 6], dtype=np.uint8)) & 3
        out = fields.reshape(block.shape[0], -1)[:, :n_acc].astype(np.int8)
        out[out == 3] = -1
        return out

    def __getitem__(self, key):
        if not isinstance(key, tuple):
            key = (key, slice(None))
        rows, cols = key
        one_row = isinstance(rows, (int, np.integer))
        block = self.packed[[rows], :] if one_row else self.packed[rows, :]
        if isinstance(cols, (int, np.integer)):
            c = int(cols) + (self.shape[1] if cols < 0 else 0)
            v = ((np.asarray(block[:, c // 4]) >> (2 * (c % 4))) & 3).astype(np.int8)
            v[v == 3] = -1
            return v[0] if one_row else v
        full = self.unpack(block, self.shape[1])[:, cols]
        return full[0] if one_row else full

    def __array__(self, dtype=None, copy=None):
        a = self[:, :]
        return a.astype(dtype) if dtype is not None else a


def save_native(path, snps, accessions, positions, chrs, chr_regions, packed=False):
    """Write the native flat panel directory ``path`` (conventionally ``*.snpm``): ``snps.npy`` (int8 [n_snp, n_acc]) or, with
    ``packed``, ``snps.p2.npy`` (2 bits per call: a quarter of the disk and of the bytes a load moves; only for DBs whose codes
    are -1 / 0 / 1 / 2) plus ``meta.npz``."""
    os.makedirs(path, exist_ok=True)
    n_snp, n_acc = snps.shape
    for stale in ("snps.npy", "snps.p2.npy"):
        if os.path.exists(os.path.join(path, stale)):
            os.remove(os.path.join(path, stale))
    if packed:
        mm = np.lib.format.open_memmap(os.path.join(path, "snps.p2.npy"), mode="w+", dtype=np.uint8, shape=(n_snp, (n_acc + 3) // 4))
        for r0 in range(0, n_snp, 1 << 16):
            mm[r0:r0 + (1 << 16)] = _lib.pack_rows_host(np.asarray(snps[r0:r0 + (1 << 16)]))
    else:
        mm = np.lib.format.open_memmap(os.path.join(path, "snps.npy"), mode="w+", dtype=np.int8, shape=(n_snp, n_acc))
        for r0 in range(0, n_snp, 1 << 16):
            mm[r0:r0 + (1 << 16)] = snps[r0:r0 + (1 << 16)]
    mm.flush()
    del mm
    np.savez(os.path.join(path, "meta.npz"), accessions=np.asarray(accessions).astype("S"),
             positions=np.asarray(positions, dtype="i4"), chrs=np.asarray(chrs).astype("S"),
             chr_regions=np.asarray(chr_regions, dtype=np.int64))


def _load_any(path):
    if os.path.isdir(path):
        meta = np.load(os.path.join(path, "meta.npz"))
        p2 = os.path.join(path, "snps.p2.npy")
        if os.path.exists(p2):                           # packed flat panel: 2 bits per call on disk
            snps = PackedRows(np.load(p2, mmap_mode="r"), len(meta["accessions"]))
            g = MemGenotype(snps, meta["accessions"], meta["positions"], meta["chrs"].astype("U"), meta["chr_regions"])
            g.npy_packed_path = p2
            return g
        snps = np.load(os.path.join(path, "snps.npy"), mmap_mode="r")
        g = MemGenotype(snps, meta["accessions"], meta["positions"], meta["chrs"].astype("U"), meta["chr_regions"])
        g.npy_path = os.path.join(path, "snps.npy")      # lets Genotype.panel() stream the file natively
        return g
    if path.endswith(".npz"):
        d = parsers._StoredNpz.open(path) or np.load(path)       # stored members: views of one memory map (no pass through zipfile)
        return MemGenotype(d["snps"], d["accessions"], d["positions"], np.asarray(d["chrs"]).astype("U"), d["chr_regions"])
    # The reference's HDF5 DB (pygwas/genotype.py:310-326, :534-673): read by the library's own reader (csrc/snpm_h5.cpp) -- no
    # h5py / libhdf5 needed, and the chunks go from the file through the loader's threads into the staging slabs.  Files in a
    # form that reader does not take (unlimited dimensions or more than 8 group members in the "latest" HDF5 file format, other
    # filters) go through h5py where it is installed.
    from .. import h5 as native_h5
    try:
        f = native_h5.File(path)
        g = MemGenotype(f["snps"], f["accessions"][:], f["positions"][:], f["positions"].attrs["chrs"].astype("U"),
                        f["positions"].attrs["chr_regions"])
        g.h5_source = (f, "snps")               # lets Genotype.panel() load rows natively (snpm_panel_load_h5)
        return g
    except IOError as native_error:
        try:
            import h5py
        except ImportError:
            raise IOError("%s; h5py is not installed either -- convert the DB where it is (python -m snpmatch_amd makedb-native)"
                          % native_error)
        h5 = h5py.File(path, "r")
        return MemGenotype(h5["snps"], h5["accessions"][:], h5["positions"][:],
                           h5["positions"].attrs["chrs"].astype("U"), h5["positions"].attrs["chr_regions"])


def load_genotype_files(h5file, hdf5_acc_file=None):
    return Genotype(h5file, hdf5_acc_file)


class Genotype(object):

    def __init__(self, hdf5_file, hdf5_acc_file):
        assert hdf5_file is not None or hdf5_acc_file is not None, "Provide atleast one hdf5 genotype file"
        self._panel = None
        if hdf5_file is None:
            assert os.path.exists(hdf5_acc_file), "Path to %s seems to be broken" % hdf5_acc_file
            self.g_acc = _load_any(hdf5_acc_file)
            return None
        assert os.path.exists(hdf5_file), "Path to %s seems to be broken" % hdf5_file
        self.g = _load_any(hdf5_file)
        if hdf5_acc_file is None:
            hdf5_acc_file = re.sub(r'\.hdf5$', '', hdf5_file) + '.acc.hdf5'
            if len(glob(hdf5_acc_file)) > 0:
                self.g_acc = _load_any(hdf5_acc_file)
            else:
                self.g_acc = self.g        # flat panels serve rows and columns from the same matrix
        else:
            self.g_acc = _load_any(hdf5_acc_file)
        self.accessions = self.g.accessions.astype('U')
        self.chrs = self.g.chrs.astype('U')

    @classmethod
    def from_arrays(cls, snps, accessions, positions, chrs, chr_regions):
        self = cls.__new__(cls)
        self._panel = None
        self.g = MemGenotype(snps, accessions, positions, chrs, chr_regions)
        self.g_acc = self.g
        self.accessions = self.g.accessions.astype('U')
        self.chrs = self.g.chrs.astype('U')
        return self

    # ------------------------------------------------------------------ device residency
    def panel(self, ctx=None, packed=None):
        """The DB matrix resident in HBM (created on first use; slabs go through pinned staging).
        ``packed``: True = 2 bits per call, False = a byte per call, None (default) = environment SNPMATCH_PACKED
        ("1" / "0"; unset or "auto": packed when the DB's codes allow it -- same results bit for bit, a quarter of the HBM and of
        the bytes over PCIe, the scans 2-3 x faster; DBs with codes other than -1/0/1/2 stay int8).
        Where the accession columns live:
          * under ``torch.distributed.run`` (``dist.job()``): this rank's accession shard on this rank's GPU;
          * several GPUs visible to ONE process (SNPMATCH_GPUS, default all): an ``engine.GroupPanel`` -- a shard per
            GPU, results joined by one RCCL all-gather inside the library (no launcher);
          * otherwise: the whole matrix on one GPU."""
        from .. import dist, engine
        if self._panel is None or getattr(self._panel, "h", None) is None:
            if packed is None:
                env = os.environ.get("SNPMATCH_PACKED", "auto").strip().lower()
                packed = None if env in ("", "auto") else env != "0"
            n_acc = len(self.accessions)
            job = dist.job()
            self._shard = job.bounds(n_acc) if job else None
            if self._shard is not None:
                # accession-sharded job (one process per GPU): this rank holds columns [a0, a1) of every SNP row
                assert self._shard[1] > self._shard[0], "more ranks than accession quads: this rank's shard is empty"
                self._panel = self._member_panel(ctx or engine.default_context(), self._shard[0], self._shard[1], packed)
                return self._panel
            group = None
            # several GPUs from one process: when asked for (SNPMATCH_GPUS), or by default for DBs of at least 1 GiB -- below
            # that one GPU loads and scores the DB faster than a communicator is set up
            big = int(self.g.snps.shape[0]) * n_acc >= GROUP_MIN_BYTES
            asked = os.environ.get("SNPMATCH_GPUS", "") != ""
            if ctx is None and (big or asked):
                ids = engine.group_devices()
                if ids is not None:
                    try:
                        group = engine.default_group(engine.GroupPanel.usable_members(n_acc, len(ids)))
                    except (RuntimeError, OSError) as e:
                        # several GPUs were found but their communicator could not be formed (RCCL missing or unable to reach
                        # a peer): a job that did not ask for them runs on one GPU, a job that did is told
                        if asked:
                            raise
                        log.warning("the %d visible GPUs cannot form a group (%s): using one GPU", len(ids), e)
                        group = None
            if group is not None:
                self._panel = engine.GroupPanel.build(group, n_acc, lambda c, a0, a1: self._member_panel(c, a0, a1, packed))
            else:
                self._panel = self._member_panel(ctx or engine.default_context(), 0, n_acc, packed)
        return self._panel

    def _member_panel(self, ctx, a0, a1, packed):
        """Columns [a0, a1) of the DB on one GPU, planned by the HBM the GPU has free (SNPM_HBM_BUDGET_GB overrides; 1 GB =
        1e9 bytes): the int8 matrix whole -> the 2-bit packed matrix whole (same results, a quarter of the bytes; DBs with
        call codes other than -1/0/1/2 cannot take this step) -> SNP slabs streamed from the file / array through two
        half-buffers (``engine.StreamedPanel``; the reference reads any size through g.g.snps[idx, :])."""
        from .. import engine
        npy = getattr(self.g, "npy_path", None)          # native flat panel: file -> pinned slabs -> HBM
        h5_source = getattr(self.g, "h5_source", None)   # the reference's HDF5 file: chunks -> loader threads -> pinned slabs -> HBM
        p2 = getattr(self.g, "npy_packed_path", None)    # packed flat panel: the file's 2-bit rows travel as they are
        store = (engine.RowStore(npy=npy) if npy else engine.RowStore(npy_packed=(p2, len(self.accessions))) if p2
                 else engine.RowStore(h5=h5_source) if h5_source else engine.RowStore(snps=self.g.snps))
        env = os.environ.get("SNPM_HBM_BUDGET_GB", "")
        budget = int(float(env) * 1e9) if env else int(0.85 * ctx.mem_info()[0])
        n_loc = a1 - a0

        def need(pk):
            return (store.n_snp + 32) * ctx.row_pitch(n_loc, pk) + 256

        # packed None (auto): the packed panel first -- it loads faster (a quarter of the bytes cross PCIe, packed by the host
        # threads) and scans faster -- then int8; False: int8 first, packed only when int8 does not fit; True: packed
        unpackable = False
        for pk in ([True, False] if packed is None else [True] if packed else [False, True]):
            if need(pk) > budget:
                continue
            try:
                return engine.Panel.from_store(ctx, store, packed=pk, cols=(a0, a1))
            except AssertionError:
                if not pk:
                    raise
                log.info("DB holds codes a packed panel cannot store; using the int8 panel")
                unpackable = True
                if packed and need(False) <= budget:
                    return engine.Panel.from_store(ctx, store, packed=False, cols=(a0, a1))
        log.info("DB shard of %.1f GB does not fit the HBM budget of %.1f GB: streaming SNP slabs", need(False) / 1e9, budget / 1e9)
        # slabs: packed when asked for, or when the file itself is packed; in auto mode an int8 source streams as int8 (a call
        # code a packed panel refuses could otherwise surface in the middle of a job, slabs after the first are not probed)
        if (packed or (packed is None and p2)) and not unpackable:
            sp = None
            try:
                sp = engine.StreamedPanel(ctx, store, cols=(a0, a1), packed=True, budget_bytes=budget)
                sp.store.load(sp.halves[0], sp.cols, (0, min(sp.rows_cap, store.n_snp)), 0)     # probe for codes a packed panel refuses
                return sp
            except AssertionError as e:
                # (the constructor itself asserts when the budget holds no row: then there is nothing to free)
                log.info("packed slabs not usable (%s); streaming int8 slabs", e)
                if sp is not None:
                    sp.free()
        return engine.StreamedPanel(ctx, store, cols=(a0, a1), packed=False, budget_bytes=budget)

    # ------------------------------------------------------------------ position intersection (a5)
    def _region_is_increasing(self, ci, pos):
        """DB positions of chromosome ``ci`` strictly increasing?  (checked once per DB object)"""
        cache = self.__dict__.setdefault("_increasing_regions", {})
        if ci not in cache:
            cache[ci] = bool(len(pos) < 2 or np.all(pos[1:] > pos[:-1]))
        return cache[ci]

    def get_positions_idxs(self, commonSNPsCHR, commonSNPsPOS, _parsed=None):
        """(db_row_idx, sample_idx) of the positions present in both; core/snp_genotype.py:43-44.
        Same result as ``get_common_positions(chromosomes, positions, ...)`` without materialising one
        chromosome string per DB row: the DB side is walked region by region (pygwas chr_regions).
        ``_parsed``: the ``ParseInputs`` these arrays belong to, when its chromosome names are already filtered
        (``Genotyper`` / ``CrossIdentifier`` call ``filter_chr_names`` on construction): nothing is copied or re-derived."""
        if _parsed is not None and getattr(_parsed, "g_chr_codes", None) is not None and len(_parsed.g_chr_codes) == len(commonSNPsPOS):
            ins = _parsed
        else:
            ins = parsers.ParseInputs("")
            ins.load_snp_info(snpCHR=commonSNPsCHR, snpPOS=commonSNPsPOS, snpGT="", snpWEI=np.nan, DPmean=0)
            ins.filter_chr_names()
        db_ids = self.__dict__.get("_db_chr_ids")
        if db_ids is None:                       # once per DB object
            db_ids = self._db_chr_ids = np.array([re.sub("chr", "", c, flags=re.IGNORECASE) for c in self.g.chrs.astype("U").tolist()], dtype="str")
        sample_ids = ins.g_chrs_ids.tolist()
        sample_pos = np.ascontiguousarray(ins.pos, dtype=np.int64)
        positions = self.__dict__.get("_positions_i64")
        if positions is None:                    # one int64 copy per DB object (HDF5 stores int32)
            positions = self._positions_i64 = np.ascontiguousarray(self.g.positions, dtype=np.int64)
        regions = np.asarray(self.g.chr_regions)
        idx1 = [np.zeros(0, dtype=int)]
        idx2 = [np.zeros(0, dtype=int)]
        seen = set()
        for ci, cid in enumerate(db_ids):
            if cid in seen:             # a chromosome id listed twice: fall back to the generic path
                return self.get_common_positions(self.g.chromosomes, positions, commonSNPsCHR, commonSNPsPOS)
            seen.add(cid)
            if regions[ci][1] <= regions[ci][0] or cid not in ins.g_chrs_ids:
                continue
            s, e = int(regions[ci][0]), int(regions[ci][1])
            ix2 = np.flatnonzero(ins.g_chr_codes == sample_ids.index(cid))
            p1 = positions[s:e]
            if len(ix2) and ix2[-1] - ix2[0] + 1 == len(ix2):       # one run of the sorted input: a view, no gather
                p2 = sample_pos[ix2[0]:ix2[-1] + 1]
            else:
                p2 = sample_pos[ix2]
            # native sorted merge (strictly increasing inputs); the DB side is verified once per chromosome, after
            # which a short sample list is located by galloping search instead of a walk over every DB position
            merged = _lib.intersect_sorted(p1, p2, a_verified=self._region_is_increasing(ci, p1))
            if merged is not None:
                idx1.append(s + merged[0])
                idx2.append(ix2[merged[1]])
            else:                                           # the reference's np.in1d pair, quirks included
                idx1.append(s + np.where(np.isin(p1, p2, assume_unique=True))[0])
                idx2.append(ix2[np.where(np.isin(p2, p1, assume_unique=True))[0]])
        return (np.concatenate(idx1).astype(int), np.concatenate(idx2).astype(int))

    @staticmethod
    def get_common_positions(input_1_chr, input_1_pos, input_2_chr, input_2_pos):
        """Rows of input 1 and of input 2 whose (chromosome, position) occurs in both, chromosomes in the order in
        which input 1 first names them (the contract of core/snp_genotype.py:46-68; 'Chr1' / 'chr1' / '1' are one
        chromosome).  Each side is grouped once (one stable sort of its chromosome index); a chromosome both sides
        hold is intersected by the native sorted merge when both position lists are strictly increasing, otherwise by
        the membership masks the reference builds (np.isin with assume_unique, quirks on repeated positions included)."""
        assert len(input_1_chr) == len(input_1_pos), "Both chromosome and position array provided should be of same length"
        assert len(input_2_chr) == len(input_2_pos), "Both chromosome and position array provided should be of same length"
        side1, side2 = _rows_by_chromosome(input_1_chr, input_1_pos), _rows_by_chromosome(input_2_chr, input_2_pos)
        hits1, hits2 = [np.zeros(0, dtype=int)], [np.zeros(0, dtype=int)]
        for cid in side1.order:
            if cid not in side2.rows:
                continue
            rows1, rows2 = side1.rows[cid], side2.rows[cid]
            pos1, pos2 = side1.pos[rows1], side2.pos[rows2]
            merged = _lib.intersect_sorted(pos1, pos2)
            if merged is not None:
                hits1.append(rows1[merged[0]])
                hits2.append(rows2[merged[1]])
            else:
                hits1.append(rows1[np.isin(pos1, pos2, assume_unique=True)])
                hits2.append(rows2[np.isin(pos2, pos1, assume_unique=True)])
        return (np.concatenate(hits1).astype(int), np.concatenate(hits2).astype(int))

    def get_matching_accs_ix(self, accs, return_np=False):
        acc_ix = []
        for ea in accs:
            t_ix = np.where(self.accessions == ea)[0]
            acc_ix.append(None if len(t_ix) == 0 else t_ix[0])
        if return_np:
            acc_ix = np.array([a for a in acc_ix if a is not None], dtype="int")
        return acc_ix

    # ------------------------------------------------------------------ --refine support
    def identify_segregating_snps(self, accs_ix):
        """DB rows where the given accessions do not all carry the same informative call
        (core/snp_genotype.py:188-211, segregting_snps :378-383) -- scanned on the device, where the DB lives."""
        mask = self.segregating_mask(accs_ix)
        return None if mask is None else np.where(mask)[0]

    def segregating_mask(self, accs_ix):
        """the same as a mask over the DB rows (what ``Genotyper.filter_tophits`` indexes with its matched rows: turning 11M
        flags into 7M indices and testing 200k rows against them cost 50 ms of a --refine run, the scan itself 1 ms)"""
        assert type(accs_ix) is np.ndarray, "provide an np array for list of indices to be considered"
        assert len(accs_ix) > 1, "polymorphism happens in more than 1 line"
        if len(accs_ix) > (len(self.accessions) / 2):
            return None
        panel = self.panel()                       # resident after the genome-wide pass; uploaded now otherwise
        shard = getattr(self, "_shard", None)      # (a GroupPanel combines its members' scans itself)
        if shard is not None:
            # accession-sharded: every rank scans the listed accessions it holds; a row segregates when some
            # rank saw two different calls, or two ranks saw different ones
            from .. import dist
            accs_ix = np.asarray(accs_ix)
            local = accs_ix[(accs_ix >= shard[0]) & (accs_ix < shard[1])] - shard[0]
            mask, first = panel.segregating_first(local)
            both = dist.job().all_gather_bytes(np.stack([mask, first]))          # [world, 2, n_snp]
            firsts = both[:, 1, :]
            seen = firsts != 0xFF
            lo = np.where(seen, firsts, 255).min(axis=0)
            hi = np.where(seen, firsts, 0).max(axis=0)
            return both[:, 0, :].any(axis=0) | (seen.any(axis=0) & (lo != hi))
        return panel.segregating_rows(accs_ix)                      # one device scan over the listed columns (k_segregating)


class _ChromosomeRows(object):
    """One side of a position intersection: ``order`` = chromosome ids as first named, ``rows[id]`` = its row numbers
    in input order, ``pos`` = all positions as integers."""
    __slots__ = ("order", "rows", "pos")


def _rows_by_chromosome(chrs, pos):
    ins = parsers.ParseInputs("")
    ins.load_snp_info(snpCHR=chrs, snpPOS=pos, snpGT="", snpWEI=np.nan, DPmean=0)
    ins.filter_chr_names()
    side = _ChromosomeRows()
    side.order = [str(c) for c in ins.g_chrs_ids.tolist()]
    side.pos = np.asarray(ins.pos).astype(int)
    side.rows = {}
    if len(side.order):
        names, inv = np.unique(ins.g_chrs, return_inverse=True)
        by_name = np.argsort(inv, kind="stable")                      # rows grouped by chromosome, input order kept
        ends = np.cumsum(np.bincount(inv, minlength=len(names)))
        for k, name in enumerate(names.tolist()):
            side.rows[str(name)] = by_name[(ends[k - 1] if k else 0):ends[k]]
    return side
