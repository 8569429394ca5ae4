"""
Genome description and window segmentation for ``cross``.

``Genome(ref_json)`` (interface of the reference's core/genomes.py:16-127) loads chromosome names and
lengths from a genome id shipped under ``snpmatch_amd/resources/genomes`` or from a JSON file with
``ref_chrs`` / ``ref_chrlen``.  ``get_bins_genome(g, binLen)`` and ``get_bins_arrays(chrs, pos, binLen)``
yield, chromosome after chromosome in genome order, one ``(chr_ix, [start, end], [indices])`` triple per
window ``[1 + k*binLen, (k+1)*binLen]`` -- indices into the DB rows and into the sample respectively.

Positions of one chromosome are sorted in every real input, so the members of all windows of a
chromosome come from two ``searchsorted`` calls.  Inputs that are not sorted (or hold positions < 1) go
through ``_scan_windows``, which walks position by position with the control flow of the reference
(core/genomes.py:111-127) so that its behaviour on such inputs is reproduced as well.
"""
import json
import logging
import os.path
from glob import glob

import numpy as np

log = logging.getLogger(__name__)

_GENOME_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'resources', 'genomes')
_MISMATCH = "Please change default --genome option"


def _bare(names):
    """lower-case chromosome names without 'chr'"""
    return np.char.replace(np.char.lower(np.array(names, dtype="str")), "chr", "")


def _scan_windows(chr_len, positions, bin_len, first_index):
    cursor = 0
    for lo in range(1, int(chr_len), bin_len):
        hi = lo + bin_len - 1
        members = []
        k = cursor
        while k < len(positions):
            p = positions[k]
            k += 1
            if p < lo:
                continue            # (the reference does not advance its index here either)
            if p > hi:
                break
            members.append(cursor + first_index)
            cursor += 1
        yield ([lo, hi], members)


def get_bins_echr(real_chrlen, chr_pos, binLen, rel_ix):
    """windows of one chromosome: ([start, end], [rel_ix + index of every position inside])"""
    chr_pos = np.asarray(chr_pos)
    binLen = int(binLen)
    unsorted = len(chr_pos) > 0 and (chr_pos[0] < 1 or bool(np.any(np.diff(chr_pos) < 0)))
    if unsorted:
        for w in _scan_windows(real_chrlen, chr_pos, binLen, rel_ix):
            yield w
        return
    starts = np.arange(1, int(real_chrlen), binLen, dtype=np.int64)
    first = np.searchsorted(chr_pos, starts, side="left") + rel_ix
    last = np.searchsorted(chr_pos, starts + (binLen - 1), side="right") + rel_ix
    for s, a, b in zip(starts.tolist(), first.tolist(), last.tolist()):
        yield ([s, s + binLen - 1], list(range(a, b)))


class Genome(object):

    def __init__(self, ref_json):
        shipped = os.path.join(_GENOME_DIR, str(ref_json) + '.json')
        path = shipped if os.path.exists(shipped) else ref_json
        assert os.path.exists(path), "Reference json file missing: %s" % path
        with open(path) as fh:
            self.json = json.load(fh)
        self.chrs = np.array(self.json['ref_chrs'], dtype="str")
        self.chrlen = np.array(self.json['ref_chrlen'], dtype=int)
        self.chrs_ids = _bare(self.chrs)

    def get_genome_ids(self):
        return sorted(os.path.basename(f)[:-len(".json")] for f in glob(os.path.join(_GENOME_DIR, '*.json')))

    def get_chr_ind(self, echr):
        """index of a chromosome name (or of every name in an array) in this genome; None if unknown"""
        def lookup(name):
            hit = np.flatnonzero(self.chrs_ids == _bare([name])[0])
            return int(hit[0]) if len(hit) == 1 else None
        if isinstance(echr, (bytes, np.bytes_)):
            echr = echr.decode()
        if isinstance(echr, str):
            return lookup(echr)
        return np.array([lookup(str(e)) for e in np.asarray(echr)], dtype="int8")

    def _check(self, ids, what):
        assert len(ids) <= len(self.chrs_ids), _MISMATCH
        shared = np.intersect1d(ids, self.chrs_ids)
        assert len(shared) > 0, _MISMATCH
        if len(shared) < len(self.chrs_ids):
            log.warning("Some reference contigs are missing in " + what)

    def get_bins_genome(self, g, binLen):
        """windows over the DB: ``g`` exposes ``chrs``, ``chr_regions`` (row range per chromosome), ``positions``"""
        db_ids = _bare(g.chrs)
        self._check(db_ids, "genotype hdf5 file")
        positions = np.asarray(g.positions)
        row0 = 0
        for chr_ix, cid in enumerate(self.chrs_ids):
            where = np.flatnonzero(db_ids == cid)
            if len(where):
                row0, row1 = int(g.chr_regions[where[0]][0]), int(g.chr_regions[where[0]][1])
                here = positions[row0:row1]
            else:
                here = positions[:0]            # chromosome absent from the DB: windows without members
            for span, members in get_bins_echr(self.chrlen[chr_ix], here, binLen, row0):
                yield (chr_ix, span, members)

    def get_bins_arrays(self, g_chrs, g_snppos, binLen):
        """windows over a sample given as parallel chromosome / position arrays"""
        ids = _bare(g_chrs)
        self._check(np.unique(ids), "given SNPs")
        g_snppos = np.asarray(g_snppos)
        for chr_ix, cid in enumerate(self.chrs_ids):
            mine = np.flatnonzero(ids == cid)
            offset = int(mine[0]) if len(mine) else 0
            for span, members in get_bins_echr(self.chrlen[chr_ix], g_snppos[mine], binLen, offset):
                yield (chr_ix, span, members)

    def window_table(self, binLen):
        """(chr_ix, start, end) of every window of the genome, in iteration order"""
        return [(c, s, s + int(binLen) - 1) for c in range(len(self.chrs_ids))
                for s in range(1, int(self.chrlen[c]), int(binLen))]
