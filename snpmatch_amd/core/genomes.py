"""
Genome description and window segmentation for ``cross`` (reference: core/genomes.py:16-127).

``Genome(ref_json)`` loads chromosome names and lengths (a genome id shipped under
``snpmatch_amd/resources/genomes`` or a path to a JSON with ``ref_chrs`` / ``ref_chrlen``);
``get_bins_genome`` / ``get_bins_arrays`` yield, chromosome after chromosome in genome order, one
``(chr_ix, [start, end], [indices])`` triple per window ``[1 + k*binLen, (k+1)*binLen]``
(core/genomes.py:111-116), for the DB positions and for the sample positions respectively.

Implementation: positions of one chromosome are sorted in every real input, so a window's indices are
one ``searchsorted`` range; unsorted or non-positive positions take the literal position-by-position
walk the reference uses (core/genomes.py:117-125) so that even its quirks are reproduced.
"""
import json
import logging
import os.path
from glob import glob

import numpy as np

log = logging.getLogger(__name__)

_RES = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'resources', 'genomes')


def _norm_ids(chrs):
    return np.char.replace(np.char.lower(np.array(chrs, dtype="str")), "chr", "")


class Genome(object):

    def __init__(self, ref_json):
        if ref_json in self.get_genome_ids():
            ref_json = os.path.join(_RES, ref_json + '.json')
        assert os.path.exists(ref_json), "Reference json file missing: %s" % ref_json
        with open(ref_json) as ref_genome:
            self.json = json.load(ref_genome)
        self.chrs = np.array(self.json['ref_chrs'], dtype="str")
        self.chrlen = np.array(self.json['ref_chrlen'], dtype=int)
        self.chrs_ids = _norm_ids(self.chrs)

    def get_genome_ids(self):
        return [os.path.basename(ef).replace(".json", "") for ef in glob(os.path.join(_RES, '*.json'))]

    def get_chr_ind(self, echr):
        real_chrs = np.array([ec.replace("Chr", "").replace("chr", "") for ec in self.chrs])
        if isinstance(echr, (str, bytes, np.str_, np.bytes_)):
            if isinstance(echr, (bytes, np.bytes_)):
                echr = echr.decode()
            echr_num = str(echr).replace("Chr", "").replace("chr", "")
            hit = np.where(real_chrs == echr_num)[0]
            return hit[0] if len(hit) == 1 else None
        echr = np.array(echr)
        ret = np.zeros(len(echr), dtype="int8")
        for ec in np.unique(echr):
            t_ix = np.where(real_chrs == str(ec).replace("Chr", "").replace("chr", ""))[0]
            ret[np.where(echr == ec)[0]] = t_ix[0]
        return ret

    def get_bins_genome(self, g, binLen):
        """windows over the DB positions (``g`` exposes chrs, chr_regions, positions); core/genomes.py:73-91."""
        binLen = int(binLen)
        g_chrs_ids = _norm_ids(g.chrs)
        common_chr_ids = np.intersect1d(g_chrs_ids, self.chrs_ids)
        assert len(g_chrs_ids) <= len(self.chrs_ids), "Please change default --genome option"
        assert len(common_chr_ids) > 0, "Please change default --genome option"
        if len(common_chr_ids) < len(self.chrs_ids):
            log.warning("Some reference contigs are missing in genotype hdf5 file")
        positions = np.asarray(g.positions)
        start = 0
        for chr_ix in range(len(self.chrs_ids)):
            t_g_ix = np.where(g_chrs_ids == self.chrs_ids[chr_ix])[0]
            if len(t_g_ix) == 0:
                chr_pos = np.zeros(0, dtype=int)
            else:
                start = int(g.chr_regions[t_g_ix[0]][0])
                end = int(g.chr_regions[t_g_ix[0]][1])
                chr_pos = positions[start:end]
            for e_bin in get_bins_echr(self.chrlen[chr_ix], chr_pos, binLen, start):
                yield (chr_ix, e_bin[0], e_bin[1])

    def get_bins_arrays(self, g_chrs, g_snppos, binLen):
        """windows over the sample positions; core/genomes.py:93-108."""
        g_chrs = _norm_ids(g_chrs)
        g_snppos = np.asarray(g_snppos)
        g_chrs_ids = np.unique(g_chrs)
        common_chr_ids = np.intersect1d(g_chrs_ids, self.chrs_ids)
        assert len(g_chrs_ids) <= len(self.chrs_ids), "Please change default --genome option"
        assert len(common_chr_ids) > 0, "Please change default --genome option"
        if len(common_chr_ids) < len(self.chrs_ids):
            log.warning("Some reference contigs are missing in given SNPs")
        for chr_ix in range(len(self.chrs_ids)):
            chr_pos_ix = np.where(g_chrs == self.chrs_ids[chr_ix])[0]
            rel = int(chr_pos_ix[0]) if len(chr_pos_ix) > 0 else 0
            for e_bin in get_bins_echr(self.chrlen[chr_ix], g_snppos[chr_pos_ix], binLen, rel):
                yield (chr_ix, e_bin[0], e_bin[1])

    def window_table(self, binLen):
        """(chr_ix, start, end) of every window of the genome, in iteration order."""
        out = []
        for chr_ix in range(len(self.chrs_ids)):
            for t in range(1, int(self.chrlen[chr_ix]), int(binLen)):
                out.append((chr_ix, t, t + int(binLen) - 1))
        return out


def _walk_bins(real_chrlen, chr_pos, binLen, rel_ix):
    """position-by-position walk with the reference's exact control flow (core/genomes.py:111-127)."""
    ind = 0
    npos = len(chr_pos)
    for t in range(1, int(real_chrlen), int(binLen)):
        lo, hi = int(t), int(t) + int(binLen) - 1
        result, closed = [], False
        k = ind
        while k < npos:
            epos = chr_pos[k]
            k += 1
            if epos >= lo:
                if epos <= hi:
                    result.append(ind + rel_ix)
                else:
                    closed = True
                    break
                ind += 1
        yield ([lo, hi], result)
        if closed:
            continue


def get_bins_echr(real_chrlen, chr_pos, binLen, rel_ix):
    chr_pos = np.asarray(chr_pos)
    n = len(chr_pos)
    binLen = int(binLen)
    if n > 0 and (chr_pos[0] < 1 or np.any(chr_pos[1:] < chr_pos[:-1])):
        for b in _walk_bins(real_chrlen, chr_pos, binLen, rel_ix):
            yield b
        return
    starts = np.arange(1, int(real_chrlen), binLen, dtype=np.int64)
    lo = np.searchsorted(chr_pos, starts, side="left")
    hi = np.searchsorted(chr_pos, starts + binLen - 1, side="right")
    for k in range(len(starts)):
        yield ([int(starts[k]), int(starts[k]) + binLen - 1], list(range(int(lo[k]) + rel_ix, int(hi[k]) + rel_ix)))
