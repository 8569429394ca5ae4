"""
Sample input parsing: the boundary producer of the hot path.

Interface of the reference's ``snpmatch.core.parsers`` (core/parsers.py:12-218): ``parseGT``,
``snp_binary_to_gt``, ``ParseInputs`` (attributes ``chrs, pos, gt, wei, dp``; ``filter_chr_names`` ->
``g_chrs, g_chrs_ids``; the ``<input>.snpmatch.npz`` cache with keys ``chr, pos, gt, wei, dp`` and the
``<input>.snpmatch.stats.json`` summary), ``import_vcf_file``, ``potatoParser``.

VCF text is read by ``_vcf.read_calls`` (no scikit-allel); weights are ``exp(-PL/10)`` where PL is
present and one-hot on the called genotype otherwise (core/parsers.py:132-157).

Deliberate fix: BED inputs carry ``dp = "NA"``, on which the reference crashes (``np.nanmean("NA")``,
core/parsers.py:113 and core/snpmatch.py:133); here an unknown depth is reported as NaN.
"""
import json
import logging
import os
import re
import sys

import numpy as np
from ._report import pd          # pandas, imported at first use

from . import _report
from . import _vcf

log = logging.getLogger(__name__)

_CODE_TO_GT = {-1: "./.", 0: "0/0", 1: "1/1", 2: "0/1"}


def die(msg):
    sys.stderr.write('Error: ' + msg + '\n')
    sys.exit(1)


def _gt_separator(head):
    """'|' or '/' as the genotype text ``head`` uses it, None for anything else"""
    head = str(head)
    return "|" if "|" in head else ("/" if "/" in head else None)


def parseGT(snpGT, _sep=None):
    """genotype text -> int8 codes: 0 hom-ref, 1 hom-alt, 2 het, -1 no call (anything else stays 0).
    The separator ('/' or '|') is taken from the first entry; purely numeric input is returned as codes."""
    snpGT = np.asarray(snpGT)
    codes = np.zeros(len(snpGT), dtype="int8")
    if len(codes) == 0:
        return codes
    sep = _sep or _gt_separator(snpGT[0])
    if sep is None:
        if str(snpGT[0]).isdigit():
            return np.array(np.copy(snpGT), dtype="int8")
        die("unable to parse the format of GT in vcf!")
    text = snpGT.astype("U")
    if text.dtype == np.dtype("<U3") and text.flags.c_contiguous:
        # three-character calls ('0/1', '1|1', './.'): compare the code points as integers instead of 200k strings four times
        cp = text.view(np.uint32).reshape(len(text), 3)
        a, mid, b = cp[:, 0], cp[:, 1], cp[:, 2]
        ok = mid == ord(sep)
        one_a, one_b, zero_a, zero_b = a == ord("1"), b == ord("1"), a == ord("0"), b == ord("0")
        codes[ok & one_a & one_b] = 1
        codes[ok & ((zero_a & one_b) | (one_a & zero_b))] = 2
        codes[ok & (a == ord(".")) & (b == ord("."))] = -1
        return codes
    for pattern, code in ((("1", "1"), 1), (("0", "1"), 2), (("1", "0"), 2), ((".", "."), -1)):
        codes[text == sep.join(pattern)] = code
    return codes


def snp_binary_to_gt(snpBinary):
    codes = np.array(snpBinary, dtype="int8")
    out = np.zeros(len(codes), dtype="S8")
    for code, text in _CODE_TO_GT.items():
        out[codes == code] = text
    return out


def _one_hot_weights(gt_text):
    """[n,3] weights (ref, het, alt): 1 in the column of the called genotype, 0 elsewhere"""
    codes = parseGT(gt_text)
    wei = np.zeros((len(codes), 3))
    for code, column in ((0, 0), (2, 1), (1, 2)):
        wei[codes == code, column] = 1.0
    return wei


def _weight_codes(pl, no_pl, wei):
    """Dictionary codes of a VCF sample's weights: (codes uint16 [n, 3], table float64) with ``table[codes] == wei`` bit for bit,
    or None.  The weights are exp(-PL / 10) of integer PLs (rows without PL: one-hot 0 / 1), so a few hundred table entries
    describe them all; the one-call scoring path then sends 6 instead of 24 bytes of weights per SNP to the GPU
    (``snpm_genotype_once_coded``).  The table is made with the same numpy expression as the weights and the identity is
    verified here, once, at parse time."""
    if pl.size == 0 or pl.min() < -1 or pl.max() > 60000 or not np.array_equal(pl, np.floor(pl)):
        return None
    k = int(pl.max()) + 1
    table = np.concatenate([np.exp(np.arange(k, dtype=float) / (-10)), [0.0], np.exp(np.array([-1.0]) / (-10))])
    codes = np.where(pl < 0, k + 1, pl).astype(np.uint16)          # a PL of -1 inside a partly present triple: exp(0.1), as the reference computes it
    codes[no_pl] = np.where(wei[no_pl] == 1.0, 0, k).astype(np.uint16)     # one-hot rows: 1.0 = exp(0) = table[0], 0.0 = table[k]
    if not np.array_equal(table[codes].view(np.uint64), np.ascontiguousarray(wei).view(np.uint64)):
        return None
    return codes, table


def _hard_call_codes(wei):
    """(codes, table) of a sample whose weights are all 0.0 or 1.0 (BED input, a VCF without PL, :118-127), else None: the same
    compact form for ``snpm_genotype_once_coded`` -- one pass over the weights"""
    wei = np.asarray(wei)
    if wei.ndim != 2 or wei.shape[1] != 3 or wei.size == 0 or wei.dtype != np.float64:
        return None
    wei = np.ascontiguousarray(wei)
    ones = wei == 1.0
    if not np.all(ones | (wei.view(np.uint64) == 0)):          # +0.0 only: the table holds its bits
        return None
    return ones.astype(np.uint16), np.array([0.0, 1.0])


class _StoredNpz(object):
    """The arrays of an ``.npz`` whose members are stored uncompressed (what ``np.savez`` writes: the parse cache) as views of ONE
    memory map -- ``np.load`` goes through ``zipfile``, which copies every member through Python and checksums it (0.15 s for the
    74-MB cache of a 1M-record sample, as long as parsing the VCF itself takes).  ``open`` returns None for anything it is not
    sure about (compressed members, object arrays, Fortran order, zip64 oddities): the caller then uses ``np.load``."""

    def __init__(self, arrays):
        self._arrays = arrays
        self.files = list(arrays)

    def __getitem__(self, key):
        return self._arrays[key]

    @classmethod
    def open(cls, path):
        import mmap
        import struct
        import zipfile
        try:
            with zipfile.ZipFile(path) as zf:
                infos = zf.infolist()
            if not infos or any(i.compress_type != zipfile.ZIP_STORED or not i.filename.endswith(".npy") or i.flag_bits & 0x1 for i in infos):
                return None
            with open(path, "rb") as fh:
                mm = mmap.mmap(fh.fileno(), 0, access=mmap.ACCESS_READ)
            arrays = {}
            for i in infos:
                o = i.header_offset
                if mm[o:o + 4] != b"PK\x03\x04":
                    return None
                n_name, n_extra = struct.unpack("<HH", mm[o + 26:o + 30])
                start = o + 30 + n_name + n_extra                          # the member's bytes: an .npy file
                if mm[start:start + 6] != b"\x93NUMPY":
                    return None
                major = mm[start + 6]
                if major == 1:
                    hlen = struct.unpack("<H", mm[start + 8:start + 10])[0]
                    hstart = start + 10
                elif major in (2, 3):
                    hlen = struct.unpack("<I", mm[start + 8:start + 12])[0]
                    hstart = start + 12
                else:
                    return None
                import ast
                head = ast.literal_eval(mm[hstart:hstart + hlen].decode("latin1" if major < 3 else "utf8"))
                dtype = np.dtype(head["descr"])
                if head.get("fortran_order") or dtype.hasobject:
                    return None
                shape = tuple(head["shape"])
                count = int(np.prod(shape, dtype=np.int64)) if shape else 1
                data = hstart + hlen
                if data + count * dtype.itemsize > start + i.file_size or i.file_size != i.compress_size:
                    return None
                arrays[i.filename[:-4]] = np.frombuffer(mm, dtype=dtype, count=count, offset=data).reshape(shape)
            return cls(arrays)
        except Exception:                                                   # noqa: BLE001 -- any surprise: the ordinary reader
            return None


class ParseInputs(object):

    def __init__(self, inFile, logDebug=True, outFile="parser"):
        prefix = inFile + ".snpmatch" if (outFile == "parser" or not outFile) else outFile
        cache = inFile + ".snpmatch.npz"
        if os.path.isfile(cache):
            log.info("using cached parse %s", cache)
            self._load_npz(cache)
        elif os.path.isfile(inFile):
            base = os.path.basename(inFile)
            if base.endswith(".npz"):
                log.info("reading parsed sample %s", inFile)
                self._load_npz(inFile)
                return
            log.info('parsing %s', inFile)
            if base.endswith(".vcf") or base.endswith(".vcf.gz"):
                fields = self.read_vcf(inFile, logDebug)
            elif base.endswith(".bed"):
                fields = self.read_bed(inFile, logDebug)
            else:
                die("input file type %s not supported" % os.path.splitext(inFile)[1])
            self.load_snp_info(*fields)
            self._adopt_codes(getattr(self, "_pending_codes", None) or _hard_call_codes(self.wei))
            self.save_snp_info(prefix, background=True)
            self.case_interpret_inputs(prefix + ".stats.json")
            log.info("parsed %d SNP calls", len(self.chrs))
        # anything else (e.g. ParseInputs("")) leaves an empty object to be filled with load_snp_info

    def _load_npz(self, path):
        z = _StoredNpz.open(path) or np.load(path)
        # the small members are copied out of the memory map (a cache file that another writer rewrites IN PLACE -- the reference's
        # own writer does -- would change them under a running job); the wide ones (chr / gt text, weights) stay views: read-only
        self.load_snp_info(z['chr'], np.array(z['pos']), z['gt'], z['wei'], np.array(z['dp']))
        if 'wei_codes' in z.files and 'wei_table' in z.files:       # written by this package's parser; verified, not trusted
            codes, table = z['wei_codes'], z['wei_table']
            if codes.shape == self.wei.shape and codes.dtype == np.uint16 and len(table) and int(codes.max(initial=0)) < len(table) and \
                    np.array_equal(table[codes].view(np.uint64), np.ascontiguousarray(self.wei).view(np.uint64)):
                self._adopt_codes((codes, table))
        else:                                                       # a cache the reference wrote: hard calls are coded on the spot
            self._adopt_codes(_hard_call_codes(np.ascontiguousarray(self.wei)))

    def _adopt_codes(self, pair):
        """dictionary codes of ``self.wei`` (``_weight_codes``): used by ``Genotyper``'s one-call path as long as ``self.wei`` still
        is the array they describe -- which is made read-only, so that an in-place edit cannot leave them stale (assign a new array
        to ``wei`` to change weights: the codes are dropped)."""
        self._wei_codes = self._wei_table = self._wei_coded_for = None
        if pair is None:
            return
        self._wei_codes, self._wei_table = np.ascontiguousarray(pair[0], dtype=np.uint16), np.ascontiguousarray(pair[1], dtype=np.float64)
        self.wei.flags.writeable = False
        self._wei_coded_for = self.wei

    def gt_codes_of(self, rows):
        """``parseGT(self.gt[rows])`` without gathering or re-parsing 200k strings per call: the whole column is parsed once per
        separator (``parseGT`` reads the separator from the FIRST entry it is given, here ``self.gt[rows[0]]``) and kept while
        ``self.gt`` is the same array.  None when that first entry has no separator (the caller parses the text itself)."""
        rows = np.asarray(rows)
        if len(rows) == 0 or len(self.gt) == 0:
            return np.zeros(0, dtype="int8")
        sep = _gt_separator(self.gt[int(rows[0])])
        if sep is None:
            return None
        cache = self.__dict__.setdefault("_gt_code_cache", {})
        hit = cache.get(sep)
        if hit is None or hit[0] is not self.gt:
            hit = cache[sep] = (self.gt, parseGT(self.gt, _sep=sep))
        return hit[1][rows]

    def weight_codes(self):
        """(codes, table) when they still describe ``self.wei``, else None"""
        if getattr(self, "_wei_codes", None) is not None and self._wei_coded_for is self.wei and not self.wei.flags.writeable:
            return self._wei_codes, self._wei_table
        return None

    def case_interpret_inputs(self, outFile):
        """``<prefix>.stats.json``: SNPs per chromosome, depth, heterozygosity, low-SNP warning"""
        from . import snpmatch
        n = len(self.chrs)
        few = n < snpmatch.snp_thres
        # SNPs per chromosome name from the runs of equal names (a sorted input has one run per chromosome) instead of a
        # sort of every name; the keys come out in np.unique's (sorted) order as before
        c = self.chrs
        starts = np.concatenate([[0], np.flatnonzero(c[1:] != c[:-1]) + 1]) if n > 1 else np.zeros(min(n, 1), dtype=np.int64)
        lens = np.diff(np.concatenate([starts, [n]])) if n else np.zeros(0, dtype=np.int64)
        names, inv = np.unique(c[starts], return_inverse=True) if n else (np.zeros(0, dtype="U1"), np.zeros(0, dtype=np.int64))
        counts = np.bincount(inv, weights=lens, minlength=len(names)).astype(np.int64) if n else np.zeros(0, dtype=np.int64)
        stats = {
            "snps": dict((str(k), int(v)) for k, v in zip(names, counts)),
            "interpretation": {"case": int(few),
                               "text": "Attention: low number of SNPs provided" if few else "Sufficient number of SNPs"},
            "num_of_snps": n,
            "depth": _report.mean_depth(self.dp),
            "percent_heterozygosity": snpmatch.getHeterozygosity(self.gt),
        }
        with open(outFile, "w") as fh:
            fh.write(json.dumps(stats))

    @staticmethod
    def get_wei_from_GT(snpGT):
        return _one_hot_weights(snpGT)

    @staticmethod
    def read_bed(inFile, logDebug):
        """three columns: chromosome, position, genotype (any whitespace / comma separator)"""
        log.info("reading BED-like table %s", inFile)
        table = pd.read_csv(inFile, header=None, sep=None, engine='python', usecols=[0, 1, 2])
        gt = np.array(table[2])
        return (np.array(table[0], dtype="str"), np.array(table[1], dtype=int), gt, _one_hot_weights(gt), "NA")

    def read_vcf(self, inFile, logDebug):
        """called sites of the first sample: (chr, pos, gt, weights, depth)"""
        calls = import_vcf_file(inFile, logDebug, samples_to_load=[0])
        gt = calls['gt'][:, 0]
        if calls.get('called') is not None:             # marked by the native reader while it copied the genotypes out
            called = np.flatnonzero(calls['called'])
        else:
            called = np.flatnonzero((gt != './.') & (gt != '.|.'))
        gt = gt[called]
        self._pending_codes = None
        if 'wei' in calls:
            pl = calls['wei'][called, 0]
            no_pl = np.all(pl == -1, axis=1)
            wei = np.exp(pl / (-10))
            wei[no_pl] = _one_hot_weights(gt[no_pl])
            self._pending_codes = _weight_codes(pl, no_pl, wei)
        else:
            wei = _one_hot_weights(gt)
        return (calls['chr'][called], calls['pos'][called], gt, wei, calls['dp'][called])

    def filter_chr_names(self):
        """``g_chrs``: chromosome names without a (case-insensitive) 'chr'; ``g_chrs_ids``: their distinct
        values in order of first appearance; ``g_chr_codes``: the position of every SNP's chromosome in ``g_chrs_ids``
        (integers: what the position intersection compares).  A sorted input names each chromosome in one run, so the
        distinct names are taken from the run heads and the regex runs on those only."""
        if len(self.chrs) == 0:
            self.g_chrs = self.g_chrs_ids = np.zeros(0, dtype="U1")
            self.g_chr_codes = np.zeros(0, dtype=np.int64)
            return
        c = self.chrs
        starts = np.concatenate([[0], np.flatnonzero(c[1:] != c[:-1]) + 1]) if len(c) > 1 else np.zeros(1, dtype=np.int64)
        lens = np.diff(np.concatenate([starts, [len(c)]]))
        uniq, first_h, inv_h = np.unique(c[starts], return_index=True, return_inverse=True)
        bare = np.array([re.sub("chr", "", x, flags=re.IGNORECASE) for x in uniq.tolist()], dtype="str")
        ids, code_of = [], np.zeros(len(uniq), dtype=np.int64)
        for k in np.argsort(starts[first_h], kind="stable"):        # 'Chr1' and 'chr1' are the same chromosome
            if bare[k] not in ids:
                ids.append(bare[k])
            code_of[k] = ids.index(bare[k])
        self.g_chrs = np.repeat(bare[inv_h], lens)
        self.g_chrs_ids = np.array(ids, dtype=self.g_chrs.dtype)
        self.g_chr_codes = np.repeat(code_of[inv_h], lens)

    def load_snp_info(self, snpCHR, snpPOS, snpGT, snpWEI, DPmean):
        self.chrs = np.array(snpCHR, dtype="str")
        self.pos = np.array(snpPOS, dtype=int)
        self.gt = np.array(snpGT, dtype="str")
        self.wei = np.array(snpWEI, dtype=float)
        self.dp = DPmean

    def save_snp_info(self, outFile, background=False):
        """the parse as ``<outFile>.npz`` (the cache the next run of the same input loads).  ``background``: written by a
        thread of its own, off the critical path (0.4 s of a 1M-record sample); the interpreter waits for it at exit,
        ``wait_for_cache`` earlier (an accession-sharded job's other ranks load the file)."""
        log.info("caching the parse as %s.npz", outFile)
        arrays = dict(chr=self.chrs, pos=self.pos, gt=self.gt, wei=self.wei, dp=self.dp)
        if self.weight_codes() is not None:             # extra keys: the reference's loader reads its five by name
            arrays.update(wei_codes=self._wei_codes, wei_table=self._wei_table)
        def write():
            # always a new inode: a reader that mapped the previous cache (_StoredNpz hands out views of a read-only map) keeps
            # its bytes -- a file rewritten in place would be truncated under the mapping (SIGBUS, or arrays that change)
            tmp = outFile + ".tmp%d" % os.getpid()
            try:
                np.savez(tmp, **arrays)                 # numpy appends .npz
                os.replace(tmp + ".npz", outFile + ".npz")      # readers never see a half-written cache
            except OSError as e:          # a read-only input directory costs the cache, not the run
                log.warning("could not cache the parse: %s", e)
        if not background:
            write()
            return
        import threading
        self._cache_writer = threading.Thread(target=write, name="snpmatch-parse-cache")
        self._cache_writer.start()

    def wait_for_cache(self):
        t = getattr(self, "_cache_writer", None)
        if t is not None:
            t.join()
            self._cache_writer = None

    def save_to_bed(self, outFile):
        pd.DataFrame({"chr": self.chrs, "pos": self.pos, "gt": self.gt}).to_csv(outFile, sep="\t", index=None, header=False)


def import_vcf_file(inFile, logDebug=False, samples_to_load=[0], add_fields=None):
    """dict with 'samples', 'gt' [n, s], 'wei' (the PL triples, -1 = missing; only when PL occurs), 'chr',
    'pos', 'dp' (INFO/DP, or "NA" entries when the file has none) -- the keys the reference builds from
    scikit-allel's output"""
    calls = _vcf.read_calls(inFile, tuple(samples_to_load))
    if not calls["has_gt"]:
        die("input VCF file doesnt have required GT field")
    out = {'samples': calls["samples"], 'gt': calls["gt"], 'chr': calls["chr"], 'pos': calls["pos"], 'called': calls.get("called")}
    if calls["pl"] is not None:
        out['wei'] = calls["pl"]
    out['dp'] = calls["dp"] if calls["dp"] is not None else np.repeat("NA", len(calls["pos"]))
    for name in (add_fields or ()):
        log.warning("Field %s is not loaded by this reader" % name)
    return out


def potatoParser(inFile, logDebug, outFile="parser"):
    parsed = ParseInputs(inFile, logDebug, outFile)
    return (parsed.chrs, parsed.pos, parsed.gt, parsed.wei, parsed.dp)
