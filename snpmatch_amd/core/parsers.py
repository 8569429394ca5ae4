"""
Sample input parsing: the boundary producer of the hot path (chrs, pos, gt, wei, dp).

Mirrors the interface of the reference's ``snpmatch.core.parsers`` (core/parsers.py:12-218):
``parseGT``, ``snp_binary_to_gt``, ``ParseInputs`` (same attributes, same ``.npz`` cache keys
``chr, pos, gt, wei, dp``, same ``.stats.json``), ``import_vcf_file``, ``potatoParser``.

The reference reads VCF through scikit-allel, which is not a dependency here: ``import_vcf_file``
is a plain text reader that returns the same fields with the same conventions
(core/parsers.py:178-213): GT of the first sample as ``'0/0'``-style strings, PL as float with -1
for missing (three values, Number=G of a diploid call), INFO/DP as ``variants/DP`` (-1 when absent).
This is CPU work on a few thousand lines; the scoring itself never runs here.

Deliberate fix: BED inputs carry ``dp = "NA"``; the reference then crashes in ``np.nanmean("NA")``
(core/parsers.py:113, core/snpmatch.py:133).  Here "NA" depth is reported as NaN.
"""
import gzip
import json
import logging
import os
import re

import numpy as np
import pandas as pd

log = logging.getLogger(__name__)

snp_thres = 4000          # core/snpmatch.py:18 (kept here as well to avoid an import cycle)


def die(msg):
    import sys
    sys.stderr.write('Error: ' + msg + '\n')
    sys.exit(1)


def parseGT(snpGT):
    """GT strings -> int8 codes (0 ref, 1 alt, 2 het, -1 no call); core/parsers.py:12-35."""
    snpGT = np.asarray(snpGT)
    snpBinary = np.zeros(len(snpGT), dtype="int8")
    if len(snpBinary) == 0:
        return snpBinary
    first = str(snpGT[0])
    if first.find('|') != -1:
        sep = "|"
    elif first.find('/') != -1:
        sep = "/"
    elif first.isdigit():
        return np.array(np.copy(snpGT), dtype="int8")
    else:
        die("unable to parse the format of GT in vcf!")
    gt = snpGT.astype("U")
    snpBinary[gt == "1" + sep + "1"] = 1
    snpBinary[(gt == "0" + sep + "1") | (gt == "1" + sep + "0")] = 2
    snpBinary[gt == "." + sep + "."] = -1
    return snpBinary


def snp_binary_to_gt(snpBinary):
    """core/parsers.py:37-44."""
    snpBinary = np.array(snpBinary, dtype="int8")
    snpGT = np.zeros(len(snpBinary), dtype="S8")
    snpGT[snpBinary == -1] = "./."
    snpGT[snpBinary == 0] = "0/0"
    snpGT[snpBinary == 1] = "1/1"
    snpGT[snpBinary == 2] = "0/1"
    return snpGT


def _nanmean_depth(dp):
    """np.nanmean(dp) that tolerates the "NA" depth of BED inputs (see module docstring)."""
    try:
        arr = np.asarray(dp, dtype=float)
    except (TypeError, ValueError):
        return float("nan")
    if arr.size == 0:
        return float("nan")
    if np.all(np.isnan(arr)):
        return float("nan")
    return float(np.nanmean(arr))


class ParseInputs(object):
    """core/parsers.py:59-175."""

    def __init__(self, inFile, logDebug=True, outFile="parser"):
        if outFile == "parser" or not outFile:
            outFile = inFile + ".snpmatch"
        if os.path.isfile(inFile + ".snpmatch.npz"):
            log.info("snpmatch parser dump found! loading %s", inFile + ".snpmatch.npz")
            snps = np.load(inFile + ".snpmatch.npz")
            self.load_snp_info(snps['chr'], snps['pos'], snps['gt'], snps['wei'], snps['dp'])
            log.info("done!")
        elif os.path.isfile(inFile):
            _, inType = os.path.splitext(inFile)
            if inType == '.npz':
                log.info("loading snpmatch parser file! %s", inFile)
                snps = np.load(inFile)
                self.load_snp_info(snps['chr'], snps['pos'], snps['gt'], snps['wei'], snps['dp'])
            else:
                log.info('running snpmatch parser!')
                if inType == '.vcf' or os.path.basename(inFile).endswith(".vcf.gz"):
                    (snpCHR, snpPOS, snpGT, snpWEI, DPmean) = self.read_vcf(inFile, logDebug)
                elif inType == '.bed':
                    (snpCHR, snpPOS, snpGT, snpWEI, DPmean) = self.read_bed(inFile, logDebug)
                else:
                    die("input file type %s not supported" % inType)
                self.load_snp_info(snpCHR, snpPOS, snpGT, snpWEI, DPmean)
                self.save_snp_info(outFile)
                self.case_interpret_inputs(outFile + ".stats.json")
            log.info("done!")

    def load_snp_info(self, snpCHR, snpPOS, snpGT, snpWEI, DPmean):
        self.chrs = np.array(snpCHR, dtype="str")
        self.pos = np.array(snpPOS, dtype=int)
        self.gt = np.array(snpGT, dtype="str")
        self.wei = np.array(snpWEI, dtype=float)
        self.dp = DPmean

    def save_snp_info(self, outFile):
        log.info("creating snpmatch parser file: %s", outFile + '.npz')
        np.savez(outFile, chr=self.chrs, pos=self.pos, gt=self.gt, wei=self.wei, dp=self.dp)

    def case_interpret_inputs(self, outFile):
        from . import snpmatch
        NumSNPs = len(self.chrs)
        case, note = 0, "Sufficient number of SNPs"
        if NumSNPs < snpmatch.snp_thres:
            note, case = "Attention: low number of SNPs provided", 1
        ids, counts = np.unique(self.chrs, return_counts=True)
        statdict = {
            "snps": dict(('%s' % ids[i], int(counts[i])) for i in range(len(ids))),
            "interpretation": {"case": case, "text": note},
            "num_of_snps": NumSNPs,
            "depth": _nanmean_depth(self.dp),
            "percent_heterozygosity": snpmatch.getHeterozygosity(self.gt),
        }
        with open(outFile, "w") as out_stats:
            out_stats.write(json.dumps(statdict))

    @staticmethod
    def read_bed(inFile, logDebug):
        log.info("reading the position file")
        targetSNPs = pd.read_csv(inFile, header=None, sep=None, engine='python', usecols=[0, 1, 2])
        snpCHR = np.array(targetSNPs[0], dtype="str")
        snpPOS = np.array(targetSNPs[1], dtype=int)
        snpGT = np.array(targetSNPs[2])
        snpWEI = ParseInputs.get_wei_from_GT(snpGT)
        return (snpCHR, snpPOS, snpGT, snpWEI, "NA")

    @staticmethod
    def get_wei_from_GT(snpGT):
        """hard 0/1 weights from the called genotype (core/parsers.py:132-139)."""
        snpBinary = parseGT(snpGT)
        snpWEI = np.ones((len(snpGT), 3))
        snpWEI[snpBinary != 0, 0] = 0
        snpWEI[snpBinary != 1, 2] = 0
        snpWEI[snpBinary != 2, 1] = 0
        return snpWEI

    def read_vcf(self, inFile, logDebug):
        """core/parsers.py:141-157."""
        snp_inputs = import_vcf_file(inFile, logDebug, samples_to_load=[0])
        gt = snp_inputs['gt'][:, 0]
        snpsREQ = np.where((gt != './.') & (gt != '.|.'))[0]
        snpGT = gt[snpsREQ]
        if 'wei' in snp_inputs:
            snpWEI = snp_inputs['wei'][snpsREQ, 0]
            missing_pls = np.all(snpWEI == -1, axis=1)
            snpWEI = np.exp(snpWEI / (-10))
            if missing_pls.any():
                snpWEI[missing_pls, ] = self.get_wei_from_GT(snpGT[missing_pls])
        else:
            snpWEI = self.get_wei_from_GT(snpGT)
        snpCHR = snp_inputs['chr'][snpsREQ]
        snpPOS = snp_inputs['pos'][snpsREQ]
        snpDP = snp_inputs['dp'][snpsREQ]
        return (snpCHR, snpPOS, snpGT, snpWEI, snpDP)

    def filter_chr_names(self):
        """strip 'chr' case-insensitively; ids in order of first appearance (core/parsers.py:159-163).
        Chromosome names take few distinct values: the regex runs on the unique names only."""
        if len(self.chrs) == 0:
            self.g_chrs = np.zeros(0, dtype="U1")
            self.g_chrs_ids = self.g_chrs
            return
        uniq, first, inv = np.unique(self.chrs, return_index=True, return_inverse=True)
        stripped = np.array([re.sub("chr", "", c, flags=re.IGNORECASE) for c in uniq.tolist()], dtype="str")
        self.g_chrs = stripped[inv]
        # order of first appearance of the STRIPPED ids ('Chr1' and 'chr1' are the same chromosome)
        order = np.argsort(first, kind="stable")
        ids = []
        for k in order:
            if stripped[k] not in ids:
                ids.append(stripped[k])
        self.g_chrs_ids = np.array(ids, dtype=self.g_chrs.dtype)

    def save_to_bed(self, outFile):
        input_df = pd.DataFrame(np.column_stack((self.chrs, self.pos, self.gt)), columns=["chr", 'pos', 'gt'])
        input_df.to_csv(outFile, sep="\t", index=None, header=False)


def _open_text(path):
    if path.endswith(".gz"):
        return gzip.open(path, "rt")
    return open(path, "r")


def import_vcf_file(inFile, logDebug=False, samples_to_load=[0], add_fields=None):
    """Text VCF reader returning what the reference extracts from scikit-allel
    (core/parsers.py:178-213): dict with 'samples', 'gt' [n, s] str, 'wei' [n, s, 3] float (PL, -1 =
    missing; only when some record carries PL), 'chr', 'pos', 'dp' (INFO/DP, -1 = missing)."""
    chroms, poss, gts, pls, dps = [], [], [], [], []
    samples = []
    have_pl = False
    have_gt = False
    have_info_dp = False
    sel = list(samples_to_load)
    with _open_text(inFile) as fh:
        for line in fh:
            if line.startswith("##"):
                if line.startswith("##INFO=<ID=DP,"):
                    have_info_dp = True
                continue
            if line.startswith("#"):
                cols = line.rstrip("\n").split("\t")
                samples = cols[9:]
                continue
            f = line.rstrip("\n").split("\t")
            if len(f) < 8:
                continue
            chroms.append(f[0])
            poss.append(int(f[1]))
            dp = -1
            if f[7] != ".":
                for kv in f[7].split(";"):
                    if kv.startswith("DP="):
                        try:
                            dp = int(kv[3:])
                            have_info_dp = True
                        except ValueError:
                            dp = -1
                        break
            dps.append(dp)
            fmt = f[8].split(":") if len(f) > 8 else []
            row_gt, row_pl = [], []
            for si in sel:
                col = 9 + si
                val = f[col].split(":") if len(f) > col else []
                gt = "./."
                pl = (-1.0, -1.0, -1.0)
                for k, key in enumerate(fmt):
                    if k >= len(val):
                        break
                    if key == "GT":
                        have_gt = True
                        gt = val[k]
                        if gt == ".":
                            gt = "./."
                    elif key == "PL" and val[k] != ".":
                        parts = val[k].split(",")
                        vals = []
                        for x in parts[:3]:
                            vals.append(-1.0 if x == "." else float(x))
                        while len(vals) < 3:
                            vals.append(-1.0)
                        pl = tuple(vals)
                        have_pl = True
                row_gt.append(gt)
                row_pl.append(pl)
            gts.append(row_gt)
            pls.append(row_pl)
    if not have_gt and len(chroms) > 0:
        die("input VCF file doesnt have required GT field")
    snp_inputs = {}
    snp_inputs['samples'] = np.array([samples[i] for i in sel if i < len(samples)]).astype('U')
    snp_inputs['gt'] = np.array(gts, dtype='U').reshape(len(chroms), len(sel))
    if have_pl:
        snp_inputs['wei'] = np.array(pls, dtype=float).reshape(len(chroms), len(sel), 3)
    snp_inputs['chr'] = np.array(chroms, dtype="str").astype('U')
    snp_inputs['pos'] = np.array(poss, dtype=int)
    if have_info_dp:
        snp_inputs['dp'] = np.array(dps, dtype=int)
    else:
        snp_inputs['dp'] = np.repeat("NA", len(poss))
    if add_fields is not None:
        for ef in add_fields:
            log.warning("Field %s is not loaded by this reader" % ef)
    return snp_inputs


def potatoParser(inFile, logDebug, outFile="parser"):
    inputs = ParseInputs(inFile, logDebug, outFile)
    return (inputs.chrs, inputs.pos, inputs.gt, inputs.wei, inputs.dp)
