"""
Plain-text VCF reader for single-sample SNP calls (the reference delegates this to scikit-allel,
core/parsers.py:178-213; scikit-allel is not a dependency of this package).  A single sample column goes
through the library's C++ reader (``snpm_vcf_parse``, one pass over the file); the Python loop below is the
generic path (several samples, or files the native reader declines).

``read_calls`` returns, for the selected samples, what the scoring path consumes:
  chr  [n]        CHROM as text
  pos  [n]        POS
  gt   [n, s]     genotype text exactly as written ('0/0', '0|1', './.'; a bare '.' becomes './.')
  pl   [n, s, 3]  phred-scaled genotype likelihoods (ref/ref, ref/alt, alt/alt), -1 where absent;
                  None when no record carries PL
  dp   [n]        INFO/DP, -1 where absent; None when no record carries it
"""
import gzip

import numpy as np


def _open(path):
    return gzip.open(path, "rt") if path.endswith(".gz") else open(path, "r")


def _info_depth(info):
    if info == ".":
        return None
    for item in info.split(";"):
        if item.startswith("DP="):
            try:
                return int(item[3:])
            except ValueError:
                return None
    return None


def _sample_fields(keys, text):
    """(gt, pl) of one sample column given the FORMAT keys"""
    gt, pl = "./.", None
    for key, val in zip(keys, text.split(":")):
        if key == "GT":
            gt = "./." if val == "." else val
        elif key == "PL" and val != ".":
            nums = [(-1.0 if x == "." else float(x)) for x in val.split(",")[:3]]
            pl = tuple(nums + [-1.0] * (3 - len(nums)))
    return gt, pl


def _read_calls_native(path, sample):
    """one sample column through the library's single-pass reader; None when it declines the file"""
    from .. import _lib
    raw = _lib.vcf_parse(path, sample)
    if raw is None:
        return None
    n = len(raw["pos"])
    return {
        "samples": np.array(raw["names"][sample:sample + 1], dtype="U"),
        "has_gt": raw["has_gt"] or n == 0,
        "chr": raw["chr"],                         # unicode arrays filled by the library: no conversion pass
        "pos": raw["pos"],
        "gt": raw["gt"].reshape(n, 1),
        "pl": raw["pl"].reshape(n, 1, 3) if raw["has_pl"] else None,
        "dp": raw["dp"] if raw["has_dp"] else None,
        "called": raw["called"],                   # bool [n] (or None): genotype is not './.' / '.|.'
    }


def read_calls(path, samples=(0,), native=True):
    if native and len(samples) == 1:
        fast = _read_calls_native(path, int(samples[0]))
        if fast is not None:
            return fast
    names, chrom, pos, depth, gts, pls = [], [], [], [], [], []
    any_gt = any_pl = any_dp = False
    with _open(path) as fh:
        for line in fh:
            if line.startswith("#"):
                if line.startswith("#CHROM"):
                    names = line.rstrip("\n").split("\t")[9:]
                continue
            rec = line.rstrip("\n").split("\t")
            if len(rec) < 8:
                continue
            chrom.append(rec[0])
            pos.append(int(rec[1]))
            d = _info_depth(rec[7])
            any_dp |= d is not None
            depth.append(-1 if d is None else d)
            keys = rec[8].split(":") if len(rec) > 8 else []
            any_gt |= "GT" in keys
            row_gt, row_pl = [], []
            for s in samples:
                g, p = _sample_fields(keys, rec[9 + s]) if len(rec) > 9 + s else ("./.", None)
                any_pl |= p is not None
                row_gt.append(g)
                row_pl.append(p if p is not None else (-1.0, -1.0, -1.0))
            gts.append(row_gt)
            pls.append(row_pl)
    n, s = len(chrom), len(samples)
    return {
        "called": None,
        "samples": np.array([names[i] for i in samples if i < len(names)], dtype="U"),
        "has_gt": any_gt or n == 0,
        "chr": np.array(chrom, dtype="U"),
        "pos": np.array(pos, dtype=int),
        "gt": np.array(gts, dtype="U").reshape(n, s),
        "pl": np.array(pls, dtype=float).reshape(n, s, 3) if any_pl else None,
        "dp": np.array(depth, dtype=int) if any_dp else None,
    }
