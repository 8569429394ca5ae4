"""
Result tables and JSON summaries of the inbred / cross runs (host glue around the device results).

File formats are the reference's (core/snpmatch.py:122-168 for ``*.scores.txt`` / ``*.matches.json``,
core/csmatch.py:44-61 and :131-186 for the window table and the cross interpretation); the code here is
organised around numpy arrays rather than per-accession Python loops.
"""
import json

import numpy as np


class _LazyPandas(object):
    """``pd`` that imports pandas at its first use: an `inbred` run on a VCF never needs it (0.25-0.5 s of a 0.6-s cold run)"""

    def __getattr__(self, name):
        import pandas
        return getattr(pandas, name)


pd = _LazyPandas()

SCORE_COLUMNS = ("accs", "matches", "ninfo", "probabilities", "likelihood", "lrt", "num_snps", "dp")
WINDOW_COLUMNS = ("acc", "snps_match", "snps_info", "score", "likelihood", "identical", "num_amb", "window_index")

INBRED_CASES = {
    0: "Unique hit",
    1: "Ambiguous sample",
    2: "Ambiguous sample: Accessions in top hits can be really close",
    3: "Ambiguous sample: Sample might contain mixture of DNA or contamination",
    4: "Ambiguous sample: Many input SNP positions are missing in db positions. Maybe sample  not one in database",
}


def ratio_or_nan(num, den):
    """elementwise num / den with NaN wherever den <= 0 (the vector form of get_fraction)"""
    num = np.asarray(num, dtype=float)
    den = np.asarray(den, dtype=float)
    out = np.full(np.broadcast(num, den).shape, np.nan)
    np.divide(num, den, out=out, where=den > 0)
    return out


def mean_depth(dp):
    """mean sequencing depth of the sample; BED inputs carry the string "NA" -> NaN (a float is taken as the mean itself)"""
    if isinstance(dp, float):
        return dp
    try:
        arr = np.asarray(dp, dtype=float)
    except (TypeError, ValueError):
        return float("nan")
    if arr.size == 0 or np.all(np.isnan(arr)):
        return float("nan")
    return float(np.nanmean(arr))


# ----------------------------------------------------------------------------- inbred
def scores_frame(accs, matches, ninfo, fractions, likelihood, lrt, num_snps, dp):
    """one row per accession, columns in the order of ``*.scores.txt`` (written without header)"""
    data = dict(zip(SCORE_COLUMNS, (accs, matches, ninfo, fractions, likelihood, lrt, num_snps, mean_depth(dp))))
    return pd.DataFrame(data, columns=list(SCORE_COLUMNS))


def _csv_float(x):
    """a float as ``DataFrame.to_csv`` writes it: shortest round-trip text, empty for NaN"""
    x = float(x)
    return "" if x != x else repr(x)


def write_scores_table(path, accs, matches, ninfo, fractions, likelihood, lrt, num_snps, dp):
    """``*.scores.txt`` (core/snpmatch.py:122-138: tab-separated, no header, no index) written directly -- the bytes
    ``scores_frame(...).to_csv(path, header=None, sep="\t", index=None)`` would write (checked against pandas in
    tests/test_host_logic_cpu.py), at a tenth of its cost for a 1135-accession table.  Accession names that would need CSV
    quoting go through pandas."""
    accs = [str(a) for a in np.asarray(accs).tolist()]
    if any(("\t" in a) or ('"' in a) or ("\n" in a) or ("\r" in a) for a in accs):
        scores_frame(accs, matches, ninfo, fractions, likelihood, lrt, num_snps, dp).to_csv(path, header=None, sep="\t", index=None)
        return
    tail = "\t%s\t%s\n" % (_csv_value(num_snps), _csv_float(mean_depth(dp)))
    cols = [accs, _csv_column(matches), _csv_column(ninfo), _csv_column(np.asarray(fractions, dtype=float)),
            _csv_column(np.asarray(likelihood, dtype=float)), _csv_column(np.asarray(lrt, dtype=float))]
    with open(path, "w") as fh:                      # column-wise conversions, one join per row: 1 ms for 1135 accessions
        fh.write(tail.join(map("\t".join, zip(*cols))) + (tail if accs else ""))


def _csv_column(a):
    """a numeric column as the strings ``to_csv`` writes: integers as they are, floats as shortest round-trip text, NaN empty"""
    a = np.asarray(a)
    if a.dtype.kind in "iub":
        return list(map(str, a.tolist()))
    a = a.astype(float, copy=False)
    out = list(map(repr, a.tolist()))
    for i in np.flatnonzero(np.isnan(a)).tolist():
        out[i] = ""
    return out


def _csv_value(v):
    """an integer or float scalar as ``to_csv`` writes it"""
    if isinstance(v, (int, np.integer)):
        return str(int(v))
    return _csv_float(v)


def inbred_case(n_top, mean_top_fraction, overlap, prob_thres, overlap_thres=0.5):
    """interpretation code of an inbred run from its top hits (accessions with lrt below the threshold)"""
    if n_top == 1:
        case = 0
    elif mean_top_fraction > prob_thres:
        case = 2
    elif overlap > overlap_thres:
        case = 3
    elif overlap < overlap_thres:
        case = 4
    else:
        case = 1
    return case, INBRED_CASES[case]


def matches_summary(accs, fractions, ninfo, lrt, overlap, num_snps, lr_thres, prob_thres):
    """dict behind ``*.matches.json``: top hits ordered by decreasing match fraction"""
    top = np.flatnonzero(lrt < lr_thres)
    order = top[np.argsort(-fractions[top])]
    coverage = ratio_or_nan(ninfo, np.full(len(ninfo), num_snps))
    mean_top = np.nanmean(fractions[top]) if len(top) else np.nan
    case, text = inbred_case(len(top), mean_top, overlap, prob_thres)
    hits = [(str(accs[i]), float(fractions[i]), int(ninfo[i]), float(coverage[i])) for i in order]
    return {"overlap": [overlap, num_snps], "matches": hits, "interpretation": {"case": case, "text": text}}


class _NotPlain(Exception):
    pass


def _json_text(obj, default=None):
    """``json.dumps(obj, sort_keys=True, indent=4, default=default)`` for the plain structures the reports are made of (dicts
    with string keys, lists / tuples, strings, bools, ints, floats, None), text for text the same -- built with joins instead of
    the standard library's generator-per-token encoder, which Python runs without its C accelerator as soon as ``indent`` is
    set (17 ms of a 25-ms warm ``cross`` run went into three such dumps).  Anything else raises ``_NotPlain``."""
    quote = json.encoder.encode_basestring_ascii
    inf = float("inf")

    def scalar(o):
        if o is None:
            return "null"
        if o is True:
            return "true"
        if o is False:
            return "false"
        if isinstance(o, str):
            return quote(o)
        if isinstance(o, int):
            return int.__repr__(o)
        if isinstance(o, float):
            if o != o:
                return "NaN"
            if o == inf:
                return "Infinity"
            if o == -inf:
                return "-Infinity"
            return float.__repr__(o)
        return None

    def enc(o, pad):
        t = scalar(o)
        if t is not None:
            return t
        inner = pad + "    "
        if isinstance(o, (list, tuple)):
            if not o:
                return "[]"
            return "[\n" + inner + (",\n" + inner).join([enc(v, inner) for v in o]) + "\n" + pad + "]"
        if isinstance(o, dict):
            if not o:
                return "{}"
            for k in o:
                if not isinstance(k, str):
                    raise _NotPlain()
            return "{\n" + inner + (",\n" + inner).join([quote(k) + ": " + enc(o[k], inner) for k in sorted(o)]) + "\n" + pad + "}"
        if default is not None:
            return enc(default(o), pad)
        raise _NotPlain()

    return enc(obj, "")


def dump_json(obj, path, **kw):
    try:
        if set(kw) - {"default"}:
            raise _NotPlain()
        text = _json_text(obj, kw.get("default"))
    except (_NotPlain, RecursionError):
        text = json.dumps(obj, sort_keys=True, indent=4, **kw)
    with open(path, "w") as fh:
        fh.write(text)


def update_json(path, **fields):
    with open(path) as fh:
        obj = json.load(fh)
    obj.update(fields)
    dump_json(obj, path)


# ----------------------------------------------------------------------------- cross: window table
def float_text(values):
    """numpy's float -> text conversion (shortest round-trip repr): the reference stacks the numeric
    columns with the accession names, which turns them into strings"""
    return np.asarray(values, dtype=float).astype("U32")


def window_rows(window_index, accs, scores, ninfo, likelihood, lrt, identical, lr_thres):
    """rows of one window: the accessions whose likelihood ratio to the window's best is below the
    threshold, reported only when at least one but not all accessions qualify"""
    keep = np.flatnonzero(lrt < lr_thres)
    if len(keep) == 0 or len(keep) >= len(accs):
        return pd.DataFrame(columns=list(WINDOW_COLUMNS))
    sc = np.asarray(scores, dtype=float)[keep]
    ni = np.asarray(ninfo)[keep]
    frame = pd.DataFrame({
        "acc": np.asarray(accs)[keep].astype(str),
        "snps_match": sc.astype(int),
        "snps_info": ni.astype(float).astype(int),
        "score": float_text(ratio_or_nan(sc, ni)).astype(object),
        "likelihood": float_text(np.asarray(likelihood)[keep]).astype(object),
        "identical": np.asarray(identical, dtype=float)[keep],
        "num_amb": len(keep),
        "window_index": window_index,
    }, columns=list(WINDOW_COLUMNS))
    return frame


def window_table(window_index, accs, scores, ninfo, likelihood, lrt, identical, lr_thres):
    """``window_rows`` of many windows at once: 2-D inputs [n_windows, n_accessions], ``window_index`` [n_windows];
    rows in window order, accessions ascending inside a window (the concatenation of the per-window frames)."""
    lrt = np.asarray(lrt)
    keep = lrt < lr_thres
    n_amb = keep.sum(axis=1)
    keep &= ((n_amb > 0) & (n_amb < lrt.shape[1]))[:, None]
    w, a = np.nonzero(keep)
    if len(w) == 0:
        return pd.DataFrame(columns=list(WINDOW_COLUMNS))
    sc = np.asarray(scores, dtype=float)[w, a]
    ni = np.asarray(ninfo)[w, a]
    return pd.DataFrame({
        "acc": np.asarray(accs)[a].astype(str),
        "snps_match": sc.astype(int),
        "snps_info": ni.astype(float).astype(int),
        "score": float_text(ratio_or_nan(sc, ni)).astype(object),
        "likelihood": float_text(np.asarray(likelihood)[w, a]).astype(object),
        "identical": np.asarray(identical, dtype=float)[w, a],
        "num_amb": n_amb[w],
        "window_index": np.asarray(window_index)[w],
    }, columns=list(WINDOW_COLUMNS))


# ----------------------------------------------------------------------------- cross: interpretation
def json_default(o):
    """numpy integers -> int; anything else unserialisable becomes null (as the reference's encoder hook)"""
    if isinstance(o, np.integer):
        return int(o)
    return None


def interpret_cross(summary, windows, result_accs, result_likelis, db_accessions, window_chr_ids):
    """Extend the inbred-style ``summary`` of a cross run (only when its case is >= 3) with the F1 / F2 /
    contamination call derived from the window table.  Returns True when the summary was extended."""
    if summary["interpretation"]["case"] < 3:
        return False
    win = np.asarray(windows["window_index"])
    acc = np.asarray(windows["acc"])
    n_amb = np.asarray(windows["num_amb"])
    best_identical = windows.groupby("window_index")["identical"].max()
    # NB: positions inside the sorted list of reported windows, compared below with window numbers -- kept
    # as the reference computes it
    identical_pos = np.flatnonzero(np.asarray(best_identical) == 1)
    n_windows = len(np.unique(win))
    summary["identical_windows"] = [float(len(identical_pos)) / n_windows if n_windows > 0 else np.nan, n_windows]
    homo_windows = np.intersect1d(win[n_amb < 20], identical_pos)
    in_homo = np.isin(win, homo_windows)
    names, counts = np.unique(acc[in_homo], return_counts=True)
    summary["matches"] = [(names[i], int(counts[i])) for i in np.argsort(-counts)]

    best = int(np.argsort(result_likelis)[0])
    is_insilico = ~np.isin(result_accs, db_accessions)
    no_windows = {"chr_bins": None, "coordinates": {"x": None, "y": None}}
    if is_insilico[best]:
        mother, father = result_accs[best].split("x")[0], result_accs[best].split("x")[1]
        summary["interpretation"] = {"case": 5, "text": "Sample may be a F1! or a contamination!"}
        summary["parents"] = {"mother": [mother, 1], "father": [father, 1]}
        summary["genotype_windows"] = no_windows
        return True
    names, counts = np.unique(acc[n_amb == 1], return_counts=True)          # windows with a single candidate
    if len(names) == 0:
        summary["interpretation"] = {"case": 7, "text": "Sample may just be contamination!"}
        summary["genotype_windows"] = no_windows
        summary["parents"] = {"mother": [None, 0], "father": [None, 1]}
        return True
    lead = np.argsort(-counts)[:2]
    parents = names[lead].astype("str")
    support = counts[lead].astype("int")
    xs = np.array(np.unique(win), dtype="int")
    ys = np.repeat("NA", len(xs)).astype("S25")
    acc_text = acc.astype("str")
    for name in parents:
        owned = win[(acc_text == name) & in_homo]
        ys[np.isin(xs, owned)] = name
    if len(parents) == 1:
        summary["interpretation"] = {"case": 6, "text": "Sample may be a F2! but only one parent found!"}
        summary["parents"] = {"mother": [parents[0], support[0]], "father": ["NA", "NA"]}
        chr_bins = None
    else:
        summary["interpretation"] = {"case": 6, "text": "Sample may be a F2!"}
        summary["parents"] = {"mother": [parents[0], support[0]], "father": [parents[1], support[1]]}
        ids, n_per = np.unique(window_chr_ids, return_counts=True)
        chr_bins = dict((ids[i], n_per[i]) for i in range(len(ids)))
    summary["genotype_windows"] = {"chr_bins": chr_bins, "coordinates": {"x": xs.tolist(), "y": ys.tolist()}}
    return True
