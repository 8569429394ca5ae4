"""
CPU ORACLE for the SNPmatch Genotyper hot path  --  TEST INFRASTRUCTURE ONLY.

This module is a numpy restatement of the reference's scoring arithmetic.  It is
NOT part of the product: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  The product path
(``snpmatch_amd``) never imports anything under ``oracle/`` and fails loudly if
the HIP library is missing.

Parity pinning: every function here is checked in ``tests/test_oracle_golden.py``
against (a) the reference's own known-answer tests
(``/root/reference/tests/test_inbred.py:22-24``, README rows ``README.md:90-92``)
and (b) golden vectors produced by running the unmodified reference in the
build container (``tests/golden/make_golden.py``; the reference cannot travel to
the GPU box, the vectors can).

Each function cites the reference lines it follows (paths relative to
``/root/reference``).

Summation-order model (verified bit-for-bit against the reference on numpy
2.2.6, see tests): ``matchGTsAccs`` reduces an F-ordered ``[N_acc, n]`` product
over its strided axis, which numpy executes as plain row-after-row adds
(N_acc >= 2).  A panel of ONE accession is the exception: a ``[1, n]`` array is
contiguous along the reduced axis too, and numpy then takes its contiguous-axis
route -- ``A_c = np.sum`` of a vector: pairwise summation (leaves of <= 128
elements with eight strided accumulators) inside 8192-element buffer pieces,
the pieces added in order (``np_sum_restated`` below; goldens g1b / g2b / g5b).
So for one call with ``n`` SNP rows and N_acc >= 2::

    A_c[a] = (((0 + t_c[0,a]) + t_c[1,a]) + ...) + t_c[n-1,a]     c in (ref, het, alt)
    score[a] = ((0 + A_ref[a]) + A_het[a]) + A_alt[a]

with ``t_c[s,a] = W[s,col(c)] if db[s,a] == code(c) else 0``, codes
ref=0 -> W[:,0], het=2 -> W[:,1], alt=1 -> W[:,2]; negative db values are
missing and match nothing.
"""
import numpy as np
import numpy.ma

P_MATCH = 0.99999999          # core/snpmatch.py:44
LR_THRES = 3.841              # core/snpmatch.py:17
CHUNK_SIZE = 1000             # core/snpmatch.py:173, core/csmatch.py:16

# (db code, weight column) in the order the reference adds them: core/snpmatch.py:85-87
CATEGORIES = ((0, 0), (2, 1), (1, 2))


# --------------------------------------------------------------------------- a1
def match_gts_accs(wei, db, skip_hets_db=False):
    """Exact sequential model of ``matchGTsAccs`` (core/snpmatch.py:74-89).

    wei: float64 [n,3]; db: int8 [n, n_acc].  Returns (score f64 [n_acc], ninfo i64 [n_acc]).
    Row loop in Python, vectorised over accessions: use for small/medium cases.
    """
    wei = np.asarray(wei, dtype=np.float64)
    db = np.asarray(db)
    assert wei.shape[0] == db.shape[0], \
        "please provide same number of positions for both sample and db"      # :75
    assert wei.ndim == 2 and wei.shape[1] == 3, \
        "SNP weights should be a np.array with  shape == n,3"                # :76
    if skip_hets_db:                                                          # :78-79 (on a copy)
        db = np.where(db == 2, np.int8(-1), db)
    n, n_acc = db.shape
    score = np.zeros(n_acc)
    if n_acc == 1:
        # ONE accession: the reference's [1, n] product is contiguous along the axis it reduces, and numpy sums a
        # contiguous axis pairwise (8192-element buffer pieces added in order), not row after row
        for code, col in CATEGORIES:                                          # :85-87
            t = (db[:, 0] == code).astype(np.int64) * wei[:, col]
            score = score + np_sum_restated(t)
        ninfo = n - (db < 0).sum(axis=0).astype(np.int64)                     # :88
        return score, ninfo
    for code, col in CATEGORIES:                                              # :85-87
        acc = np.zeros(n_acc)
        for s in range(n):
            # (mask as int) * weight, then added: a non-matching element adds +0.0
            acc = acc + (db[s] == code).astype(np.int64) * wei[s, col]
        score = score + acc
    ninfo = n - (db < 0).sum(axis=0).astype(np.int64)                         # :88
    return score, ninfo


def match_gts_accs_graph(wei, db, skip_hets_db=False):
    """Same *expression graph* as the reference (masked arrays, int casts, fp64
    products, axis reductions: core/snpmatch.py:81-88).  This is what the
    reference's CPU path costs; ``bench.py`` times it as ``cpu_baseline``.
    """
    assert wei.shape[0] == db.shape[0]
    assert wei.shape[1] == 3
    if skip_hets_db:
        db = db.copy()
        db[db == 2] = -1
    n_acc = db.shape[1]
    informative = numpy.ma.masked_less(db, 0)
    total = np.zeros(n_acc)
    for code, col in CATEGORIES:
        target = np.full(db.shape, code, dtype="int8")
        hits = np.array(numpy.ma.masked_less(db, 0) == target, dtype=int).T
        total = total + np.multiply(hits, wei[:, col]).sum(axis=1)
    ninfo = np.repeat(db.shape[0], n_acc) - np.sum(informative.mask.astype(int), axis=0)
    return total, ninfo


# --------------------------------------------------------------------------- a2
def genotyper_scores(wei, db_rows, chunk_size=CHUNK_SIZE, skip_hets_db=False, match=match_gts_accs):
    """Chunk loop of ``Genotyper.genotyper`` (core/snpmatch.py:207-225).

    wei: [n,3] weights of the matched sample SNPs (already gathered, :221);
    db_rows: int8 [n, n_acc] matched DB rows (already gathered, :222).
    Returns (ScoreList f64, NumInfoSites i64) before the int truncation of :96.
    """
    n, n_acc = db_rows.shape
    score = np.zeros(n_acc, dtype="float")                                    # :208
    ninfo = np.zeros(n_acc, dtype="uint32")                                   # :209
    for j in range(0, n, chunk_size):                                         # :218
        t_s, t_n = match(wei[j:j + chunk_size], db_rows[j:j + chunk_size], skip_hets_db)
        score = score + t_s                                                   # :224
        ninfo = ninfo + t_n                                                   # :225
    return score, ninfo.astype(np.int64)


# --------------------------------------------------------------------------- a3
def likeli_test(n, y):
    """``likeliTest`` (core/snpmatch.py:40-55)."""
    assert y <= n, "provided y is greater than n"                            # :43
    p = P_MATCH
    if n == 0:
        return np.nan                                                         # :45-46
    ps = float(y) / n                                                         # :47
    if y == n:
        return 1                                                              # :48-49
    if y > 0:
        a = y * np.log(ps / p)                                                # :51
        b = (n - y) * np.log((1 - ps) / (1 - p))                              # :52
        return a + b
    return np.nan                                                             # :54-55


def get_fraction(x, y, y_min=0):
    """core/snpmatch.py:25-28."""
    if y <= y_min:
        return np.nan
    return float(x) / y


# --------------------------------------------------------------------------- a4
def calculate_likelihoods(scores, ninfo, amin="calc"):
    """``GenotyperOutput.calculate_likelihoods`` (core/snpmatch.py:106-117)."""
    scores = np.asarray(scores)
    ninfo = np.asarray(ninfo)
    lik = np.array([likeli_test(ninfo[i], scores[i]) for i in range(len(scores))], dtype=float)
    if isinstance(amin, str) and amin == "calc":
        top = np.nanmin(lik)                                                  # :112
    else:
        top = float(amin)                                                     # :114
    lrt = np.array([get_fraction(lik[i], top) for i in range(len(scores))], dtype=float)
    return lik, lrt


# --------------------------------------------------------------------------- a5
def strip_chr(chrs):
    """``filter_chr_names`` (core/parsers.py:159-163): case-insensitive removal of 'chr'."""
    import re
    return np.array([re.sub("chr", "", str(c), flags=re.IGNORECASE) for c in chrs], dtype="str")


def get_common_positions(chr1, pos1, chr2, pos2):
    """``Genotype.get_common_positions`` (core/snp_genotype.py:46-68)."""
    assert len(chr1) == len(pos1)
    assert len(chr2) == len(pos2)
    c1 = strip_chr(chr1)
    c2 = strip_chr(chr2)
    pos1 = np.asarray(pos1)
    pos2 = np.asarray(pos2)

    def first_seen(c):
        _, idx = np.unique(c, return_index=True)
        return c[np.sort(idx)]

    ids1 = first_seen(c1) if len(c1) else c1
    ids2 = first_seen(c2) if len(c2) else c2
    common = np.intersect1d(ids1, ids2)
    common = ids1[np.isin(ids1, common)]          # ordered as in input 1 (the DB), :58
    out1 = np.zeros(0, dtype=int)
    out2 = np.zeros(0, dtype=int)
    for cid in common:                                                        # :61-67
        ix1 = np.where(c1 == cid)[0]
        ix2 = np.where(c2 == cid)[0]
        p1 = np.array(pos1[ix1], dtype=int)
        p2 = np.array(pos2[ix2], dtype=int)
        out1 = np.append(out1, ix1[np.where(np.isin(p1, p2, assume_unique=True))[0]])
        out2 = np.append(out2, ix2[np.where(np.isin(p2, p1, assume_unique=True))[0]])
    return out1, out2


# --------------------------------------------------------------------------- a7 (windows)
def segregating_rows(snps, accs_ix, block=1 << 16):
    """DB rows on which the listed accessions segregate, as `snpmatch inbred --refine` selects them
    (core/snp_genotype.py:188-211 with segregting_snps :378-383): per row the calls of those accessions as floats,
    negative -> NaN, sorted; run = (#equal neighbours) + 1, info = #non-NaN; kept where run / info < 1 and info > 0
    (i.e. at least two different informative calls).  None when more than half of all accessions are listed (:193)."""
    accs_ix = np.asarray(accs_ix)
    if len(accs_ix) > snps.shape[1] / 2:
        return None
    run = np.zeros(0, dtype=int)
    info = np.zeros(0, dtype=int)
    for j in range(0, snps.shape[0], block):                     # :199-203 walks the DB in blocks
        t = np.array(snps[j:j + block, :][:, accs_ix], dtype=float)
        t[t < 0] = np.nan                                         # :379
        t = np.sort(t, axis=1)                                    # :380
        info = np.append(info, np.sum(~np.isnan(t), axis=1))      # :381
        run = np.append(run, np.nansum(t[:, 1:] == t[:, :-1], axis=1) + 1)     # :382
    div = np.divide(run, info, out=np.zeros(len(run)), where=info != 0)         # :206
    return np.setdiff1d(np.where(div < 1)[0], np.where(info == 0)[0])            # :207-208


def bins_echr(real_chrlen, chr_pos, bin_len, rel_ix):
    """``get_bins_echr`` (core/genomes.py:111-127) as a list, same walk."""
    out = []
    ind = 0
    npos = len(chr_pos)
    for t in range(1, int(real_chrlen), int(bin_len)):
        lo, hi = int(t), int(t) + int(bin_len) - 1
        result = []
        skipped = True
        k = ind
        while k < npos:
            epos = chr_pos[k]
            k += 1
            if epos >= lo:
                if epos <= hi:
                    result.append(ind + rel_ix)
                elif epos > hi:
                    skipped = False
                    out.append(((lo, hi), result))
                    break
                ind += 1
        if skipped:
            out.append(((lo, hi), result))
    return out


def genome_windows_db(genome_chr_ids, genome_chrlen, db_chrs, db_chr_regions, db_positions, bin_len):
    """``Genome.get_bins_genome`` (core/genomes.py:73-91): list of (chr_ix, (lo,hi), [db row idx])."""
    g_ids = np.char.replace(np.char.lower(np.array(db_chrs, dtype="str")), "chr", "")
    out = []
    start = 0
    for chr_ix in range(len(genome_chr_ids)):
        t = np.where(g_ids == genome_chr_ids[chr_ix])[0]
        if len(t) == 0:
            chr_pos = np.zeros(0, dtype=int)
        else:
            start = int(db_chr_regions[t[0]][0])
            end = int(db_chr_regions[t[0]][1])
            chr_pos = db_positions[start:end]
        for b, idx in bins_echr(genome_chrlen[chr_ix], chr_pos, bin_len, start):
            out.append((chr_ix, b, idx))
    return out


def genome_windows_sample(genome_chr_ids, genome_chrlen, chrs, pos, bin_len):
    """``Genome.get_bins_arrays`` (core/genomes.py:93-108)."""
    c = np.char.replace(np.char.lower(np.array(chrs, dtype="str")), "chr", "")
    pos = np.asarray(pos)
    out = []
    for chr_ix in range(len(genome_chr_ids)):
        ix = np.where(c == genome_chr_ids[chr_ix])[0]
        rel = int(ix[0]) if len(ix) > 0 else 0
        for b, idx in bins_echr(genome_chrlen[chr_ix], pos[ix], bin_len, rel):
            out.append((chr_ix, b, idx))
    return out


def window_scores(wei, db_rows, win_off, skip_hets_db=False, match=match_gts_accs):
    """Per-window ``matchGTsAccs`` of ``window_genotyper`` (core/csmatch.py:80-90):
    one call per window over that window's matched rows (no 1000-row chunking),
    totals accumulated window after window (:88-89).

    wei [n,3], db_rows [n,n_acc] are the matched rows in window order; win_off [n_win+1].
    Returns (score [n_win,n_acc], ninfo [n_win,n_acc], tot_score [n_acc], tot_ninfo [n_acc]).
    """
    n_win = len(win_off) - 1
    n_acc = db_rows.shape[1]
    score = np.zeros((n_win, n_acc))
    ninfo = np.zeros((n_win, n_acc), dtype=np.int64)
    tot_s = np.zeros(n_acc, dtype="uint32")                                   # :72
    tot_n = np.zeros(n_acc, dtype="uint32")                                   # :73
    for w in range(n_win):
        a, b = int(win_off[w]), int(win_off[w + 1])
        if b > a:                                                             # :86
            s, ni = match(wei[a:b], db_rows[a:b], skip_hets_db)
            score[w] = s
            ninfo[w] = ni
            tot_s = tot_s + s
            tot_n = tot_n + ni
    return score, ninfo, np.asarray(tot_s, dtype=float), np.asarray(tot_n, dtype=np.int64)


# --------------------------------------------------------------------------- a8
def test_identity(x, n, error_rate=0.02, pthres=0.05):
    """``np_test_identity`` (core/snpmatch.py:57-72): binom.sf((n-x)-1, n, p) >= pthres."""
    from scipy import stats
    st = stats.binom.sf(np.asarray(n) - np.asarray(x) - 1, np.asarray(n), error_rate)
    return np.array(st >= pthres).astype(int)


test_identity.__test__ = False  # not a pytest test


# --------------------------------------------------------------------------- in-silico F1s
def insilico_f1_pairs(cols, wei):
    """match_insilico_f1s, core/csmatch.py:115-125: ``cols`` int8 [n, k] = the calls of the k selected accessions
    at the matched SNPs, ``wei`` float [n, 3].  Returns (score [k(k-1)/2], ninfo) in combination order.  The sums
    are numpy's own np.sum over freshly gathered contiguous vectors, exactly as the reference writes them."""
    import itertools
    cols = np.asarray(cols)
    wei = np.asarray(wei, dtype=float)
    score, ninfo = [], []
    for i, j in itertools.combinations(range(cols.shape[1]), 2):
        gtp1, gtp2 = cols[:, i], cols[:, j]
        homalt = np.where((gtp1 == 1) & (gtp2 == 1))[0]
        homref = np.where((gtp1 == 0) & (gtp2 == 0))[0]
        het = np.where((gtp1 != -1) & (gtp2 != -1) & (gtp1 != gtp2))[0]
        score.append(np.sum(wei[homalt, 2]) + np.sum(wei[homref, 0]) + np.sum(wei[het, 1]))
        ninfo.append(len(homalt) + len(homref) + len(het))
    return np.array(score, dtype=float), np.array(ninfo, dtype=np.int64)


NP_SUM_CHUNK = 8192           # numpy's default ufunc buffer size (np.getbufsize())
NP_PW_LEAF = 128              # PW_BLOCKSIZE of numpy's pairwise summation


def np_pairwise_sum(a):
    """numpy's DOUBLE_pairwise_sum (numpy/_core/src/umath/loops_utils.h.src), restated: the order the device
    kernels of the in-silico F1 scores reproduce."""
    n = len(a)
    if n < 8:
        res = np.float64(0.0)
        for x in a:
            res = res + x
        return res
    if n <= NP_PW_LEAF:
        r = np.array(a[:8], dtype=np.float64)
        k8 = n - n % 8
        for i in range(8, k8, 8):
            r = r + a[i:i + 8]
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))
        for i in range(k8, n):
            res = res + a[i]
        return res
    n2 = n // 2
    n2 -= n2 % 8
    return np_pairwise_sum(a[:n2]) + np_pairwise_sum(a[n2:])


def np_sum_restated(a):
    """np.sum of a contiguous float64 vector as numpy 2.2 evaluates it: the reduction runs over the ufunc
    buffer (8192 elements) at a time, each piece summed pairwise, pieces added in order starting from 0.0
    (checked bit-for-bit against np.sum in tests/test_oracle_golden.py)."""
    a = np.asarray(a, dtype=np.float64)
    res = np.float64(0.0)
    for c0 in range(0, len(a), NP_SUM_CHUNK):
        res = res + np_pairwise_sum(a[c0:c0 + NP_SUM_CHUNK])
    return res


# --------------------------------------------------------------------------- helpers
def weights_from_pl(pl):
    """PL -> weights (core/parsers.py:147-150): exp(PL / -10)."""
    return np.exp(np.asarray(pl, dtype=float) / (-10))


def weights_from_gt_codes(codes):
    """``get_wei_from_GT`` (core/parsers.py:132-139) from int8 codes (0 ref, 1 alt, 2 het)."""
    codes = np.asarray(codes)
    w = np.ones((len(codes), 3))
    w[codes != 0, 0] = 0
    w[codes != 1, 2] = 0
    w[codes != 2, 1] = 0
    return w
