/*
 * CPU ORACLE (plain C) for the SNPmatch Genotyper hot path -- TEST INFRASTRUCTURE ONLY.
 *
 * Not part of the product.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load the library built from this file
 * (oracle/liboracle.so).  The product (snpmatch_amd) never links or loads it.
 *
 * It restates, with the reference's exact fp64 summation order, the arithmetic of
 *   matchGTsAccs            /root/reference/snpmatch/core/snpmatch.py:74-89
 *   Genotyper.genotyper     /root/reference/snpmatch/core/snpmatch.py:207-225 (chunk loop)
 *   window_genotyper        /root/reference/snpmatch/core/csmatch.py:80-90    (per-window calls)
 *   likeliTest              /root/reference/snpmatch/core/snpmatch.py:40-55
 * and is pinned bit-for-bit against golden vectors generated from the reference
 * (tests/golden/make_golden.py, tests/test_oracle_golden.py).
 *
 * Build: see oracle/Makefile.  -ffp-contract=off and no -ffast-math are REQUIRED:
 * the order and rounding of every fp64 add is part of the contract.
 *
 * Order model (one matchGTsAccs call over rows [r0,r1)):
 *   for c in (ref: db==0 -> W[:,0]; het: db==2 -> W[:,1]; alt: db==1 -> W[:,2]):
 *       A_c[a] = sequential sum over rows of (db[row,a]==code ? W[row,col] : +0.0)
 *   score[a] = ((0 + A_ref[a]) + A_het[a]) + A_alt[a]
 *   ninfo[a] = (r1-r0) - #{rows: db[row,a] < 0}
 * with skip_hets: db==2 is treated as -1 (missing) for both score and ninfo.
 * Panels of ONE accession (n_acc == 1): A_c is numpy's sum of a contiguous vector instead (np_sum_vector below).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static const int CAT_CODE[3] = {0, 2, 1};
static const int CAT_COL[3]  = {0, 1, 2};

/* numpy's DOUBLE_pairwise_sum (numpy/_core/src/umath/loops_utils.h.src), restated: fewer than 8 elements
 * sequentially; up to 128 elements eight strided accumulators combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7))
 * and the tail added one by one; longer runs split at n/2 rounded down to a multiple of 8. */
static double np_pairwise(const double *a, int64_t n)
{
    if (n < 8) {
        double res = 0.0;
        for (int64_t i = 0; i < n; ++i) res = res + a[i];
        return res;
    }
    if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        int64_t i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] = r[j] + a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res = res + a[i];
        return res;
    }
    int64_t n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise(a, n2) + np_pairwise(a + n2, n - n2);
}

/* np.sum of a contiguous fp64 vector as numpy 2.2 evaluates it: 8192-element buffer pieces, each summed
 * pairwise, added in order onto 0.0 (pinned against np.sum and the reference in tests/test_oracle_golden.py). */
static double np_sum_vector(const double *a, int64_t n)
{
    double res = 0.0;
    for (int64_t c0 = 0; c0 < n; c0 += 8192)
        res = res + np_pairwise(a + c0, n - c0 < 8192 ? n - c0 : 8192);
    return res;
}

/* One matchGTsAccs call.  db: [n rows] gathered through row_idx (or dense when row_idx==NULL)
 * with row pitch `pitch` bytes; wei: [n,3] rows r0.. of the matched-weight array.
 * out_score/out_ninfo: [n_acc]; tmp: scratch [n_acc] doubles. */
static void match_call(const int8_t *db, int64_t pitch, const int64_t *row_idx, int64_t r0, int64_t r1,
                       const double *wei, int64_t n_acc, int skip_hets, double *out_score,
                       int64_t *out_ninfo, double *tmp)
{
    for (int64_t a = 0; a < n_acc; ++a) { out_score[a] = 0.0; out_ninfo[a] = r1 - r0; }
    /* ONE accession: the reference's [1, n] product is contiguous along the reduced axis, numpy sums it as a
     * vector (pairwise inside 8192-element pieces), not row after row (snpmatch.py:85-87 with N_acc == 1) */
    double *vec = NULL;
    if (n_acc == 1 && r1 > r0) vec = (double *)malloc(sizeof(double) * (size_t)(r1 - r0));
    for (int c = 0; c < 3; ++c) {
        const int8_t code = (int8_t)CAT_CODE[c];
        const int col = CAT_COL[c];
        if (skip_hets && code == 2) {
            /* het rows became missing: A_het == 0, adding +0.0 leaves score unchanged */
            continue;
        }
        if (vec) {
            for (int64_t r = r0; r < r1; ++r) {
                const int8_t *row = db + (row_idx ? row_idx[r] : r) * pitch;
                vec[r - r0] = (row[0] == code) ? wei[3 * r + col] : 0.0;
            }
            out_score[0] = out_score[0] + np_sum_vector(vec, r1 - r0);
            continue;
        }
        for (int64_t a = 0; a < n_acc; ++a) tmp[a] = 0.0;
        for (int64_t r = r0; r < r1; ++r) {
            const int8_t *row = db + (row_idx ? row_idx[r] : r) * pitch;
            const double w = wei[3 * r + col];
            for (int64_t a = 0; a < n_acc; ++a) tmp[a] = tmp[a] + (row[a] == code ? w : 0.0);
        }
        for (int64_t a = 0; a < n_acc; ++a) out_score[a] = out_score[a] + tmp[a];
    }
    free(vec);
    for (int64_t r = r0; r < r1; ++r) {
        const int8_t *row = db + (row_idx ? row_idx[r] : r) * pitch;
        if (skip_hets) {
            for (int64_t a = 0; a < n_acc; ++a) out_ninfo[a] -= (row[a] < 0 || row[a] == 2);
        } else {
            for (int64_t a = 0; a < n_acc; ++a) out_ninfo[a] -= (row[a] < 0);
        }
    }
}

/* matchGTsAccs on a dense [n, n_acc] block. */
int oracle_match(const double *wei, const int8_t *db, int64_t n, int64_t n_acc, int64_t pitch,
                 int skip_hets, double *score, int64_t *ninfo)
{
    double *tmp = (double *)malloc(sizeof(double) * (size_t)(n_acc > 0 ? n_acc : 1));
    if (!tmp) return -1;
    match_call(db, pitch, NULL, 0, n, wei, n_acc, skip_hets, score, ninfo, tmp);
    free(tmp);
    return 0;
}

/* Genotyper.genotyper chunk loop: rows of the panel gathered through row_idx[n] (NULL = dense),
 * weights wei[n,3] already in matched order; ScoreList += chunk, NumInfoSites += chunk. */
int oracle_genotyper(const int8_t *db, int64_t pitch, int64_t n_acc, const int64_t *row_idx,
                     const double *wei, int64_t n, int64_t chunk, int skip_hets, double *score,
                     int64_t *ninfo)
{
    size_t na = (size_t)(n_acc > 0 ? n_acc : 1);
    double *tmp = (double *)malloc(sizeof(double) * na);
    double *cs = (double *)malloc(sizeof(double) * na);
    int64_t *cn = (int64_t *)malloc(sizeof(int64_t) * na);
    if (!tmp || !cs || !cn) { free(tmp); free(cs); free(cn); return -1; }
    for (int64_t a = 0; a < n_acc; ++a) { score[a] = 0.0; ninfo[a] = 0; }
    if (chunk <= 0) chunk = 1000;
    for (int64_t j = 0; j < n; j += chunk) {
        int64_t r1 = j + chunk < n ? j + chunk : n;
        match_call(db, pitch, row_idx, j, r1, wei, n_acc, skip_hets, cs, cn, tmp);
        for (int64_t a = 0; a < n_acc; ++a) { score[a] = score[a] + cs[a]; ninfo[a] += cn[a]; }
    }
    free(tmp); free(cs); free(cn);
    return 0;
}

/* window_genotyper: one matchGTsAccs call per window [win_off[w], win_off[w+1]) of the matched
 * list; outputs per-window [n_win, n_acc] plus totals accumulated in window order.
 * Empty windows produce zeros and do not touch the totals (csmatch.py:86). */
int oracle_windows(const int8_t *db, int64_t pitch, int64_t n_acc, const int64_t *row_idx,
                   const double *wei, const int64_t *win_off, int64_t n_win, int skip_hets,
                   double *score, int64_t *ninfo, double *tot_score, int64_t *tot_ninfo)
{
    size_t na = (size_t)(n_acc > 0 ? n_acc : 1);
    double *tmp = (double *)malloc(sizeof(double) * na);
    if (!tmp) return -1;
    for (int64_t a = 0; a < n_acc; ++a) { tot_score[a] = 0.0; tot_ninfo[a] = 0; }
    for (int64_t w = 0; w < n_win; ++w) {
        double *s = score + w * n_acc;
        int64_t *ni = ninfo + w * n_acc;
        if (win_off[w + 1] > win_off[w]) {
            match_call(db, pitch, row_idx, win_off[w], win_off[w + 1], wei, n_acc, skip_hets, s, ni, tmp);
            for (int64_t a = 0; a < n_acc; ++a) { tot_score[a] = tot_score[a] + s[a]; tot_ninfo[a] += ni[a]; }
        } else {
            for (int64_t a = 0; a < n_acc; ++a) { s[a] = 0.0; ni[a] = 0; }
        }
    }
    free(tmp);
    return 0;
}

/* likeliTest for arrays; y may be fractional (cross windows).  Returns nan where the
 * reference returns nan; the reference asserts y <= n, here that case also yields nan and
 * sets *bad = 1. */
int oracle_likelihood(const double *y, const int64_t *n, int64_t len, double *lik, int *bad)
{
    const double p = 0.99999999;
    if (bad) *bad = 0;
    for (int64_t i = 0; i < len; ++i) {
        double nn = (double)n[i], yy = y[i];
        if (yy > nn) { lik[i] = NAN; if (bad) *bad = 1; continue; }
        if (n[i] == 0) { lik[i] = NAN; continue; }
        if (yy == nn) { lik[i] = 1.0; continue; }
        if (yy > 0) {
            double ps = yy / nn;
            double a = yy * log(ps / p);
            double b = (nn - yy) * log((1 - ps) / (1 - p));
            lik[i] = a + b;
        } else {
            lik[i] = NAN;
        }
    }
    return 0;
}
