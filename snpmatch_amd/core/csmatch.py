"""
``snpmatch cross`` on MI355X.

Public surface follows the reference module ``snpmatch.core.csmatch`` (core/csmatch.py:16-200):
``CrossIdentifier`` with ``cross_identifier``, ``get_window_data`` (static), ``window_genotyper``,
``match_insilico_f1s``, ``cross_interpreter``; ``convert_int64``; ``potatoCrossIdentifier``;
``chunk_size``.

The reference scores window after window (one ``matchGTsAccs`` call each, :80-90).  Here the matched DB
rows of ALL windows form one device query with a segment per window -- by default the reference-order kernel
(fp64 window scores with the reference's bits: ``windowscore.txt`` byte-identical), with SNPMATCH_CROSS_FAST=1 the
segmented streaming pass with its certificate (counts exact, float scores to ~1e-12) -- the per-window likelihoods / minima / ratios are one
``k_likelihood`` launch with a row per window, and the binomial identity test runs on the device as well.  The window table and the
JSON interpretation are host glue (``_report``); the 45 in-silico F1s of the ten best accessions are one
more device call (``k_f1_*``, numpy's summation order).
"""
import itertools
import json
import logging
import os

import numpy as np
from ._report import pd          # pandas, imported at first use

from . import _report
from . import genomes
from . import parsers
from . import snp_genotype  # noqa: F401  (part of the reference's module surface)
from . import snpmatch
from .. import _lib
from .. import dist
from .. import engine

log = logging.getLogger(__name__)
chunk_size = 1000


def _window_segments_sorted(genome, db, sample, bin_len):
    """``_window_segments`` for the normal case -- per chromosome the DB rows are one range with strictly
    increasing positions and so are the sample's entries: one native sorted intersection per chromosome, the
    window of a matched position is (pos - 1) // bin_len.  Returns None when the inputs are not of that form."""
    bin_len = int(bin_len)
    db_ids = genomes._bare(db.chrs)
    codes = getattr(sample, "g_chr_codes", None)
    if codes is not None and len(codes) == len(sample.chrs):       # filter_chr_names already named every SNP's chromosome
        smp_ids, inverse = genomes._bare(sample.g_chrs_ids), codes      # (lower-cased like the genome's ids; 'Chr1' / 'chr1' share a code)
    else:
        names, inverse = np.unique(np.asarray(sample.chrs, dtype="str"), return_inverse=True)
        smp_ids = genomes._bare(names)
    genome._check(db_ids, "genotype hdf5 file")
    genome._check(np.unique(smp_ids), "given SNPs")
    db_pos = db.__dict__.get("_positions_i64")
    if db_pos is None:
        db_pos = db._positions_i64 = np.ascontiguousarray(db.positions, dtype=np.int64)
    increasing = db.__dict__.setdefault("_increasing_chr", {})
    smp_pos = np.ascontiguousarray(sample.pos, dtype=np.int64)
    per_db, per_sample, counts, chrom = [], [], [], []
    # a sample sorted by chromosome names each one in ONE run of `inverse`: the runs are found once (one pass over the codes
    # instead of one per chromosome); anything else takes the per-chromosome membership tests
    inverse = np.asarray(inverse)
    cuts = np.flatnonzero(inverse[1:] != inverse[:-1]) + 1 if len(inverse) > 1 else np.zeros(0, dtype=np.int64)
    run_start = np.concatenate(([0], cuts)).astype(np.int64) if len(inverse) else np.zeros(0, dtype=np.int64)
    run_end = np.concatenate((cuts, [len(inverse)])).astype(np.int64) if len(inverse) else np.zeros(0, dtype=np.int64)
    run_code = inverse[run_start] if len(inverse) else np.zeros(0, dtype=np.int64)
    runs_are_chromosomes = len(np.unique(run_code)) == len(run_code)
    for chr_ix, cid in enumerate(genome.chrs_ids):
        n_win = len(range(1, int(genome.chrlen[chr_ix]), bin_len))
        chrom.append(np.full(n_win, chr_ix, dtype=int))
        where = np.flatnonzero(db_ids == cid)
        wanted = np.flatnonzero(smp_ids == cid)
        if runs_are_chromosomes and len(wanted) == 1:
            k = np.flatnonzero(run_code == wanted[0])
            mine = np.arange(run_start[k[0]], run_end[k[0]]) if len(k) else np.zeros(0, dtype=np.int64)
        else:
            mine = np.flatnonzero(inverse == wanted[0] if len(wanted) == 1 else np.isin(inverse, wanted))
        if len(where) == 0 or len(mine) == 0 or n_win == 0:
            counts.append(np.zeros(n_win, dtype=np.int64))
            continue
        row0, row1 = int(db.chr_regions[where[0]][0]), int(db.chr_regions[where[0]][1])
        one_run = int(mine[-1]) - int(mine[0]) + 1 == len(mine)
        p1, p2 = db_pos[row0:row1], (smp_pos[int(mine[0]):int(mine[-1]) + 1] if one_run else smp_pos[mine])
        if cid not in increasing:
            increasing[cid] = bool(len(p1) == 0 or (p1[0] >= 1 and np.all(p1[1:] > p1[:-1])))
        if not increasing[cid] or not one_run or p2[0] < 1:
            return None
        hit = _lib.intersect_sorted(p1, p2, a_verified=True)
        if hit is None:                                    # sample positions not strictly increasing
            return None
        win = (p2[hit[1]] - 1) // bin_len                  # (matched positions are equal: the sample side is the short, contiguous one)
        inside = win < n_win
        per_db.append(row0 + hit[0][inside])
        per_sample.append(int(mine[0]) + hit[1][inside])
        counts.append(np.bincount(win[inside], minlength=n_win).astype(np.int64))
    empty = np.zeros(0, dtype=int)
    offsets = np.concatenate(([0], np.cumsum(np.concatenate(counts)))).astype(np.int64) if counts else np.zeros(1, np.int64)
    return (np.concatenate(per_db).astype(int) if per_db else empty,
            np.concatenate(per_sample).astype(int) if per_sample else empty,
            offsets, np.concatenate(chrom) if chrom else empty)


def _window_segments(genome, db, sample, bin_len):
    """Matched (DB row, sample row) pairs grouped by genome window.
    Returns db_rows, sample_rows (concatenated in window order), offsets [n_win + 1], chromosome index per window."""
    fast = _window_segments_sorted(genome, db, sample, bin_len)
    if fast is not None:
        return fast
    db_pos = np.asarray(db.positions)
    per_db, per_sample, offsets, chrom = [], [], [0], []
    windows = zip(genome.get_bins_genome(db, bin_len), genome.get_bins_arrays(sample.chrs, sample.pos, bin_len))
    for (chr_ix, _, db_ix), (_, _, smp_ix) in windows:
        db_ix = np.array(db_ix, dtype=int)
        smp_ix = np.array(smp_ix, dtype=int)
        in_db, in_smp = db_pos[db_ix], sample.pos[smp_ix]
        per_db.append(db_ix[np.isin(in_db, in_smp)])
        per_sample.append(smp_ix[np.isin(in_smp, in_db)])
        offsets.append(offsets[-1] + len(per_db[-1]))
        chrom.append(chr_ix)
    empty = np.zeros(0, dtype=int)
    return (np.concatenate(per_db) if per_db else empty, np.concatenate(per_sample) if per_sample else empty,
            np.array(offsets, dtype=np.int64), np.array(chrom, dtype=int))


class CrossIdentifier(object):

    def __init__(self, inputs, g, genome_id, binLen, output_id="cross.identifier", run_identifier=True,
                 identity_error_rate=0.02, skip_db_hets=False):
        assert type(inputs) is parsers.ParseInputs, "provide a parsers class"
        inputs.filter_chr_names()
        self.inputs, self.g = inputs, g
        self.genome = genomes.Genome(genome_id)
        self.binLen = binLen
        job = dist.job()             # accession-sharded run: rank 0 writes, the other ranks use a scratch prefix
        self.output_id = job.output_prefix(output_id) if job else output_id
        self.error_rate = identity_error_rate
        self._skip_db_hets = skip_db_hets
        if run_identifier:
            self.cross_identifier()

    def cross_identifier(self):
        """windows -> totals -> in-silico F1s -> interpretation; writes the four output files"""
        summary_file = self.output_id + ".scores.txt.matches.json"
        totals = self.window_genotyper(self.output_id + '.windowscore.txt')
        totals.print_json_output(summary_file)
        codes = self.inputs.gt_codes_of(totals.matchedTarInd) if hasattr(self.inputs, "gt_codes_of") else None
        snpmatch.getHeterozygosity(self.inputs.gt[totals.matchedTarInd] if codes is None else codes, summary_file, _codes=codes)
        with open(summary_file) as fh:
            self.cross_identfier_json = json.load(fh)                   # attribute name as in the reference
        self.result = self.match_insilico_f1s(totals, self.output_id + '.scores.txt')
        self.cross_interpreter(self.output_id + ".matches.json")

    @staticmethod
    def get_window_data(bin_inds, AccList, ScoreList, NumInfoSites, error_rate=0.02):
        """table rows of ONE window from its per-accession (score, informative) vectors"""
        scores = np.asarray(ScoreList, dtype=float)
        ninfo = np.asarray(NumInfoSites)
        lik, lrt = snpmatch.GenotyperOutput.calculate_likelihoods(scores, ninfo)
        same = snpmatch.np_test_identity(x=scores, n=ninfo, error_rate=error_rate)
        return _report.window_rows(bin_inds, np.asarray(AccList), scores, ninfo, lik, lrt, same, snpmatch.lr_thres)

    def window_genotyper(self, out_file, mask_acc_ix=None):
        """Score every genome window; fills ``self.windows_data`` and returns the whole-genome totals as a
        ``GenotyperOutput`` (with ``matchedTarInd`` and ``winds_chrs`` attached).  With ``out_file`` None the
        pair [windows_data, totals] is returned instead of writing the table."""
        n_acc = len(self.g.accessions)
        shown = np.arange(n_acc)
        if mask_acc_ix is not None:
            assert type(mask_acc_ix) is np.ndarray, "please provide numpy array of acc indices to be masked"
            shown = np.setdiff1d(shown, mask_acc_ix)
        db_rows, sample_rows, offsets, win_chr = _window_segments(self.genome, self.g.g, self.inputs, self.binLen)
        n_matched = int(offsets[-1])

        query = self.g.panel().query(db_rows, self.inputs.wei[sample_rows, ])
        # default (round 4): every window in the reference's summation order -- fp64 window scores with the reference's
        # bits, windowscore.txt byte-identical, which is what a drop-in is diffed against; at the 1001-Genomes shape the two
        # modes cost the same (0.41 against 0.46 ms for 399 windows, profiles/r04b_real_panel_legs_one_call.json).
        # SNPMATCH_CROSS_FAST=1: the segmented streaming pass with the certificate per (window, accession) and for the totals --
        # snps_match, snps_info, num_amb, the totals and every file derived from them identical, the float columns of
        # windowscore.txt (score, likelihood) equal to ~1e-12 relative (north_star asks for 1e-6): the faster mode on wide panels.
        # SNPMATCH_CROSS_STRICT=1 (round 3's switch) still forces the reference order.
        strict = (os.environ.get("SNPMATCH_CROSS_STRICT", "0") not in ("", "0")) or \
                 (os.environ.get("SNPMATCH_CROSS_FAST", "0") in ("", "0"))
        fast = not strict
        log.info("window scoring mode: %s", "segmented fast pass + certificate (SNPMATCH_CROSS_FAST)" if fast else "reference order")
        w_score, w_ninfo, tot_score, tot_ninfo = query.run_windows(offsets, self._skip_db_hets, fast=fast)
        query.free()
        job = dist.job()
        if job is not None:          # per-window and total results of this rank's accessions -> whole arrays everywhere
            w_score, w_ninfo = job.gather_windows(w_score, w_ninfo, n_acc)
            tot_score, tot_ninfo = job.gather_scores(tot_score, tot_ninfo, n_acc)

        accs = np.asarray(self.g.accessions)[shown]
        self.windows_data = pd.DataFrame(columns=list(_report.WINDOW_COLUMNS))
        filled = np.flatnonzero(np.diff(offsets) > 0)                     # empty windows produce no rows
        if len(filled):
            dev = engine.default_context()
            if mask_acc_ix is None:               # every accession shown: one row selection, no column gather
                sc = np.ascontiguousarray(w_score if len(filled) == len(w_score) else w_score[filled])
                ni = np.ascontiguousarray(w_ninfo if len(filled) == len(w_ninfo) else w_ninfo[filled])
            else:
                sc = np.ascontiguousarray(w_score[filled][:, shown])
                ni = np.ascontiguousarray(w_ninfo[filled][:, shown])
            lik, lrt = dev.likelihood(sc, ni)                              # one device row per window
            same = dev.binom_identity(sc.ravel(), ni.ravel(), self.error_rate, 0.05).reshape(sc.shape)
            self.windows_data = _report.window_table(filled + 1, accs, sc, ni, lik, lrt, same, snpmatch.lr_thres)
            log.info("Done analysing %s positions", n_matched)

        totals = snpmatch.GenotyperOutput(accs, tot_score[shown], tot_ninfo[shown],
                                          snpmatch.get_fraction(n_matched, len(self.inputs.pos)), n_matched, self.inputs.dp)
        totals.matchedTarInd = sample_rows
        totals.winds_chrs = np.asarray(self.genome.chrs_ids)[win_chr] if len(win_chr) else np.zeros(0, dtype="U1")
        if out_file is None:
            return [self.windows_data, totals]
        self.windows_data.to_csv(out_file, sep="\t", index=False)
        return totals

    def match_insilico_f1s(self, snpmatch_result, out_file):
        """Append the 45 pairwise in-silico F1s of the ten accessions with the highest match fraction to the
        result (a cross of two accessions is homozygous where both agree and heterozygous where they differ)."""
        assert type(snpmatch_result) is snpmatch.GenotyperOutput, "Please provide GenotyperOutput class as input"
        if not hasattr(snpmatch_result, 'probabilies'):
            snpmatch_result.get_probabilities()
        log.info("simulating F1s for top 10 accessions")
        best = np.argsort(-snpmatch_result.probabilies)[0:10]
        db_rows, sample_rows = self.g.get_positions_idxs(self.inputs.chrs, self.inputs.pos, _parsed=self.inputs)
        # all pairs in one device call (k_f1_*): per pair np.sum(W[alt, 2]) + np.sum(W[ref, 0]) + np.sum(W[het, 1])
        # with numpy's summation order, so the float scores printed below carry the reference's digits
        query = self.g.panel().query(db_rows, self.inputs.wei[sample_rows, ])
        job = dist.job()
        if job is None:
            extra_s, extra_n = query.f1_pairs(best)
        else:
            # the ten columns live on different GPUs: every rank reads the ones it holds at the matched rows, a
            # byte-sum over ranks brings them together, and the crosses run on that ten-column panel
            a0, a1 = self.g._shard
            mine = np.flatnonzero((best >= a0) & (best < a1))
            codes = np.zeros((len(best), len(db_rows)), dtype=np.uint8)
            if len(mine):
                codes[mine] = query.gather_columns(best[mine] - a0)
            codes = job.sum_bytes(codes)
            small = engine.Panel.from_host(self.g.panel().ctx, np.ascontiguousarray(codes.T).view(np.int8))
            q10 = engine.Query(small, None, self.inputs.wei[sample_rows, ])
            extra_s, extra_n = q10.f1_pairs(np.arange(len(best)))
            q10.free()
            small.free()
        query.free()
        extra_a = [self.g.accessions[i] + "x" + self.g.accessions[j] for i, j in itertools.combinations(best, 2)]
        if extra_a:
            snpmatch_result.scores = np.append(snpmatch_result.scores, extra_s)      # becomes a float column
            snpmatch_result.ninfo = np.append(snpmatch_result.ninfo, extra_n)
            snpmatch_result.accs = np.append(snpmatch_result.accs, extra_a)
        if out_file is not None:
            snpmatch_result.print_out_table(out_file, _frame=False)
        return snpmatch_result

    def cross_interpreter(self, out_file):
        """F1 / F2 / contamination call; written only for ambiguous samples (inbred case >= 3)"""
        assert 'cross_identfier_json' in dir(self), "run cross identifier first!"
        assert 'windows_data' in dir(self), "run window genotyper first!"
        log.info("running cross interpreter!")
        if _report.interpret_cross(self.cross_identfier_json, self.windows_data, self.result.accs, self.result.likelis,
                                   self.g.accessions, self.result.winds_chrs):
            _report.dump_json(self.cross_identfier_json, out_file, default=convert_int64)


def convert_int64(o):
    """JSON encoder hook: numpy int64 -> int (anything else unserialisable is written as null)"""
    if isinstance(o, np.int64):
        return int(o)


def potatoCrossIdentifier(args):
    """entry point of ``snpmatch cross`` (args as for inbred plus genome, binLen)"""
    inputs = snpmatch.parse_inputs_once(args['inFile'], args['logDebug'])
    log.info("loading genotype files!")
    g = snp_genotype.Genotype(args['hdf5File'], args['hdf5accFile'])
    log.info("running cross identifier!")
    CrossIdentifier(inputs, g, args['genome'], args['binLen'], args['outFile'], run_identifier=True,
                    skip_db_hets=args['skip_db_hets'])
    log.info("finished!")
