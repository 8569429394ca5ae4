"""
``snpmatch inbred`` on MI355X.

Public names, arguments and error behaviour follow the reference module ``snpmatch.core.snpmatch``
(core/snpmatch.py:17-268) so that callers can switch import paths: ``matchGTsAccs``, ``likeliTest``,
``get_fraction`` / ``np_get_fraction``, ``np_binom_test`` / ``np_test_identity``, ``GenotyperOutput``,
``Genotyper``, ``getHeterozygosity``, ``potatoGenotyper`` and the thresholds ``lr_thres``,
``snp_thres``, ``prob_thres`` (module attributes, read at call time).

Division of labour
  * device (libsnpmatch_hip, include/snpmatch_hip.h): weighted compare-and-count over the HBM-resident
    panel (the reference's 1000-SNP chunk loop, :218-225), likelihoods / minimum / ratios (:40-55,
    :106-117), the binomial identity test (:57-72).  No CPU implementation of these exists here.
  * host (this file and ``_report``): argument checks with the reference's messages, result objects,
    ``*.scores.txt`` / ``*.matches.json``.
"""
import logging
import os
import sys

import numpy as np

from . import _report
from . import parsers
from . import snp_genotype
from .. import dist
from .. import engine

log = logging.getLogger(__name__)
lr_thres = 3.841
snp_thres = 4000
prob_thres = 0.98


def die(msg):
    sys.stderr.write('Error: ' + msg + '\n')
    sys.exit(1)


def get_fraction(x, y, y_min=0):
    """x / y, NaN when y <= y_min"""
    return np.nan if y <= y_min else float(x) / y


np_get_fraction = np.vectorize(get_fraction, excluded="y_min")


def _device():
    return engine.default_context()


# ----------------------------------------------------------------------------- kernels behind functions
def likeliTest(n, y):
    """likelihood of y matches out of n informative sites (k_likelihood on the device).
    NaN for n == 0 or y == 0, the sentinel 1 for a perfect match, AssertionError when y > n."""
    assert y <= n, "provided y is greater than n"
    if n == 0 or y == 0:
        return np.nan
    if y == n:
        return 1
    return float(_device().likelihood(np.array([float(y)]), np.array([int(n)]))[0][0])


def np_binom_test(x, n, p, alternative=None):
    """one-sided binomial test p-values ('greater' / 'larger': P(X >= x); 'less' / 'smaller': P(X <= x))"""
    x = np.atleast_1d(np.asarray(x, dtype=float))
    n = np.atleast_1d(np.asarray(n)).astype(np.int64)
    if alternative in ('larger', 'greater'):
        return _device().binom_identity(n - x, n, p, 0.0, return_sf=True)[1]          # sf(x - 1)
    if alternative in ('smaller', 'less'):
        return 1.0 - _device().binom_identity(n - x - 1, n, p, 0.0, return_sf=True)[1]  # 1 - sf(x)
    raise NotImplementedError("two-sided binomial test is not on the inbred / cross path")


def np_test_identity(x, n, error_rate=0.0005, pthres=0.05):
    """1 where n - x mismatches out of n are compatible with the given error rate"""
    x = np.atleast_1d(np.asarray(x, dtype=float))
    n = np.atleast_1d(np.asarray(n)).astype(np.int64)
    return _device().binom_identity(x, n, error_rate, pthres).astype(int)


def matchGTsAccs(sampleWei, t1001snps, skip_hets_db=False):
    """(score, ninfo) per accession column of ``t1001snps`` (int8 [n, n_acc]) under the sample weights
    ``sampleWei`` (float [n, 3]: ref, het, alt).  Runs on the GPU in the reference's summation order, so the
    fp64 scores carry the reference's bits.  The inputs are not modified."""
    sampleWei = np.asarray(sampleWei)
    t1001snps = np.asarray(t1001snps)
    assert sampleWei.shape[0] == t1001snps.shape[0], "please provide same number of positions for both sample and db"
    assert sampleWei.shape[1] == 3, "SNP weights should be a np.array with  shape == n,3"
    return _device().score_dense(sampleWei, t1001snps, skip_hets_db)


# ----------------------------------------------------------------------------- result object
class GenotyperOutput(object):
    """Per-accession result of one scoring run.  ``scores`` are truncated to integers on construction, as
    the reference does; likelihoods are functions of (scores, ninfo) only."""

    def __init__(self, AccList, ScoreList, NumInfoSites, overlap, NumMatSNPs, DPmean):
        self.accs = np.array(AccList, dtype="str")
        self.scores = np.array(ScoreList, dtype="int")
        self.ninfo = np.array(NumInfoSites, dtype="int")
        self.overlap = overlap
        self.num_snps = NumMatSNPs
        self.dp = DPmean

    def get_probabilities(self):
        self.probabilies = _report.ratio_or_nan(self.scores, self.ninfo)      # attribute name as in the reference

    @staticmethod
    def calculate_likelihoods(scores, ninfo, amin="calc"):
        fixed_top = None if (isinstance(amin, str) and amin == "calc") else float(amin)
        return _device().likelihood(np.asarray(scores, dtype=float), np.asarray(ninfo).astype(np.int64),
                                    truncate=False, amin=fixed_top)

    def get_likelihoods(self, amin="calc"):
        # Genotyper's one-call path (snpm_genotype_once) brings the likelihoods of these very counts back with them: they are
        # used as long as scores / ninfo still are the arrays they were computed from
        pre = getattr(self, "_device_likelihoods", None)
        if pre is not None and isinstance(amin, str) and amin == "calc" and np.array_equal(pre[0], self.scores) \
                and np.array_equal(pre[1], self.ninfo):
            self.likelis, self.lrts = pre[2], pre[3]
            return
        self.likelis, self.lrts = self.calculate_likelihoods(self.scores, self.ninfo, amin)

    def _refresh(self):
        self.get_likelihoods()
        self.get_probabilities()

    def print_out_table(self, outFile, _frame=True):
        """``*.scores.txt`` + the table as a DataFrame (:122-138).  ``_frame=False`` (the package's own callers, which drop the
        return value): no DataFrame is built -- an `inbred` run then never imports pandas."""
        self._refresh()
        dp_mean = _report.mean_depth(self.dp)        # once: a pass over the sample's depth column
        if outFile:
            _report.write_scores_table(outFile, self.accs, self.scores, self.ninfo, self.probabilies, self.likelis, self.lrts,
                                       self.num_snps, dp_mean)
        if not _frame:
            return None
        return _report.scores_frame(self.accs, self.scores, self.ninfo, self.probabilies, self.likelis, self.lrts,
                                    self.num_snps, dp_mean)

    def print_json_output(self, outFile):
        self._refresh()
        _report.dump_json(_report.matches_summary(self.accs, self.probabilies, self.ninfo, self.lrts, self.overlap,
                                                  self.num_snps, lr_thres, prob_thres), outFile)

    def case_interpreter(self, topHits):
        topHits = np.asarray(topHits, dtype=int)
        mean_top = np.nanmean(self.probabilies[topHits]) if len(topHits) else np.nan
        return _report.inbred_case(len(topHits), mean_top, self.overlap, prob_thres)


# ----------------------------------------------------------------------------- driver
class Genotyper(object):
    """Score one parsed sample (``inputs``) against a DB (``g``) and write ``<outFile>.scores.txt`` /
    ``<outFile>.matches.json``; ``filter_tophits`` adds the ``--refine`` pass."""

    def __init__(self, inputs, g, outFile, run_genotyper=True, skip_db_hets=False, chunk_size=1000):
        assert type(g) is snp_genotype.Genotype, "provide a snp_genotype.Genotype class for genotypes"
        inputs.filter_chr_names()
        job = dist.job()             # accession-sharded run: rank 0 writes, the other ranks use a scratch prefix
        self.inputs, self.g, self.outFile = inputs, g, (job.output_prefix(outFile) if job else outFile)
        self.chunk_size = chunk_size
        self._skip_db_hets = skip_db_hets
        self.num_lines = len(g.g.accessions)
        if run_genotyper:
            self.result = self.genotyper()
            self.write_genotyper_output(self.result)

    def get_common_positions(self):
        self.commonSNPs = self.g.get_positions_idxs(self.inputs.chrs, self.inputs.pos, _parsed=self.inputs)

    def genotyper(self, filter_pos_ix=None, mask_acc_ix=None, _filter_mask=None):
        """One pass over the matched SNPs.  ``filter_pos_ix``: restrict to these DB rows;
        ``mask_acc_ix``: leave these accessions out of the returned result.
        (``_filter_mask``: the same restriction as a flag per DB row, ``filter_tophits``' shortcut.)"""
        self.get_common_positions()
        db_rows, sample_rows = self.commonSNPs
        if filter_pos_ix is not None or _filter_mask is not None:
            if _filter_mask is not None:
                sel = np.flatnonzero(np.asarray(_filter_mask)[db_rows])
            else:
                assert type(filter_pos_ix) is np.ndarray, "provide np array for indices to be considered"
                sel = np.flatnonzero(np.isin(db_rows, filter_pos_ix))
            if len(sel) < 100:
                log.info("#positions in segregating sites are are too little: %s" % len(sel))
            db_rows, sample_rows = db_rows[sel], sample_rows[sel]
            self.commonSNPs = (db_rows, sample_rows)
        n_matched = len(db_rows)
        # the reference walks the matched SNPs in chunk_size-row matchGTsAccs calls; here that is one query
        # against the HBM-resident panel whose counts are certified identical to that loop's
        panel = self.g.panel()
        job = dist.job()
        once = None
        if type(panel) is engine.Panel and job is None:
            # the whole matrix resident on one GPU: ONE library call gathers the matched weight rows (:221), scores them and returns
            # counts and likelihoods together (snpm_genotype_once)
            coded = self.inputs.weight_codes() if hasattr(self.inputs, "weight_codes") else None
            if coded is not None and panel.n_snp < 2 ** 31:         # a parsed VCF: 6 instead of 24 bytes of weights per SNP to the GPU
                once = panel.genotype_once(db_rows, coded[0], sample_rows, self.chunk_size, self._skip_db_hets, engine.MODE_EXACT,
                                           table=coded[1])
            else:
                once = panel.genotype_once(db_rows, self.inputs.wei, sample_rows, self.chunk_size, self._skip_db_hets, engine.MODE_EXACT)
            scores, ninfo = once["score"], once["ninfo"]
        else:
            query = panel.query(db_rows, self.inputs.wei[sample_rows, ])
            scores, ninfo = query.run(self.chunk_size, self._skip_db_hets, engine.MODE_EXACT)
            query.free()
        if job is not None:          # this rank scored its accession shard: one all-gather makes the vectors whole
            scores, ninfo = job.gather_scores(scores, ninfo, self.num_lines)
        log.info("Done analysing %s positions", n_matched)
        overlap = get_fraction(n_matched, len(self.inputs.pos))
        accs = self.g.g.accessions
        if mask_acc_ix is not None:
            assert type(mask_acc_ix) is np.ndarray, "provide a numpy array of accessions indices to mask"
            shown = np.setdiff1d(np.arange(self.num_lines), mask_acc_ix)
            accs, scores, ninfo = accs[shown], scores[shown], ninfo[shown]
            once = None              # the minimum of another accession set: GenotyperOutput asks the device again
        out = GenotyperOutput(accs, scores, ninfo, overlap, n_matched, self.inputs.dp)
        if once is not None:
            out._device_likelihoods = (out.scores.copy(), out.ninfo.copy(), once["lik"], once["lrt"])
        return out

    def write_genotyper_output(self, result):
        log.info("writing score file!")
        result.print_out_table(self.outFile + '.scores.txt', _frame=False)
        result.print_json_output(self.outFile + ".matches.json")
        codes = self.inputs.gt_codes_of(self.commonSNPs[1]) if hasattr(self.inputs, "gt_codes_of") else None
        getHeterozygosity(self.inputs.gt[self.commonSNPs[1]] if codes is None else codes, self.outFile + ".matches.json", _codes=codes)
        return result

    def filter_tophits(self):
        """--refine: when several (but fewer than half of the) accessions are indistinguishable, rescore
        them on the SNPs that segregate among them and write ``<outFile>.refined.scores.txt``."""
        self.result = self.genotyper()
        self.write_genotyper_output(self.result)
        top = np.flatnonzero(self.result.lrts < lr_thres)
        if len(top) == 1:
            log.info("Done! It is a perfect hit")
            return None
        log.info("#lines indistinguishable: %s" % len(top))
        if len(top) > (self.num_lines / 2):
            log.info("too many lines are indistinguishable, skipping refining likelihoods step")
            return None
        log.info("refining likelihoods for only indistinguishable lines")
        others = np.flatnonzero(self.result.lrts >= lr_thres)
        if hasattr(self.g, "segregating_mask"):
            self.result_fine = self.genotyper(mask_acc_ix=others, _filter_mask=self.g.segregating_mask(top))
        else:
            self.result_fine = self.genotyper(filter_pos_ix=self.g.identify_segregating_snps(top), mask_acc_ix=others)
        log.info("writing output: %s" % self.outFile + ".refined.scores.txt")
        self.result_fine.print_out_table(self.outFile + ".refined.scores.txt", _frame=False)


def getHeterozygosity(snpGT, outFile='default', _codes=None):
    """fraction of heterozygous calls among ``snpGT``; also recorded in the JSON file when one is given
    (``_codes``: the calls already as ``parseGT`` codes)"""
    n_het = int(np.count_nonzero((parsers.parseGT(snpGT) if _codes is None else _codes) == 2))
    het = get_fraction(n_het, len(snpGT))
    if outFile != 'default':
        _report.update_json(outFile, percent_heterozygosity=het)
    return het


def genotype_batch(inputs_list, g, out_files, skip_db_hets=False, chunk_size=1000):
    """``Genotyper`` for many samples against one DB in ONE device call (the reference starts a process per sample,
    :256-268): every sample's positions are intersected with the DB on the host, all of them are scored by one segmented
    launch with the per-sample certificate (``engine.score_batch``), and each sample gets the files ``Genotyper`` writes
    (``<out>.scores.txt``, ``<out>.matches.json``).  Returns the list of ``GenotyperOutput``."""
    assert type(g) is snp_genotype.Genotype, "provide a snp_genotype.Genotype class for genotypes"
    assert len(inputs_list) == len(out_files)
    samples, common = [], []
    for inputs in inputs_list:
        inputs.filter_chr_names()
        db_rows, sample_rows = g.get_positions_idxs(inputs.chrs, inputs.pos, _parsed=inputs)
        common.append((db_rows, sample_rows))
        samples.append((db_rows, inputs.wei[sample_rows, ]))
    res = engine.score_batch(g.panel(), samples, chunk_size, skip_db_hets, engine.MODE_EXACT, likelihoods=False)
    job = dist.job()
    if job is not None:              # this rank scored its accession shard: B rows gathered like window rows; rank 0 writes
        res["score"], res["ninfo"] = job.gather_windows(res["score"], res["ninfo"], len(g.g.accessions))
        out_files = [job.output_prefix(f) for f in out_files]
    results = []
    for b, inputs in enumerate(inputs_list):
        n_matched = len(common[b][0])
        out = GenotyperOutput(g.g.accessions, res["score"][b], res["ninfo"][b], get_fraction(n_matched, len(inputs.pos)),
                              n_matched, inputs.dp)
        out.print_out_table(out_files[b] + '.scores.txt', _frame=False)
        out.print_json_output(out_files[b] + ".matches.json")
        codes = inputs.gt_codes_of(common[b][1]) if hasattr(inputs, "gt_codes_of") else None
        getHeterozygosity(inputs.gt[common[b][1]] if codes is None else codes, out_files[b] + ".matches.json", _codes=codes)
        results.append(out)
    return results


def potatoGenotyperBatch(args):
    """entry point of ``snpmatch_amd inbred-batch`` (args: inFiles, hdf5File, hdf5accFile, outFile, logDebug, skip_db_hets,
    batchSize): what ``potatoGenotyper`` writes for every sample, with the DB loaded once and ``batchSize`` samples per device
    call.  Sample ``<dir>/<name>.<ext>`` writes ``<outFile>.<name>.scores.txt`` / ``.matches.json``."""
    log.info("loading database files")
    g = snp_genotype.Genotype(args['hdf5File'], args['hdf5accFile'])
    files = list(args['inFiles'])
    names = []
    for f in files:
        base = os.path.basename(f)
        for ext in (".vcf.gz", ".vcf", ".bed", ".npz"):
            if base.endswith(ext):
                base = base[:-len(ext)]
                break
        names.append(base)
    assert len(set(names)) == len(names), "sample file names must be distinct (they name the outputs)"
    step = max(1, int(args.get('batchSize', 64) or 64))
    for b0 in range(0, len(files), step):
        inputs = [parse_inputs_once(f, args['logDebug']) for f in files[b0:b0 + step]]
        log.info("scoring samples %d .. %d of %d", b0 + 1, b0 + len(inputs), len(files))
        genotype_batch(inputs, g, ["%s.%s" % (args['outFile'], n) for n in names[b0:b0 + step]], skip_db_hets=args['skip_db_hets'])
    log.info("finished!")


def parse_inputs_once(in_file, log_debug):
    """ParseInputs writes a cache next to the input: in an accession-sharded job rank 0 parses, the others load its cache"""
    job = dist.job()
    if job is None:
        return parsers.ParseInputs(inFile=in_file, logDebug=log_debug)
    inputs = parsers.ParseInputs(inFile=in_file, logDebug=log_debug) if job.is_writer else None
    if inputs is not None:
        inputs.wait_for_cache()      # the cache is written in the background: complete before the other ranks look for it
    job.barrier()
    return inputs if inputs is not None else parsers.ParseInputs(inFile=in_file, logDebug=log_debug)


def potatoGenotyper(args):
    """entry point of ``snpmatch inbred`` (args: inFile, hdf5File, hdf5accFile, outFile, logDebug, refine, skip_db_hets)"""
    inputs = parse_inputs_once(args['inFile'], args['logDebug'])
    log.info("loading database files")
    g = snp_genotype.Genotype(args['hdf5File'], args['hdf5accFile'])
    log.info("running genotyper!")
    job = Genotyper(inputs, g, args['outFile'], run_genotyper=not args['refine'], skip_db_hets=args['skip_db_hets'])
    if args['refine']:
        job.filter_tophits()
    log.info("finished!")
