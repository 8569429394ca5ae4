"""
SNPmatch ``inbred`` on MI355X: host-side mirror of the reference's ``snpmatch.core.snpmatch``
(core/snpmatch.py:17-268) over libsnpmatch_hip (include/snpmatch_hip.h).

Same names, argument meaning and error behaviour as the reference for this path:
``matchGTsAccs``, ``likeliTest``, ``get_fraction`` / ``np_get_fraction``, ``np_binom_test`` /
``np_test_identity``, ``GenotyperOutput``, ``Genotyper``, ``getHeterozygosity``,
``potatoGenotyper`` and the module constants ``lr_thres``, ``snp_thres``, ``prob_thres``.

What runs where
  * scoring (``matchGTsAccs``, the 1000-SNP chunk loop of ``Genotyper.genotyper``), the likelihoods
    (``likeliTest`` over accessions, nanmin, ratio) and the binomial identity test run in HIP kernels /
    the C-ABI library; there is no CPU implementation of them in this package.
  * argument checks, tables and JSON files are host glue with the reference's formats.
"""
import json
import logging
import sys

import numpy as np
import pandas as pd

from . import parsers
from . import snp_genotype
from .. import engine

log = logging.getLogger(__name__)
lr_thres = 3.841
snp_thres = 4000
prob_thres = 0.98


def die(msg):
    sys.stderr.write('Error: ' + msg + '\n')
    sys.exit(1)


def get_fraction(x, y, y_min=0):
    if y <= y_min:
        return np.nan
    return float(x) / y


np_get_fraction = np.vectorize(get_fraction, excluded="y_min")


def likeliTest(n, y):
    """n informative sites, y matched sites -> likelihood (core/snpmatch.py:40-55); evaluated by the
    device kernel k_likelihood like every other likelihood of this package."""
    assert y <= n, "provided y is greater than n"
    if n == 0:
        return np.nan
    if y == n:
        return 1
    if y > 0:
        lik, _ = engine.default_context().likelihood(np.array([float(y)]), np.array([int(n)]))
        return float(lik[0])
    elif y == 0:
        return np.nan


def np_binom_test(x, n, p, alternative=None):
    """survival / distribution function of the binomial (core/snpmatch.py:57-68)."""
    ctx = engine.default_context()
    x = np.atleast_1d(np.asarray(x, dtype=float))
    n = np.atleast_1d(np.asarray(n))
    if alternative in ['larger', 'greater']:
        # binom.sf(x - 1, n, p); the library computes sf((n - x') - 1) for x' = n - x
        _, sf = ctx.binom_identity(n - x, n.astype(np.int64), p, 0.0, return_sf=True)
        return sf
    elif alternative in ['smaller', 'less']:
        _, sf = ctx.binom_identity(n - x - 1, n.astype(np.int64), p, 0.0, return_sf=True)
        return 1.0 - sf
    raise NotImplementedError("two-sided binomial test is not on the inbred / cross path")


def np_test_identity(x, n, error_rate=0.0005, pthres=0.05):
    """1 where the mismatches n - x are compatible with `error_rate` (core/snpmatch.py:70-72)."""
    x = np.atleast_1d(np.asarray(x, dtype=float))
    n = np.atleast_1d(np.asarray(n)).astype(np.int64)
    return engine.default_context().binom_identity(x, n, error_rate, pthres).astype(int)


def matchGTsAccs(sampleWei, t1001snps, skip_hets_db=False):
    """score[a] = sum_s W[s, category(db[s,a])], ninfo[a] = #non-missing (core/snpmatch.py:74-89),
    computed on the GPU in the reference's summation order (fp64 bit-exact)."""
    sampleWei = np.asarray(sampleWei)
    t1001snps = np.asarray(t1001snps)
    assert sampleWei.shape[0] == t1001snps.shape[0], "please provide same number of positions for both sample and db"
    assert sampleWei.shape[1] == 3, "SNP weights should be a np.array with  shape == n,3"
    return engine.default_context().score_dense(sampleWei, t1001snps, skip_hets_db)


class GenotyperOutput(object):
    """core/snpmatch.py:91-168."""

    def __init__(self, AccList, ScoreList, NumInfoSites, overlap, NumMatSNPs, DPmean):
        self.accs = np.array(AccList, dtype="str")
        self.scores = np.array(ScoreList, dtype="int")
        self.ninfo = np.array(NumInfoSites, dtype="int")
        self.overlap = overlap
        self.num_snps = NumMatSNPs
        self.dp = DPmean

    def get_probabilities(self):
        with np.errstate(divide='ignore', invalid='ignore'):
            probs = np.where(self.ninfo > 0, self.scores / np.where(self.ninfo > 0, self.ninfo, 1).astype(float), np.nan)
        self.probabilies = np.array(probs, dtype="float")

    @staticmethod
    def calculate_likelihoods(scores, ninfo, amin="calc"):
        """(likelihoods, likelihood ratios to the top hit) on the device; core/snpmatch.py:106-117."""
        scores = np.asarray(scores, dtype=float)
        ninfo = np.asarray(ninfo)
        a = None if (isinstance(amin, str) and amin == "calc") else float(amin)
        lik, lrt = engine.default_context().likelihood(scores, ninfo.astype(np.int64), truncate=False, amin=a)
        return (lik, lrt)

    def get_likelihoods(self, amin="calc"):
        (self.likelis, self.lrts) = self.calculate_likelihoods(self.scores, self.ninfo, amin)

    def print_out_table(self, outFile):
        self.get_likelihoods()
        self.get_probabilities()
        output_table = pd.DataFrame({
            'accs': self.accs,
            'matches': self.scores,
            'ninfo': self.ninfo,
            'probabilities': self.probabilies,
            'likelihood': self.likelis,
            'lrt': self.lrts,
            'num_snps': self.num_snps,
            'dp': parsers._nanmean_depth(self.dp)
        })
        output_table = output_table[['accs', 'matches', 'ninfo', 'probabilities', 'likelihood', 'lrt', 'num_snps', 'dp']]
        if outFile:
            output_table.to_csv(outFile, header=None, sep="\t", index=None)
        return output_table

    def print_json_output(self, outFile):
        self.get_likelihoods()
        self.get_probabilities()
        topHits = np.where(self.lrts < lr_thres)[0]
        overlapScore = [get_fraction(self.ninfo[i], self.num_snps) for i in range(len(self.accs))]
        sorted_order = topHits[np.argsort(-self.probabilies[topHits])]
        (case, note) = self.case_interpreter(topHits)
        matches_dict = [(str(self.accs[i]), float(self.probabilies[i]), int(self.ninfo[i]), float(overlapScore[i]))
                        for i in sorted_order]
        topHitsDict = {'overlap': [self.overlap, self.num_snps], 'matches': matches_dict,
                       'interpretation': {'case': case, 'text': note}}
        with open(outFile, "w") as out_stats:
            out_stats.write(json.dumps(topHitsDict, sort_keys=True, indent=4))

    def case_interpreter(self, topHits):
        overlap_thres = 0.5
        case = 1
        note = "Ambiguous sample"
        if len(topHits) == 1:
            case = 0
            note = "Unique hit"
        elif np.nanmean(self.probabilies[topHits]) > prob_thres:
            case = 2
            note = "Ambiguous sample: Accessions in top hits can be really close"
        elif self.overlap > overlap_thres:
            case = 3
            note = "Ambiguous sample: Sample might contain mixture of DNA or contamination"
        elif self.overlap < overlap_thres:
            case = 4
            note = "Ambiguous sample: Many input SNP positions are missing in db positions. Maybe sample  not one in database"
        return (case, note)


class Genotyper(object):
    """core/snpmatch.py:170-241; the chunk loop runs as one device query per call of genotyper()."""

    def __init__(self, inputs, g, outFile, run_genotyper=True, skip_db_hets=False, chunk_size=1000):
        assert type(g) is snp_genotype.Genotype, "provide a snp_genotype.Genotype class for genotypes"
        inputs.filter_chr_names()
        self.chunk_size = chunk_size
        self.inputs = inputs
        self.g = g
        self.num_lines = len(self.g.g.accessions)
        self.outFile = outFile
        self._skip_db_hets = skip_db_hets
        if run_genotyper:
            self.result = self.genotyper()
            self.write_genotyper_output(self.result)

    def get_common_positions(self):
        self.commonSNPs = self.g.get_positions_idxs(self.inputs.chrs, self.inputs.pos)

    def filter_tophits(self):
        self.result = self.genotyper()
        self.write_genotyper_output(self.result)
        self.result.get_likelihoods()
        topHits = np.where(self.result.lrts < lr_thres)[0]
        if len(topHits) == 1:
            log.info("Done! It is a perfect hit")
            return None
        log.info("#lines indistinguishable: %s" % len(topHits))
        log.info("refining likelihoods for only indistinguishable lines")
        if len(topHits) > (self.num_lines / 2):
            log.info("too many lines are indistinguishable, skipping refining likelihoods step")
            return None
        seg_ix = self.g.identify_segregating_snps(topHits)
        self.result_fine = self.genotyper(filter_pos_ix=seg_ix, mask_acc_ix=np.where(self.result.lrts >= lr_thres)[0])
        log.info("writing output: %s" % self.outFile + ".refined.scores.txt")
        self.result_fine.print_out_table(self.outFile + ".refined.scores.txt")

    def genotyper(self, filter_pos_ix=None, mask_acc_ix=None):
        self.get_common_positions()
        if filter_pos_ix is not None:
            assert type(filter_pos_ix) is np.ndarray, "provide np array for indices to be considered"
            t_ix = np.where(np.isin(self.commonSNPs[0], filter_pos_ix))[0]
            if t_ix.shape[0] < 100:
                log.info("#positions in segregating sites are are too little: %s" % t_ix.shape[0])
            self.commonSNPs = (self.commonSNPs[0][t_ix], self.commonSNPs[1][t_ix])
        NumMatSNPs = len(self.commonSNPs[0])
        # ScoreList += matchGTsAccs(chunk) over 1000-SNP chunks (core/snpmatch.py:218-225): one query
        # on the HBM-resident panel; match counts and informative sites are bit-exact with that loop.
        panel = self.g.panel()
        query = engine.Query(panel, self.commonSNPs[0], self.inputs.wei[self.commonSNPs[1], ])
        ScoreList, NumInfoSites = query.run(self.chunk_size, self._skip_db_hets, engine.MODE_EXACT)
        query.free()
        log.info("Done analysing %s positions", NumMatSNPs)
        overlap = get_fraction(NumMatSNPs, len(self.inputs.pos))
        if mask_acc_ix is not None:
            assert type(mask_acc_ix) is np.ndarray, "provide a numpy array of accessions indices to mask"
            keep = np.setdiff1d(np.arange(self.num_lines), mask_acc_ix)
            return GenotyperOutput(self.g.g.accessions[keep], ScoreList[keep], NumInfoSites[keep], overlap, NumMatSNPs,
                                   self.inputs.dp)
        return GenotyperOutput(self.g.g.accessions, ScoreList, NumInfoSites, overlap, NumMatSNPs, self.inputs.dp)

    def write_genotyper_output(self, result):
        log.info("writing score file!")
        result.get_likelihoods()
        result.print_out_table(self.outFile + '.scores.txt')
        result.print_json_output(self.outFile + ".matches.json")
        getHeterozygosity(self.inputs.gt[self.commonSNPs[1]], self.outFile + ".matches.json")
        return result


def getHeterozygosity(snpGT, outFile='default'):
    snpBinary = parsers.parseGT(snpGT)
    numHets = len(np.where(snpBinary == 2)[0])
    if outFile != 'default':
        with open(outFile) as json_out:
            topHitsDict = json.load(json_out)
        topHitsDict['percent_heterozygosity'] = get_fraction(numHets, len(snpGT))
        with open(outFile, "w") as out_stats:
            out_stats.write(json.dumps(topHitsDict, sort_keys=True, indent=4))
    return get_fraction(numHets, len(snpGT))


def potatoGenotyper(args):
    inputs = parsers.ParseInputs(inFile=args['inFile'], logDebug=args['logDebug'])
    log.info("loading database files")
    g = snp_genotype.Genotype(args['hdf5File'], args['hdf5accFile'])
    log.info("done!")
    log.info("running genotyper!")
    if args['refine']:
        genotyper = Genotyper(inputs, g, args['outFile'], run_genotyper=False, skip_db_hets=args['skip_db_hets'])
        genotyper.filter_tophits()
        log.info("finished!")
        return None
    Genotyper(inputs, g, args['outFile'], run_genotyper=True, skip_db_hets=args['skip_db_hets'])
    log.info("finished!")
