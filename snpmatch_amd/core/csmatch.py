"""
SNPmatch ``cross`` on MI355X: host-side mirror of the reference's ``snpmatch.core.csmatch``
(core/csmatch.py:16-200): ``CrossIdentifier`` (``cross_identifier``, ``get_window_data``,
``window_genotyper``, ``match_insilico_f1s``, ``cross_interpreter``), ``convert_int64``,
``potatoCrossIdentifier``, ``chunk_size``.

The window loop of ``window_genotyper`` (one ``matchGTsAccs`` call per genomic window,
core/csmatch.py:80-90) is ONE segmented device query: the matched DB rows of all windows are scored by
``k_strict`` with one segment per window (reference summation order, fp64 bit-exact), per-window
likelihoods / nanmin / ratios by ``k_likelihood`` with one row per window, the binomial identity test
by the library.  Tables and JSON are host glue in the reference's formats.
"""
import itertools
import json
import logging

import numpy as np
import pandas as pd

from . import genomes
from . import parsers
from . import snp_genotype
from . import snpmatch
from .. import engine

log = logging.getLogger(__name__)
chunk_size = 1000

_WINDOW_COLUMNS = ["acc", "snps_match", "snps_info", "score", "likelihood", "identical", "num_amb", "window_index"]


def _fstr(a):
    """numpy's float -> str conversion, as np.column_stack applies when stacking with strings
    (core/csmatch.py:50): shortest round-trip repr."""
    return np.asarray(a, dtype=float).astype("U32")


class CrossIdentifier(object):

    def __init__(self, inputs, g, genome_id, binLen, output_id="cross.identifier", run_identifier=True,
                 identity_error_rate=0.02, skip_db_hets=False):
        self.g = g
        assert type(inputs) is parsers.ParseInputs, "provide a parsers class"
        inputs.filter_chr_names()
        self.inputs = inputs
        self.genome = genomes.Genome(genome_id)
        self.binLen = binLen
        self.output_id = output_id
        self.error_rate = identity_error_rate
        self._skip_db_hets = skip_db_hets
        if run_identifier:
            self.cross_identifier()

    def cross_identifier(self):
        window_snpmatch_result = self.window_genotyper(self.output_id + '.windowscore.txt')
        window_snpmatch_result.print_json_output(self.output_id + ".scores.txt.matches.json")
        snpmatch.getHeterozygosity(self.inputs.gt[window_snpmatch_result.matchedTarInd],
                                   self.output_id + ".scores.txt.matches.json")
        with open(self.output_id + ".scores.txt.matches.json") as json_out:
            self.cross_identfier_json = json.load(json_out)
        self.result = self.match_insilico_f1s(window_snpmatch_result, self.output_id + '.scores.txt')
        self.cross_interpreter(self.output_id + ".matches.json")

    # ------------------------------------------------------------------ one window (API parity)
    @staticmethod
    def get_window_data(bin_inds, AccList, ScoreList, NumInfoSites, error_rate=0.02):
        """rows of the window table for one window (core/csmatch.py:44-61)."""
        ScoreList = np.asarray(ScoreList, dtype=float)
        NumInfoSites = np.asarray(NumInfoSites)
        lik, lrt = snpmatch.GenotyperOutput.calculate_likelihoods(ScoreList, NumInfoSites)
        identity = snpmatch.np_test_identity(x=ScoreList, n=NumInfoSites, error_rate=error_rate)
        return _window_frame(bin_inds, np.asarray(AccList), ScoreList, NumInfoSites, lik, lrt, identity)

    # ------------------------------------------------------------------ all windows
    def window_genotyper(self, out_file, mask_acc_ix=None):
        num_lines = len(self.g.accessions)
        if mask_acc_ix is not None:
            assert type(mask_acc_ix) is np.ndarray, "please provide numpy array of acc indices to be masked"
            keep = np.setdiff1d(np.arange(num_lines), mask_acc_ix)
        else:
            keep = np.arange(num_lines)
        # window segmentation of the DB and of the sample, then per-window intersection (core/csmatch.py:75-84)
        db_pos = np.asarray(self.g.g.positions)
        rows_db, rows_s, win_off, win_chr = [], [], [0], []
        for e_g, e_s in zip(self.genome.get_bins_genome(self.g.g, self.binLen),
                            self.genome.get_bins_arrays(self.inputs.chrs, self.inputs.pos, self.binLen)):
            gi = np.array(e_g[2], dtype=int)
            si = np.array(e_s[2], dtype=int)
            g_bin_pos = db_pos[gi]
            s_bin_pos = self.inputs.pos[si]
            rows_db.append(gi[np.isin(g_bin_pos, s_bin_pos)])
            rows_s.append(si[np.isin(s_bin_pos, g_bin_pos)])
            win_off.append(win_off[-1] + len(rows_db[-1]))
            win_chr.append(e_g[0])
        n_win = len(win_chr)
        rows_db = np.concatenate(rows_db) if n_win else np.zeros(0, dtype=int)
        rows_s = np.concatenate(rows_s) if n_win else np.zeros(0, dtype=int)
        win_off = np.array(win_off, dtype=np.int64)
        NumMatSNPs = int(win_off[-1])

        # segmented scoring on the device: [n_win, n_acc] scores / informative sites + running totals
        query = engine.Query(self.g.panel(), rows_db, self.inputs.wei[rows_s, ])
        w_score, w_ninfo, tot_score, tot_ninfo = query.run_windows(win_off, self._skip_db_hets)
        query.free()

        self.windows_data = pd.DataFrame(columns=_WINDOW_COLUMNS)
        nonempty = np.where(np.diff(win_off) > 0)[0]
        if len(nonempty) > 0:
            ctx = engine.default_context()
            sc = np.ascontiguousarray(w_score[nonempty][:, keep])
            ni = np.ascontiguousarray(w_ninfo[nonempty][:, keep])
            lik, lrt = ctx.likelihood(sc, ni)                                   # one device row per window
            ident = ctx.binom_identity(sc.ravel(), ni.ravel(), self.error_rate, 0.05).reshape(sc.shape)
            accs = np.asarray(self.g.accessions)[keep]
            frames = []
            for k, w in enumerate(nonempty):
                f = _window_frame(int(w) + 1, accs, sc[k], ni[k], lik[k], lrt[k], ident[k])
                if len(f) > 0:
                    frames.append(f)
                if (w + 1) % 50 == 0:
                    log.info("Done analysing %s positions", int(win_off[w + 1]))
            if frames:
                self.windows_data = pd.concat(frames, ignore_index=True)

        winds_chrs = np.asarray(self.genome.chrs_ids)[np.array(win_chr, dtype=int)] if n_win else np.zeros(0, dtype="U1")
        overlap = snpmatch.get_fraction(NumMatSNPs, len(self.inputs.pos))
        result = snpmatch.GenotyperOutput(np.asarray(self.g.accessions)[keep], tot_score[keep], tot_ninfo[keep], overlap,
                                          NumMatSNPs, self.inputs.dp)
        result.matchedTarInd = rows_s
        result.winds_chrs = winds_chrs
        if out_file is not None:
            self.windows_data.to_csv(out_file, sep="\t", index=False)
            return result
        return [self.windows_data, result]

    # ------------------------------------------------------------------ in-silico F1s of the top hits
    def match_insilico_f1s(self, snpmatch_result, out_file):
        """score the 45 pairwise F1s of the ten most probable accessions (core/csmatch.py:106-129)."""
        assert type(snpmatch_result) is snpmatch.GenotyperOutput, "Please provide GenotyperOutput class as input"
        if not hasattr(snpmatch_result, 'probabilies'):
            snpmatch_result.get_probabilities()
        log.info("simulating F1s for top 10 accessions")
        TopHitAccs = np.argsort(-snpmatch_result.probabilies)[0:10]
        commonSNPs = self.g.get_positions_idxs(self.inputs.chrs, self.inputs.pos)
        wei = self.inputs.wei[commonSNPs[1], ]
        cols = {}
        for i in TopHitAccs:
            cols[i] = np.asarray(self.g.g_acc.snps[:, i])[commonSNPs[0]]
        for (i, j) in itertools.combinations(TopHitAccs, 2):
            gtp1, gtp2 = cols[i], cols[j]
            homalt = np.where((gtp1 == 1) & (gtp2 == 1))[0]
            homref = np.where((gtp1 == 0) & (gtp2 == 0))[0]
            het = np.where((gtp1 != -1) & (gtp2 != -1) & (gtp1 != gtp2))[0]
            score = np.sum(wei[homalt, 2]) + np.sum(wei[homref, 0]) + np.sum(wei[het, 1])
            numinfo = len(homalt) + len(homref) + len(het)
            snpmatch_result.scores = np.append(snpmatch_result.scores, score)
            snpmatch_result.ninfo = np.append(snpmatch_result.ninfo, numinfo)
            snpmatch_result.accs = np.append(snpmatch_result.accs, self.g.accessions[i] + "x" + self.g.accessions[j])
        if out_file is not None:
            snpmatch_result.print_out_table(out_file)
        return snpmatch_result

    # ------------------------------------------------------------------ interpretation
    def cross_interpreter(self, out_file):
        """F1 / F2 / contamination call from the window table (core/csmatch.py:131-186)."""
        assert 'cross_identfier_json' in dir(self), "run cross identifier first!"
        assert 'windows_data' in dir(self), "run window genotyper first!"
        log.info("running cross interpreter!")
        if self.cross_identfier_json['interpretation']['case'] < 3:
            return
        wd = self.windows_data
        out = self.cross_identfier_json
        identical_wind = np.where(wd.groupby('window_index').max()['identical'] == 1)[0]
        num_winds = np.unique(wd['window_index']).shape[0]
        out['identical_windows'] = [snpmatch.get_fraction(identical_wind.shape[0], num_winds), num_winds]
        acc_col = wd.iloc[:, 0]
        amb_col = np.asarray(wd.iloc[:, 6])
        win_col = np.asarray(wd.iloc[:, 7])
        homo_wind = np.intersect1d(wd['window_index'][np.where(wd['num_amb'] < 20)[0]], identical_wind)
        homo_acc = np.unique(acc_col[np.where(np.isin(win_col, homo_wind))[0]], return_counts=True)
        out['matches'] = [(homo_acc[0][i], int(homo_acc[1][i])) for i in np.argsort(-homo_acc[1])]
        topMatch = np.argsort(self.result.likelis)[0]            # best likelihood, in-silico F1s included
        if topMatch in np.where(~np.isin(self.result.accs, self.g.accessions))[0]:
            mother, father = self.result.accs[topMatch].split("x")[0], self.result.accs[topMatch].split("x")[1]
            out['interpretation']['text'] = "Sample may be a F1! or a contamination!"
            out['interpretation']['case'] = 5
            out['parents'] = {'mother': [mother, 1], 'father': [father, 1]}
            out['genotype_windows'] = {'chr_bins': None, 'coordinates': {'x': None, 'y': None}}
        else:
            clean = np.unique(acc_col[np.where(amb_col == 1)[0]], return_counts=True)      # unambiguous windows
            if len(clean[0]) > 0:
                order = np.argsort(-clean[1])[0:2]
                parents = clean[0][order].astype("str")
                parents_counts = clean[1][order].astype("int")
                xdict = np.array(np.unique(win_col), dtype="int")
                ydict = np.repeat("NA", len(xdict)).astype("S25")
                acc_str = np.asarray(acc_col.astype("str"))
                in_homo = np.isin(win_col, homo_wind)
                out['interpretation']['case'] = 6
                if len(parents) == 1:
                    out['interpretation']['text'] = "Sample may be a F2! but only one parent found!"
                    out['parents'] = {'mother': [parents[0], parents_counts[0]], 'father': ["NA", "NA"]}
                    par1_ind = win_col[np.where((acc_str == parents[0]) & in_homo)[0]]
                    ydict[np.where(np.isin(xdict, par1_ind))[0]] = parents[0]
                    chr_bins = None
                else:
                    out['interpretation']['text'] = "Sample may be a F2!"
                    out['parents'] = {'mother': [parents[0], parents_counts[0]], 'father': [parents[1], parents_counts[1]]}
                    NumChrs = np.unique(self.result.winds_chrs, return_counts=True)
                    chr_bins = dict((NumChrs[0][i], NumChrs[1][i]) for i in range(len(NumChrs[0])))
                    par1_ind = win_col[np.where((acc_str == parents[0]) & in_homo)[0]]
                    par2_ind = win_col[np.where((acc_str == parents[1]) & in_homo)[0]]
                    ydict[np.where(np.isin(xdict, par1_ind))[0]] = parents[0]
                    ydict[np.where(np.isin(xdict, par2_ind))[0]] = parents[1]
                out['genotype_windows'] = {'chr_bins': chr_bins, 'coordinates': {'x': xdict.tolist(), 'y': ydict.tolist()}}
            else:
                out['interpretation']['case'] = 7
                out['interpretation']['text'] = "Sample may just be contamination!"
                out['genotype_windows'] = {'chr_bins': None, 'coordinates': {'x': None, 'y': None}}
                out['parents'] = {'mother': [None, 0], 'father': [None, 1]}
        with open(out_file, "w") as out_stats:
            out_stats.write(json.dumps(out, sort_keys=True, indent=4, default=convert_int64))


def _window_frame(bin_inds, accs, scores, ninfo, lik, lrt, identity):
    """the reference's per-window frame (core/csmatch.py:49-61): rows of the accessions whose ratio to
    the window's best likelihood is below lr_thres, kept only when 1 <= #rows < #accessions.
    'score' and 'likelihood' are strings (the reference stacks them with the accession names)."""
    num_lines = len(accs)
    amb = np.where(lrt < snpmatch.lr_thres)[0]
    if not (1 <= len(amb) < num_lines):
        return pd.DataFrame(columns=_WINDOW_COLUMNS)
    with np.errstate(divide='ignore', invalid='ignore'):
        frac = np.where(ninfo[amb] > 0, scores[amb] / np.where(ninfo[amb] > 0, ninfo[amb], 1), np.nan)
    f = pd.DataFrame({
        "acc": np.asarray(accs)[amb].astype(str),
        "snps_match": np.asarray(scores[amb], dtype=float).astype(int),
        "snps_info": np.asarray(ninfo[amb], dtype=float).astype(int),
        "score": _fstr(frac),
        "likelihood": _fstr(lik[amb]),
        "identical": np.asarray(identity[amb], dtype=float),
        "num_amb": len(amb),
        "window_index": bin_inds,
    }, columns=_WINDOW_COLUMNS)
    f["score"] = f["score"].astype(object)
    f["likelihood"] = f["likelihood"].astype(object)
    return f


def convert_int64(o):
    if isinstance(o, np.int64):
        return int(o)


def potatoCrossIdentifier(args):
    inputs = parsers.ParseInputs(inFile=args['inFile'], logDebug=args['logDebug'])
    log.info("loading genotype files!")
    g = snp_genotype.Genotype(args['hdf5File'], args['hdf5accFile'])
    log.info("done!")
    log.info("running cross identifier!")
    CrossIdentifier(inputs, g, args['genome'], args['binLen'], args['outFile'], run_identifier=True,
                    skip_db_hets=args['skip_db_hets'])
    log.info("finished!")
