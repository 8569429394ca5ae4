#!/usr/bin/env python3
"""
Build-container-only probe (needs /root/reference; not a pytest file): the oracles against the UNMODIFIED reference's matchGTsAccs on
shapes the committed goldens do not hold -- 1 ... 7 accessions, n from 1 to 50 001 (across numpy's 8192-element reduction pieces),
hard-call and PL weights, both skip_hets settings -- fp64 bit patterns and counts.  Round 3's verdict found the one-accession
divergence with a probe like this one; it is kept so that the next shape question takes a minute.

    python tests/golden/probe_oracle_vs_reference.py [seed]   # prints "cases N diffs 0" (matchGTsAccs shapes, then likelihood / LRT / identity on random counts)
                                                              # and the verdict of the same reference under the second interpreter (numpy 1.26)
"""
import os
import sys
import types
import warnings

import numpy as np

if not os.path.isdir("/root/reference"):
    sys.exit("the reference is not present here: nothing to probe")
sys.dont_write_bytecode = True
for _m in ("allel", "h5py", "hmmlearn", "hmmlearn.hmm"):
    sys.modules[_m] = types.ModuleType(_m)
sys.modules["hmmlearn"].hmm = sys.modules["hmmlearn.hmm"]
sys.path.insert(0, "/root/reference")
warnings.filterwarnings("ignore")
from snpmatch.core import snpmatch as ref_sm  # noqa: E402

assert "/root/reference" in ref_sm.__file__, "the reference must be the module imported as snpmatch here"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import c_oracle  # noqa: E402
from oracle import snpmatch_oracle as orc  # noqa: E402


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 123)
bad = tot = 0
for n_acc in (1, 2, 3, 7):
    for n in (1, 2, 7, 8, 9, 127, 128, 129, 1000, 4095, 8191, 8192, 8193, 16384, 16385, 20000, 50001):
        for frac_pl in (0.0, 0.8):
            db = rng.choice(np.array([-1, 0, 1, 2], dtype=np.int8), size=(n, n_acc), p=[0.05, 0.6, 0.33, 0.02])
            wei = np.exp(-rng.integers(0, 256, size=(n, 3)).astype(float) / 10)
            hard = rng.random(n) >= frac_pl
            wei[hard] = np.eye(3)[rng.integers(0, 3, int(hard.sum()))]
            for skip in (False, True):
                rs, rn = ref_sm.matchGTsAccs(wei, db.copy(), skip)
                ps, pn = orc.match_gts_accs(wei, db, skip)
                cs, cn = c_oracle.match(wei, db, skip)
                tot += 1
                same = np.array_equal(bits(rs), bits(ps)) and np.array_equal(bits(rs), bits(cs)) and np.array_equal(rn, pn) and np.array_equal(rn, cn)
                if not same:
                    bad += 1
                    print("DIFF n_acc %d n %d frac_pl %.1f skip %s" % (n_acc, n, frac_pl, skip))
# likelihood / ratio (core/snpmatch.py:40-55, :106-117) and the identity test (:57-72) on random counts, integer and float scores
for _ in range(200):
    m = int(rng.integers(1, 400))
    ninfo = rng.integers(0, 5000, size=m)
    scores = np.minimum(ninfo, rng.integers(0, 5000, size=m)).astype(float)
    if rng.integers(0, 2):
        scores = scores * rng.random(m)                       # window scores are floats
    if rng.integers(0, 3) == 0:
        scores[rng.integers(0, m)] = ninfo[rng.integers(0, m)] = 0
    scores = np.minimum(scores, ninfo)
    rl, rr = ref_sm.GenotyperOutput.calculate_likelihoods(scores, ninfo)
    ol, orr = orc.calculate_likelihoods(scores, ninfo)
    ri = ref_sm.np_test_identity(scores, ninfo, 0.02)
    oi = orc.test_identity(scores, ninfo, 0.02)
    tot += 1
    if not (np.array_equal(bits(rl), bits(ol)) and np.array_equal(bits(rr), bits(orr)) and np.array_equal(np.asarray(ri).astype(int), np.asarray(oi).astype(int))):
        bad += 1
        print("DIFF likelihood / identity, m %d" % m)
print("cases %d diffs %d" % (tot, bad))
# the same reference under the image's second interpreter (numpy 1.26 / scipy 1.7) on the committed G1 / G1b / G4 inputs: the
# committed bit patterns must come out again (tests/golden/probe_reference_second_numpy.py)
PY39 = "/opt/conda/bin/python3.9"
if os.path.exists(PY39):
    import subprocess
    r = subprocess.run([PY39, os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe_reference_second_numpy.py")],
                       capture_output=True, text=True)
    print("second interpreter: " + (r.stdout.strip().splitlines() or ["(no output)"])[-1])
    bad += int(r.returncode != 0)
else:
    print("second interpreter: %s is not here" % PY39)
sys.exit(1 if bad else 0)
