#!/opt/conda/bin/python3.9
"""
Build-container-only probe (needs /root/reference; not a pytest file), run under the image's SECOND interpreter
(/opt/conda/bin/python3.9: numpy 1.26, scipy 1.7 -- the goldens were generated under numpy 2.2 / scipy 1.15):

    /opt/conda/bin/python3.9 tests/golden/probe_reference_second_numpy.py      # prints "cases N diffs 0"

The UNMODIFIED reference's matchGTsAccs on the committed G1 / G1b inputs must reproduce the committed fp64 bit patterns and
counts: the summation-order rule the exactness contract follows (row after row for panels of two or more accessions, numpy's
pairwise route for one accession, core/snpmatch.py:85-87) is then pinned on two numpy generations, not an accident of one build.
likeliTest / calculate_likelihoods on the G4 grid and np_test_identity on the G5 inputs are compared the same way (libm's log
and scipy's binom.sf may differ in the last bit between builds: those are reported as max relative difference, limit 1e-12).
tests/golden/probe_oracle_vs_reference.py starts this script when the interpreter is there.
"""
import os
import sys
import types
import warnings

import numpy as np

if not os.path.isdir("/root/reference"):
    sys.exit("the reference is not present here: nothing to probe")
warnings.filterwarnings("ignore")
sys.dont_write_bytecode = True
for _m in ("allel", "h5py", "hmmlearn", "hmmlearn.hmm"):
    if _m not in sys.modules:
        try:
            __import__(_m)
        except Exception:               # noqa: BLE001
            sys.modules[_m] = types.ModuleType(_m)
if not hasattr(sys.modules["hmmlearn"], "hmm"):
    sys.modules["hmmlearn"].hmm = sys.modules["hmmlearn.hmm"]
sys.path.insert(0, "/root/reference")
from snpmatch.core import snpmatch as ref_sm  # noqa: E402

assert "/root/reference" in ref_sm.__file__
HERE = os.path.dirname(os.path.abspath(__file__))


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


tot = bad = 0
g = np.load(os.path.join(HERE, "g1_match.npz"))
for name in g["names"]:
    name = str(name)
    rs, rn = ref_sm.matchGTsAccs(g[name + "_wei"], g[name + "_db"].copy(), name.endswith("_1"))
    tot += 1
    if not (np.array_equal(bits(rs), bits(g[name + "_score"])) and np.array_equal(np.asarray(rn), g[name + "_ninfo"])):
        bad += 1
        print("DIFF G1", name)
g = np.load(os.path.join(HERE, "g1b_single_acc.npz"))
for name in g["names"]:
    name = str(name)
    key0 = name[:-2]
    rs, rn = ref_sm.matchGTsAccs(g[key0 + "_wei"], g[key0 + "_db"].copy(), name.endswith("_1"))
    tot += 1
    if not (np.array_equal(bits(rs), bits(g[name + "_score"])) and np.array_equal(np.asarray(rn), g[name + "_ninfo"])):
        bad += 1
        print("DIFF G1b", name)
# likelihoods: the G4 grid
g = np.load(os.path.join(HERE, "g4_likelihood.npz"))
worst = 0.0
for sc, ni, lw, rw in (("sc_i", "ni_i", "l_i", "r_i"), ("sc_f", "ni_f", "l_f", "r_f")):
    lik, lrt = ref_sm.GenotyperOutput.calculate_likelihoods(g[sc], g[ni])
    for got, want in ((lik, g[lw]), (lrt, g[rw])):
        tot += 1
        same_nan = np.array_equal(np.isnan(got), np.isnan(want))
        m = ~np.isnan(want)
        rel = np.abs(np.asarray(got)[m] - want[m]) / np.maximum(np.abs(want[m]), 1e-300)
        worst = max(worst, float(rel.max(initial=0.0)))
        if not same_nan or float(rel.max(initial=0.0)) > 1e-12:
            bad += 1
            print("DIFF G4 likelihoods", sc, float(rel.max(initial=0.0)))
print("numpy %s: cases %d diffs %d (likelihood grid: max relative difference %.2e)" % (np.__version__, tot, bad, worst))
sys.exit(1 if bad else 0)
