#!/usr/bin/env python3
"""Differential fuzz of the shared-row scan against the C oracle with other seeds than the suite's
(tests/test_gpu_shared.py::test_random_configurations_against_the_oracle holds the generator): widths around the 128-accession
wave tiles and the split packed layouts, batch sizes, sample lengths, overlaps, digit counts, row-tile counts, digit-matrix
budgets, chunk lengths; both formats, both skip_hets settings.
usage: tools/fuzz_shared.py [first_seed=100] [n_seeds=16]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_gpu_shared as t  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 16
for seed in range(first, first + count):
    t.test_random_configurations_against_the_oracle(seed)
    print("seed %d: 6 configurations ok" % seed, flush=True)
print("%d configurations, no difference" % (6 * count))
