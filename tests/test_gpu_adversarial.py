"""
The certificate of SNPM_MODE_EXACT under weights it was not tuned for (-m gpu).

A real VCF carries PLs in the thousands (exp(-PL/10) underflows through the denormals to exactly 0,
core/parsers.py:141-151), and snpm_query_create accepts any fp64: negative weights, weights above 1, magnitudes from
1e-300 to 1e300.  For every seeded case: STRICT == the C oracle bit for bit, EXACT has the same informative counts and the
same truncated scores (``int(ScoreList)``, core/snpmatch.py:96) as STRICT, and differs from it by no more than its
bound -- single queries, batched samples and certified windows.  NaN / infinite weights are refused with the
AssertionError of a bad argument: the reference multiplies 0/1 masks by the weights (core/snpmatch.py:85-87), so one such
weight turns every accession's score into NaN and ``int(NaN)`` raises in GenotyperOutput.
"""
import numpy as np
import pytest

from oracle import c_oracle
from snpmatch_amd import engine
from snpmatch_amd.core import snpmatch

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def rand_db(rng, n, n_acc):
    return rng.choice(np.array([-1, 0, 1, 2], dtype=np.int8), size=(n, n_acc), p=[0.05, 0.60, 0.33, 0.02])


def same_floats(a, b):
    """bit-identical, NaNs of any sign / payload counting as equal (x86 and gfx950 produce different default NaNs)"""
    a, b = np.asarray(a), np.asarray(b)
    nan = np.isnan(a) & np.isnan(b)
    return bool(np.all(nan | (bits(a) == bits(b))))


def adversarial_weights(rng, n, kind):
    if kind == "big_pl":                   # PLs up to 5000: denormals below exp(-70.8), exact zeros from PL 7451 on... and far before
        pl = rng.integers(0, 5001, size=(n, 3)).astype(np.float64)
        pl[np.arange(n), rng.integers(0, 3, size=n)] = 0.0
        if n > 5:
            pl[rng.integers(0, n, size=n // 5)] = rng.choice([7400.0, 7440.0, 7450.0, 7451.0, 8000.0], size=(n // 5, 3))
        return np.exp(pl / (-10.0))
    if kind == "magnitudes":               # 1e-300 .. 1e300, both signs
        return rng.choice([-1.0, 1.0], size=(n, 3)) * 10.0 ** rng.uniform(-300, 300, size=(n, 3))
    if kind == "tiny":                     # everything far below 1: totals truncate to 0, denormal partial sums
        return 10.0 ** rng.uniform(-320, -290, size=(n, 3))
    if kind == "negative_and_above_one":
        return rng.normal(0.0, 3.0, size=(n, 3))
    if kind == "near_integers":            # many totals within rounding distance of an integer
        w = rng.integers(0, 4, size=(n, 3)).astype(np.float64)
        w[rng.random((n, 3)) < 0.02] += 2.0 ** -30
        return w
    if kind == "large_integers":           # exact integers whose sums leave the 2^53 range
        return np.floor(10.0 ** rng.uniform(10, 16, size=(n, 3)))
    if kind == "near_overflow":            # finite weights, sums that overflow (inf, then inf - inf = NaN in BOTH orders or neither)
        w = rng.random((n, 3))
        w[rng.integers(0, n, size=max(1, n // 50))] = rng.choice([-1.0, 1.0]) * 1.0e308
        return w
    raise ValueError(kind)


KINDS = ["big_pl", "magnitudes", "tiny", "negative_and_above_one", "near_integers", "large_integers", "near_overflow"]


def check_exact_against_strict(se, ne, ss, ns, bound, tag):
    assert np.array_equal(ne, ns), tag
    fin = np.isfinite(ss)
    assert np.array_equal(np.trunc(se[fin]), np.trunc(ss[fin])), tag
    assert same_floats(se[~fin], ss[~fin]), tag                    # not finite: re-evaluated in reference order
    if np.isfinite(bound):
        assert np.all(np.abs(se[fin] - ss[fin]) <= bound), tag


def test_certificate_under_adversarial_weights():
    ctx = engine.Context(0)
    rng = np.random.default_rng(20261004)
    case = 0
    for kind in KINDS:
        for rep in range(14):
            case += 1
            n_snp = int(rng.choice([300, 2500, 9000, 20_000]))
            n_acc = int(rng.choice([3, 64, 257, 1135]))
            packed = bool(rng.integers(0, 2))
            skip = bool(rng.integers(0, 2))
            chunk = int(rng.choice([7, 1000, 1001]))
            db = rand_db(rng, n_snp, n_acc)
            n = int(rng.integers(1, n_snp + 1))
            rows = np.sort(rng.choice(n_snp, size=n, replace=False)).astype(np.int64) if rep % 2 else None
            if rows is None:
                n = n_snp
            wei = adversarial_weights(rng, n, kind)
            tag = "case %d %s: %dx%d n=%d packed=%s skip=%s chunk=%d" % (case, kind, n_snp, n_acc, n, packed, skip, chunk)
            panel = engine.Panel.from_host(ctx, db, packed=packed)
            q = engine.Query(panel, rows, wei)
            with np.errstate(all="ignore"):
                want_s, want_n = c_oracle.genotyper(db, rows, wei, chunk, skip)
            ss, ns = q.run(chunk, skip, engine.MODE_STRICT)
            assert same_floats(ss, want_s) and np.array_equal(ns, want_n), tag
            se, ne, info = q.run(chunk, skip, engine.MODE_EXACT, return_info=True)
            check_exact_against_strict(se, ne, ss, ns, q.error_bound(chunk), tag)
            if kind == "large_integers" and not info["all_integer_weights"]:
                assert info["n_strict_reeval"] > 0, tag                  # totals beyond 2^53 cannot be certified: they are re-scored
            # certified windows: per-window counts and the totals' counts as the reference-order pass gives them
            cuts = np.sort(rng.integers(0, n + 1, size=int(rng.integers(1, 9))))
            off = np.concatenate([[0], cuts, [n]]).astype(np.int64)
            ws, wn, ts, tn = q.run_windows(off, skip)
            fs, fn, fts, ftn = q.run_windows(off, skip, fast=True)
            assert np.array_equal(fn, wn) and np.array_equal(ftn, tn), tag
            fin = np.isfinite(ws)
            assert np.array_equal(np.trunc(fs[fin]), np.trunc(ws[fin])), tag
            fin = np.isfinite(ts)
            assert np.array_equal(np.trunc(fts[fin]), np.trunc(ts[fin])), tag
            # the same sample as one of three of a batched call (its own certificate per (sample, accession))
            if rep % 3 == 0:
                r_all = rows if rows is not None else np.arange(n, dtype=np.int64)
                half = max(1, n // 2)
                samples = [(r_all[:half], wei[:half]), (r_all, wei), (r_all[half - 1:], wei[half - 1:])]
                out = engine.score_batch(panel, samples, chunk, skip, engine.MODE_EXACT, likelihoods=False)
                check_exact_against_strict(out["score"][1], out["ninfo"][1], ss, ns, np.inf, tag + " (batch)")
                s0 = engine.Query(panel, samples[0][0], samples[0][1]).run(chunk, skip, engine.MODE_STRICT)
                check_exact_against_strict(out["score"][0], out["ninfo"][0], s0[0], s0[1], np.inf, tag + " (batch, sample 0)")
            q.free()
            panel.free()
    assert case == 98
    ctx.close()


def test_non_finite_weights_are_refused():
    ctx = engine.Context(0)
    rng = np.random.default_rng(1)
    db = rand_db(rng, 500, 40)
    panel = engine.Panel.from_host(ctx, db)
    for bad in (np.nan, np.inf, -np.inf):
        wei = rng.random((500, 3))
        wei[123, 1] = bad
        with pytest.raises(AssertionError, match="finite"):
            engine.Query(panel, None, wei)
        with pytest.raises(AssertionError, match="finite"):
            engine.score_batch(panel, [(np.arange(500, dtype=np.int64), wei)])
        with pytest.raises(AssertionError, match="finite"):
            snpmatch.matchGTsAccs(wei, db)
    # the largest finite double is a weight like any other
    wei = rng.random((500, 3))
    wei[7, 0] = 1.7976931348623157e308
    s, n = engine.Query(panel, None, wei).run(1000, False, engine.MODE_EXACT)
    want_s, want_n = c_oracle.genotyper(db, None, wei, 1000, False)
    assert np.array_equal(n, want_n) and same_floats(s, want_s)
    ctx.close()
