"""
Synthetic panels and samples (SURVEY.md 8d / BASELINE.md recipe) for tests and bench.py.

panel_values() is the numpy twin of the device generator k_synth (snpm_kernels.hpp): element
(snp, acc) is a pure function of (seed, snp, acc), so any slab or accession shard can be
reproduced on the CPU without touching the GPU.
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(z):
    z = (z + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def panel_values(seed, snp0, n_snp, acc0, n_acc):
    """int8 [n_snp, n_acc] with P(-1,0,1,2) = (3277, 39321, 21627, 1311)/65536; acc0 % 4 == 0."""
    assert acc0 % 4 == 0
    with np.errstate(over="ignore"):
        nq = (n_acc + 3) // 4
        snp = (np.arange(n_snp, dtype=np.uint64) + np.uint64(snp0))[:, None]
        quad = (np.arange(nq, dtype=np.uint64) + np.uint64(acc0 // 4))[None, :]
        h = _splitmix64(_splitmix64(np.uint64(seed) ^ (snp * np.uint64(0xD6E8FEB86659FD93))) + quad)
        out = np.empty((n_snp, nq * 4), dtype=np.int8)
        for j in range(4):
            u = ((h >> np.uint64(16 * j)) & np.uint64(0xFFFF)).astype(np.int64)
            c = np.where(u < 3277, -1, np.where(u < 42598, 0, np.where(u < 64225, 1, 2))).astype(np.int8)
            out[:, j::4] = c
    return out[:, :n_acc]


def panel_rows(seed, rows, acc0, n_acc):
    """panel_values for an arbitrary list of SNP rows (int8 [len(rows), n_acc])."""
    assert acc0 % 4 == 0
    rows = np.asarray(rows, dtype=np.uint64)
    with np.errstate(over="ignore"):
        nq = (n_acc + 3) // 4
        quad = (np.arange(nq, dtype=np.uint64) + np.uint64(acc0 // 4))[None, :]
        h = _splitmix64(_splitmix64(np.uint64(seed) ^ (rows[:, None] * np.uint64(0xD6E8FEB86659FD93))) + quad)
        out = np.empty((len(rows), nq * 4), dtype=np.int8)
        for j in range(4):
            u = ((h >> np.uint64(16 * j)) & np.uint64(0xFFFF)).astype(np.int64)
            out[:, j::4] = np.where(u < 3277, -1, np.where(u < 42598, 0, np.where(u < 64225, 1, 2))).astype(np.int8)
    return out[:, :n_acc]


def sample_weights(rng, codes, frac_pl=0.8):
    """[n,3] weights: frac_pl of the rows PL-derived exp(-PL/10) (integer PL in 1..255, 0 for the
    called genotype), the rest hard one-hot (core/parsers.py:132-150)."""
    n = len(codes)
    col = np.where(codes == 0, 0, np.where(codes == 2, 1, 2))
    pl = rng.integers(1, 256, size=(n, 3)).astype(np.float64)
    pl[np.arange(n), col] = 0.0
    wei = np.exp(pl / (-10.0))
    hard = rng.random(n) >= frac_pl
    onehot = np.zeros((n, 3))
    onehot[np.arange(n), col] = 1.0
    wei[hard] = onehot[hard]
    return wei


def planted_sample(rng, planted_col, err=0.02, frac_pl=0.8):
    """Sample genotype codes = the planted accession's calls with `err` random replacements."""
    codes = np.array(planted_col, dtype=np.int8)
    miss = codes < 0
    codes[miss] = rng.integers(0, 2, size=int(miss.sum()))
    flip = rng.random(len(codes)) < err
    codes[flip] = rng.choice(np.array([0, 1, 2], dtype=np.int8), size=int(flip.sum()))
    return codes, sample_weights(rng, codes, frac_pl)


def exp_table():
    """exp(-k / 10), k = 0..255, as this host's libm rounds it: the weights a PL of k yields (core/parsers.py:147-150)"""
    return np.exp(np.arange(256, dtype=np.float64) / (-10.0))


def sample_weights_twin(seed, snp0, n, planted, err=0.02, frac_pl=0.8):
    """numpy twin of the device sample generator k_synth_sample (snpm_sample_synthetic): float64 [n, 3]"""
    err_pm, pl_pm = int(round(err * 1000)), int(round(frac_pl * 1000))
    with np.errstate(over="ignore"):
        s = np.arange(n, dtype=np.uint64) + np.uint64(snp0)
        hq = _splitmix64(_splitmix64(np.uint64(seed) ^ (s * np.uint64(0xD6E8FEB86659FD93))) + np.uint64(planted // 4))
        u = ((hq >> np.uint64(16 * (planted % 4))) & np.uint64(0xFFFF)).astype(np.int64)
        code = np.where(u < 3277, 255, np.where(u < 42598, 0, np.where(u < 64225, 1, 2))).astype(np.int64)
        h = _splitmix64(_splitmix64((np.uint64(seed) ^ np.uint64(0x5851F42D4C957F2D)) + s * np.uint64(0x9FB21C651E98DF25)))
        h2 = _splitmix64(h + np.uint64(0x2545F4914F6CDD1D))
        code = np.where(code == 255, (h & np.uint64(1)).astype(np.int64), code)
        flip = ((h >> np.uint64(8)) & np.uint64(0xFFFFFF)).astype(np.int64) % 1000 < err_pm
        code = np.where(flip, ((h >> np.uint64(40)) & np.uint64(0xFFFF)).astype(np.int64) % 3, code)
        called = np.where(code == 0, 0, np.where(code == 2, 1, 2))
        is_pl = (h2 & np.uint64(0xFFFFFF)).astype(np.int64) % 1000 < pl_pm
        pa = 1 + ((h2 >> np.uint64(24)) & np.uint64(0xFFFF)).astype(np.int64) % 255
        pb = 1 + ((h2 >> np.uint64(40)) & np.uint64(0xFFFF)).astype(np.int64) % 255
    tab = exp_table()
    rows = np.arange(n)
    wei = np.zeros((n, 3))
    wei[rows, called] = 1.0
    pl = np.zeros((n, 3))
    pl[rows, called] = tab[0]
    pl[rows, (called + 1) % 3] = tab[pa]
    pl[rows, (called + 2) % 3] = tab[pb]
    wei[is_pl] = pl[is_pl]
    return wei
