"""
The ctypes stubs INTEGRATION.md shows a maintainer of the reference (-m gpu): the Python code blocks of section B are executed
VERBATIM (only the library's file name is made absolute) against a toy DB and compared with the oracle, so the documented
binding is known to work: single GPU (`hipmatch.py`) and several GPUs from one process (`hipmatch_multi.py`; here the members
share device 0 over the loopback transport, and once a real RCCL group of one rank).
"""
import os
import re

import numpy as np
import pytest

from oracle import c_oracle
from oracle import snpmatch_oracle as orc
from snpmatch_amd import _lib, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def doc_blocks():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    single = [b for b in blocks if b.startswith("# snpmatch/core/hipmatch.py")]
    multi = [b for b in blocks if b.startswith("# snpmatch/core/hipmatch_multi.py")]
    assert len(single) == 1 and len(multi) == 1
    return single[0].replace('"libsnpmatch_hip.so"', repr(_lib.LIB_PATH)), multi[0]


def test_the_documented_stubs_run_and_agree_with_the_oracle():
    _lib.load()                                  # the HIP runtime choice of the package (torch may be imported by other tests)
    single, multi = doc_blocks()
    ns = {}
    exec(compile(single, "INTEGRATION.md:hipmatch.py", "exec"), ns)
    rng = np.random.default_rng(12)
    n_snp, n_acc = 30_000, 333
    db = rng.choice(np.array([-1, 0, 1, 2], dtype=np.int8), size=(n_snp, n_acc), p=[0.05, 0.60, 0.33, 0.02])
    rows = np.sort(rng.choice(n_snp, size=7000, replace=False)).astype(np.int64)
    codes = db[rows, 17].copy()
    codes[codes < 0] = 0
    wei = synth.sample_weights(rng, codes, 0.8)
    want_s, want_n = c_oracle.genotyper(db, rows, wei, 1000, False)
    panel = ns["upload_panel"](db)
    s, n = ns["genotyper_scores"](panel, n_acc, rows, wei)
    assert np.array_equal(n, want_n) and np.array_equal(s.astype(np.int64), want_s.astype(np.int64))
    exec(compile(multi, "INTEGRATION.md:hipmatch_multi.py", "exec"), ns)
    wl, wr = orc.calculate_likelihoods(want_s.astype(np.int64), want_n)
    for device_ids, flags in (([0, 0, 0], _lib.GROUP_LOOPBACK), ([0], 0)):
        mp = ns["MultiGpuPanel"](db, device_ids, flags)
        s, n, lik, lrt = mp.genotyper_scores(rows, wei)
        assert np.array_equal(n, want_n) and np.array_equal(s.astype(np.int64), want_s.astype(np.int64))
        np.testing.assert_allclose(lik, wl, rtol=1e-12, equal_nan=True)
        np.testing.assert_allclose(lrt, wr, rtol=1e-12, equal_nan=True)
        assert int(np.nanargmin(lik)) == 17
        ns["_lib"].snpm_group_free.argtypes = [ns["_p"]]
        ns["_lib"].snpm_group_free(mp.group)
