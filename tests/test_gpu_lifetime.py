"""
Handle lifetime at the C ABI and in the Python layer (-m gpu).

Round 1's first GPU run ended in `std::bad_variant_access` -> SIGABRT after a failed test: the context had been
destroyed while a traceback still held a Panel and a Query, whose late frees then read the freed context
(gpurun_out/test1.log).  The library now orphans live panels / queries in snpm_destroy, so any order of frees is
harmless, and every Context closes itself at interpreter exit.  The cases run in child processes: what is under
test is the exit status of a script that leaves objects behind.
"""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PRELUDE = """
import sys
sys.path.insert(0, %r)
import ctypes as C
import numpy as np
from snpmatch_amd import engine, _lib
rng = np.random.default_rng(0)
db = rng.integers(-1, 3, size=(3000, 70), dtype=np.int8)
wei = rng.random((3000, 3))
""" % ROOT


def run_child(body, timeout=300):
    code = PRELUDE + textwrap.dedent(body)
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=timeout)


def test_script_that_raises_exits_with_its_exception():
    r = run_child("""
        ctx = engine.Context(0)
        panel = engine.Panel.from_host(ctx, db)
        q = engine.Query(panel, None, wei)
        s, n = q.run()
        keep = [ctx, panel, q]               # nothing is freed
        raise ValueError("user error after scoring")
    """)
    assert r.returncode == 1, (r.returncode, r.stderr[-2000:])
    assert "ValueError: user error after scoring" in r.stderr and "terminate called" not in r.stderr


def test_context_closed_before_its_panels_and_queries():
    r = run_child("""
        ctx = engine.Context(0)
        panel = engine.Panel.from_host(ctx, db)
        q = engine.Query(panel, None, wei)
        want = q.run()
        lib = ctx.lib
        # the failure of round 1 at the C ABI: destroy first, then the late frees
        ph, qh = panel.h, q.h
        assert lib.snpm_destroy(ctx.h) == 0
        ctx.h = None
        score = np.zeros(70); ninfo = np.zeros(70, dtype=np.int64)
        rc = lib.snpm_query_run(qh, 1000, 0, 0, _lib.ptr(score), _lib.ptr(ninfo), None)
        assert rc == _lib.SNPM_ERR_STATE, rc                     # orphaned handles are refused ...
        assert b"outlived" in lib.snpm_last_error(None)
        assert lib.snpm_panel_upload_rows(ph, 0, 1, _lib.ptr(db), 70) == _lib.SNPM_ERR_STATE
        assert lib.snpm_panel_free(ph) == 0                      # ... and freed without touching the device
        assert lib.snpm_query_free(qh) == 0
        panel.h = None; q.h = None
        # panel freed before its query, context still alive
        ctx2 = engine.Context(0)
        p2 = engine.Panel.from_host(ctx2, db)
        q2 = engine.Query(p2, None, wei)
        got = q2.run()
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
        p2h, q2h = p2.h, q2.h
        assert lib.snpm_panel_free(p2h) == 0
        assert lib.snpm_query_run(q2h, 1000, 0, 0, _lib.ptr(score), _lib.ptr(ninfo), None) == _lib.SNPM_ERR_STATE
        assert lib.snpm_query_free(q2h) == 0
        p2.h = None; q2.h = None
        q3 = engine.Query(engine.Panel.from_host(ctx2, db), None, wei)      # the context is still usable
        got = q3.run()
        assert np.array_equal(got[0], want[0])
        print("ok")
    """)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.returncode, r.stdout[-500:], r.stderr[-2000:])


def test_objects_left_to_the_garbage_collector_in_any_order():
    r = run_child("""
        import gc
        for order in range(3):
            ctx = engine.Context(0)
            panel = engine.Panel.from_host(ctx, db, packed=bool(order & 1))
            q = engine.Query(panel, None, wei)
            q.run()
            if order == 0:
                ctx.close(); del panel; del q
            elif order == 1:
                del ctx; panel.free(); del q
            else:
                del q, panel, ctx
            gc.collect()
        ctx = engine.Context(0)            # left open: closed by its atexit hook
        panel = engine.Panel.from_host(ctx, db)
        sys.exit(7)
    """)
    assert r.returncode == 7, (r.returncode, r.stderr[-2000:])
