"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol the header
declares, and fails loudly (no fallback) when no GPU is present."""
import ctypes as C
import os
import re

import pytest

from snpmatch_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "snpmatch_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(snpm_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert header_symbols() == sorted(_lib.SYMBOLS)


def test_library_exports_every_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libsnpmatch_hip.so not built (run ./build_lib.sh)")
    lib = C.CDLL(_lib.LIB_PATH)
    for name in header_symbols():
        assert hasattr(lib, name), name
    assert lib.snpm_version() >= 100


def test_no_silent_fallback_without_gpu():
    """Without a usable device snpm_init must fail with a message; nothing computes on the CPU."""
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libsnpmatch_hip.so not built")
    lib = _lib.load()
    n = C.c_int(0)
    rc = lib.snpm_device_count(C.byref(n))
    if rc == 0 and n.value > 0:
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    rc = lib.snpm_init(0, C.byref(h))
    assert rc == _lib.SNPM_ERR_HIP
    assert not h.value
    assert b"HIP device" in lib.snpm_last_error(None)
    from snpmatch_amd import engine
    with pytest.raises(_lib.SnpmError):
        engine.Context(0)


def test_product_never_imports_oracle():
    """The product package must not reference oracle/ (it is test infrastructure)."""
    pkg = os.path.join(ROOT, "snpmatch_amd")
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                src = open(os.path.join(base, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "liboracle" not in src, f


def test_inline_asm_lds_reads_stay_untouched_until_waited_for(tmp_path):
    """k_scan_few issues ds_read_b128 from inline asm and waits with counted s_waitcnt; the
    compiler does not know those registers are in flight.  Compile the device code to ISA and verify that nothing
    touches a destination register between its read and the wait that covers it (tools/check_inflight_regs.py)."""
    import shutil
    import subprocess
    import sys
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    asm = str(tmp_path / "device.s")
    subprocess.run([hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", "-ffp-contract=off", "-I" + os.path.join(root, "include"),
                    "-I" + os.path.join(root, "snpmatch_amd", "csrc"), "--cuda-device-only", "-S", "-o", asm,
                    os.path.join(root, "snpmatch_amd", "csrc", "snpm_api.hip")], check=True, capture_output=True)
    res = subprocess.run([sys.executable, os.path.join(root, "tools", "check_inflight_regs.py"), asm], capture_output=True, text=True)
    assert res.returncode == 0, res.stdout
    assert "checked" in res.stdout and " 0 violations" in res.stdout
    assert int(res.stdout.split("checked")[1].split()[0]) >= 16          # the asm blocks are really there
