"""
The gfx950 code object inside the built library: no kernel may spill vector registers or use scratch.

Why this is a correctness test and not a performance wish: the compiler of this image placed the spill stores of
long-lived per-lane values in front of the ``s_or_b64 exec`` that ends a divergent region -- lanes that had been inactive
there got stale scratch contents back, and one instantiation of k_fast_packed_q4 (segmented, dense rows, 32-row tiles)
returned garbage counts for every accession of its full waves (tests/test_gpu_batch.py::
test_dense_windows_on_narrow_packed_panels).  A reload also waits for every row load in flight.  Scalar registers spilled
into lanes of a vector register (v_writelane) do not depend on EXEC and are allowed.
"""
import os
import re
import shutil
import struct
import subprocess

import pytest

from snpmatch_amd import _lib

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def gfx950_code_object(path):
    data = open(path, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    at = data.find(magic)
    assert at >= 0, "no offload bundle in " + path
    (count,) = struct.unpack_from("<Q", data, at + 24)
    off = at + 32
    for _ in range(count):
        o, size, tlen = struct.unpack_from("<QQQ", data, off)
        off += 24
        triple = data[off:off + tlen].decode()
        off += tlen
        if "gfx950" in triple:
            return data[at + o:at + o + size]
    raise AssertionError("no gfx950 entry in the bundle")


@pytest.mark.skipif(not (os.path.exists(READELF) or shutil.which("llvm-readelf")), reason="llvm-readelf not available")
def test_no_kernel_spills_vector_registers(tmp_path):
    co = tmp_path / "snpm_gfx950.elf"
    co.write_bytes(gfx950_code_object(_lib.LIB_PATH))
    tool = READELF if os.path.exists(READELF) else shutil.which("llvm-readelf")
    notes = subprocess.run([tool, "--notes", str(co)], capture_output=True, text=True, check=True).stdout
    kernels = re.split(r"\n\s*- \.agpr_count", notes)[1:]
    assert len(kernels) > 100                                # every template instantiation is a kernel of its own
    bad = []
    for k in kernels:
        name = re.search(r"\.name:\s+(\S+)", k).group(1)
        spills = int(re.search(r"\.vgpr_spill_count:\s+(\d+)", k).group(1))
        scratch = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", k).group(1))
        if spills or scratch:
            bad.append((name, spills, scratch))
    assert not bad, "kernels with vector-register spills / scratch: %s" % bad[:8]
