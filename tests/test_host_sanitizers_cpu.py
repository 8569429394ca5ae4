"""
SURVEY 5 "race detection / sanitizers": the pure-host sources of the library (the native VCF reader snpm_vcf.cpp and the
sorted-merge / galloping position intersection snpm_host.cpp) are compiled here with AddressSanitizer and UBSan together
with a small driver (tests/host_asan_driver.cpp) and run on the reference's sample VCF, on hostile VCF text and on random
intersection inputs.  GPU sanitizers are not available on the pool; the device code is covered by the parity tests.
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "snpmatch_amd", "csrc")


def test_host_sources_under_asan_and_ubsan(tmp_path, golden_dir):
    exe = str(tmp_path / "host_asan_driver")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined",
           "-fno-sanitize-recover=all", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "host_asan_driver.cpp"),
           os.path.join(CSRC, "snpm_vcf.cpp"), os.path.join(CSRC, "snpm_host.cpp"), "-lz", "-o", exe]
    subprocess.check_call(cmd)
    work = tmp_path / "cases"
    work.mkdir()
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    env.pop("LD_PRELOAD", None)
    r = subprocess.run([exe, os.path.join(golden_dir, "701_501.filter.vcf.gz"), str(work)], capture_output=True, text=True,
                       env=env, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
    lines = dict(l.split(" ", 1) for l in r.stdout.strip().split("\n") if " " in l)
    assert r.stdout.strip().endswith("done")
    assert lines["sample_vcf"] == "rc=0 records=10000"               # /root/reference/sample_files: 10 000 records
    assert lines["intersect"].startswith("rounds=400 not_increasing rc=-4 -4 -4 empty rc=0")
    assert lines["intersect_pooled"] == "shapes=8"          # the threaded search with buffers of exactly min(na, nb) entries
    # 150 000 records, ~9 MB of text: lines straddle the reader's 4 MiB blocks; sums of POS / DP / PL pin the content
    n = 150000
    want = "rc=0 records=%d" % n
    possum = sum(100 + 7 * i for i in range(n))
    dpsum = sum(i % 90 for i in range(n))
    plsum = sum((10 + i % 200) + (20 + i % 100) for i in range(n))
    for key in ("big_plain", "big_gz"):
        assert lines[key].startswith(want), lines[key]
        assert "possum=%d dpsum=%d plsum=%d" % (possum, dpsum, plsum) in lines[key], lines[key]
    # the unsanitised library agrees on what it accepts and what it hands back to the Python reader
    sys.path.insert(0, ROOT)
    from snpmatch_amd import _lib
    got = _lib.vcf_parse(os.path.join(golden_dir, "701_501.filter.vcf.gz"))
    assert got is not None and len(got["pos"]) == 10000
    out = {l.split(" ")[1]: l for l in r.stdout.split("\n") if l.startswith("case ")}
    assert "rc=0 records=0" in out["empty"] and "rc=0 records=0" in out["header_only"]
    assert "rc=0 records=1" in out["no_newline_at_end"]
    for declined in ("crlf", "bad_pos", "bad_dp", "bad_pl", "huge_gt", "huge_chrom", "long_number"):
        assert "rc=-4" in out[declined], out[declined]


def test_pooled_intersection_under_tsan(tmp_path):
    """the threaded galloping search (snpm_intersect_sorted_search from 4096 searched values on: ranges of the list on the
    persistent pool's threads, hit lists closed up afterwards) under ThreadSanitizer"""
    exe = str(tmp_path / "intersect_tsan_driver")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-pthread", "-fsanitize=thread",
                           "-I", os.path.join(ROOT, "include"), "-I", CSRC, os.path.join(ROOT, "tests", "intersect_tsan_driver.cpp"),
                           os.path.join(CSRC, "snpm_host.cpp"), "-o", exe])
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1:second_deadlock_stack=1")
    env.pop("LD_PRELOAD", None)
    r = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-4000:]
    assert r.stdout.strip().endswith("done") and "fails=0" in r.stdout


def test_hdf5_reader_under_asan_and_ubsan(tmp_path, golden_dir):
    """csrc/snpm_h5.cpp parses untrusted files: the six fixtures written by real h5py (both file format generations), then 180 randomly damaged copies of
    them (bytes overwritten, files cut short), under AddressSanitizer + UBSan.  Damage may produce errors, never a fault."""
    import numpy as np
    exe = str(tmp_path / "h5_asan_driver")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=all", "-I", os.path.join(ROOT, "include"), "-I", CSRC,
                           os.path.join(ROOT, "tests", "h5_asan_driver.cpp"), os.path.join(CSRC, "snpm_h5.cpp"), "-lz", "-o", exe])
    h5dir = os.path.join(golden_dir, "h5")
    good = [os.path.join(h5dir, n) for n in ("toy_db.hdf5", "toy_db.acc.hdf5", "stress.hdf5", "latest_format.hdf5",
                                             "latest_stress.hdf5", "latest_unlimited.hdf5")]
    rng = np.random.default_rng(11)
    bad = []
    for k in range(180):
        raw = bytearray(open(good[k % 6], "rb").read())
        if k % 4 == 3:
            raw = raw[:int(rng.integers(16, len(raw)))]
        else:
            for _ in range(int(rng.integers(1, 40))):
                # most of the damage goes where the metadata lives (the first 8 KB), the rest anywhere
                hi = 8192 if rng.random() < 0.7 else len(raw)
                raw[int(rng.integers(8, min(hi, len(raw))))] = int(rng.integers(0, 256))
        p = str(tmp_path / ("damaged_%03d.hdf5" % k))
        open(p, "wb").write(bytes(raw))
        bad.append(p)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:allocator_may_return_null=1", UBSAN_OPTIONS="print_stacktrace=1")
    env.pop("LD_PRELOAD", None)
    r = subprocess.run([exe] + good + bad, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
    lines = r.stdout.strip().split("\n")
    assert lines[-1] == "done" and len(lines) == 187
    assert lines[0].startswith("toy_db.hdf5 rc=0 objects=4") and lines[1].startswith("toy_db.acc.hdf5 rc=0 objects=4")
    assert lines[2].startswith("stress.hdf5 rc=0 objects=10")
    assert lines[3].startswith("latest_format.hdf5 rc=0 objects=4")        # the "latest" file format: version-2 headers, fixed arrays
    assert lines[4].startswith("latest_stress.hdf5 rc=0 objects=15")       # (its refused dataset and its dense group do not count)
    assert lines[5].startswith("latest_unlimited.hdf5 rc=0 objects=3")     # root, accessions, positions: snps is refused by name


def test_loader_building_blocks_under_tsan_and_asan(tmp_path):
    """csrc/snpm_hostpool.hpp (thread pool, 2-bit packer, non-temporal copies, exact / O_DIRECT reads) under ThreadSanitizer and
    under AddressSanitizer + UBSan: SURVEY 5 "race detection" for the threaded part of the loader"""
    src = os.path.join(ROOT, "tests", "loader_tsan_driver.cpp")
    for name, flags in (("tsan", ["-fsanitize=thread"]), ("asan", ["-fsanitize=address,undefined", "-fno-sanitize-recover=all"])):
        exe = str(tmp_path / ("loader_driver_" + name))
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-pthread"] + flags +
                              ["-I", CSRC, src, "-o", exe])
        work = tmp_path / name
        work.mkdir()
        env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1:second_deadlock_stack=1", ASAN_OPTIONS="detect_leaks=1",
                   UBSAN_OPTIONS="print_stacktrace=1")
        env.pop("LD_PRELOAD", None)
        r = subprocess.run([exe, str(work)], capture_output=True, text=True, env=env, timeout=900)
        assert r.returncode == 0, (name, r.returncode, r.stdout[-2000:], r.stderr[-4000:])
        assert "ThreadSanitizer" not in r.stderr and "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
        assert "fails=0" in r.stdout and r.stdout.strip().endswith("done")
