#!/usr/bin/env python3
"""ParseInputs on a synthetic single-sample VCF of N records (plain, gzip and BGZF): the sample-ingestion step in front of the scoring
(core/parsers.py:141-157, 178-213).  usage: tools/time_parse.py [n_records=1000000]"""
import gzip
import os
import shutil
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snpmatch_amd.core import parsers  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
rng = np.random.default_rng(1)
chrlen = [30427671, 19698289, 23459830, 18585056, 26975502]
per = [int(n * L / sum(chrlen)) for L in chrlen]
per[-1] += n - sum(per)
tmp = tempfile.mkdtemp(prefix="snpm_parse_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
try:
    plain = os.path.join(tmp, "sample.vcf")
    with open(plain, "w") as fh:
        fh.write("##fileformat=VCFv4.2\n##source=synthetic\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS1\n")
        for c, (m, L) in enumerate(zip(per, chrlen)):
            pos = np.sort(rng.choice(L, size=m, replace=False)) + 1
            gt = rng.choice(4, size=m, p=[0.55, 0.3, 0.05, 0.1])
            pl = rng.integers(1, 256, size=(m, 3))
            dp = rng.integers(1, 40, size=m)
            lines = []
            for i in range(m):
                if gt[i] == 3:
                    lines.append("Chr%d\t%d\t.\tC\tT\t44.50\t.\tAC=0;AF=0.00;AN=0;DP=0;FS=0.000;MQ=60.00\tGT:AD:DP\t./.:0,0:0\n" % (c + 1, pos[i]))
                else:
                    g = ("0/0", "1/1", "0/1")[gt[i]]
                    p = pl[i].copy()
                    p[(0, 2, 1)[gt[i]]] = 0
                    lines.append("Chr%d\t%d\t.\tC\tT\t40.40\t.\tAC=0;AF=0.00;AN=2;DP=%d;FS=0.000;InbreedingCoeff=-0.0234;MQ=60.00;QD=13.47;SOR=1.179\t"
                                 "GT:AD:DP:GQ:PL\t%s:3,0:%d:9:%d,%d,%d\n" % (c + 1, pos[i], dp[i], g, dp[i], p[0], p[1], p[2]))
            fh.write("".join(lines))
    gz = plain + ".gz"
    with open(plain, "rb") as fi, gzip.open(gz, "wb", compresslevel=1) as fo:
        shutil.copyfileobj(fi, fo)
    print("%d records: %s %.0f MB, %s %.0f MB; SNPM_VCF_THREADS=%s" % (n, os.path.basename(plain), os.path.getsize(plain) / 1e6,
                                                                    os.path.basename(gz), os.path.getsize(gz) / 1e6, os.environ.get("SNPM_VCF_THREADS", "default")))
    # the same text as bgzip / bcftools write it: independent gzip members of <= 64 KiB (inflated side by side by the native reader)
    import struct
    import zlib
    bgz = os.path.join(tmp, "sample.bgzf.vcf.gz")
    with open(plain, "rb") as fi, open(bgz, "wb") as fo:
        while True:
            piece = fi.read(65280)
            co = zlib.compressobj(6, zlib.DEFLATED, -15)
            raw = co.compress(piece) + co.flush()
            fo.write(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(raw) + 25))
            fo.write(raw + struct.pack("<II", zlib.crc32(piece) & 0xFFFFFFFF, len(piece)))
            if not piece:
                break
    print("%s %.0f MB" % (os.path.basename(bgz), os.path.getsize(bgz) / 1e6))
    for path in (plain, gz, bgz):
        for rep in range(4):
            for f in (path + ".snpmatch.npz", path + ".snpmatch.stats.json"):
                if os.path.exists(f):
                    os.remove(f)
            t0 = time.perf_counter()
            p = parsers.ParseInputs(path)
            t1 = time.perf_counter()
            p.wait_for_cache()
            t2 = time.perf_counter()
            print("%-14s rep %d: ParseInputs %.3f s (the .npz cache is on disk %.3f s later), %d called SNPs" % (os.path.basename(path), rep, t1 - t0, t2 - t1, len(p.pos)),
                  flush=True)
finally:
    shutil.rmtree(tmp, ignore_errors=True)
