"""cold `python -m snpmatch_amd inbred` runs (a fresh process each): a 1135 x 2M .snpm DB (packed and int8), a 100k-record VCF"""
import os, sys, subprocess, tempfile, time, shutil
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from snpmatch_amd import synth
from snpmatch_amd.core import snp_genotype, genomes
n_snp, n_acc, n_s = 2_000_000, 1135, 100_000
g0 = genomes.Genome("athaliana_tair10")
frac = np.cumsum(g0.chrlen) / g0.chrlen.sum()
bounds = np.concatenate([[0], np.round(frac * n_snp).astype(np.int64)])
positions = np.concatenate([1 + (np.arange(bounds[c + 1] - bounds[c]) * int(g0.chrlen[c] - 1)) // int(bounds[c + 1] - bounds[c]) for c in range(5)])
regions = np.array([(int(bounds[c]), int(bounds[c + 1])) for c in range(5)])
tmp = tempfile.mkdtemp(prefix="snpm_cold_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
try:
    t0 = time.perf_counter()
    snps = np.concatenate([synth.panel_rows(1001, np.arange(r0, min(r0 + 250_000, n_snp)), 0, n_acc) for r0 in range(0, n_snp, 250_000)])
    for packed in (True, False):
        snp_genotype.save_native(os.path.join(tmp, "db_packed.snpm" if packed else "db_int8.snpm"), snps, np.array([str(i) for i in range(n_acc)]),
                                 positions, np.array(["1", "2", "3", "4", "5"]), regions, packed=packed)
    rng = np.random.default_rng(5)
    rows = np.sort(rng.choice(n_snp, size=n_s, replace=False))
    col = snps[rows, 417]
    del snps
    vcf = os.path.join(tmp, "sample.vcf")
    chr_of = np.searchsorted(bounds[1:], rows, side="right")
    with open(vcf, "w") as fh:
        fh.write("##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS1\n")
        pl = rng.integers(20, 256, size=(n_s, 3))
        for i in range(n_s):
            c = int(col[i]) if col[i] >= 0 and rng.random() > 0.02 else int(rng.integers(0, 3))
            g = ("0/0", "1/1", "0/1")[c]
            p = pl[i].copy(); p[(0, 2, 1)[c]] = 0
            fh.write("Chr%d\t%d\t.\tC\tT\t40\t.\tDP=%d\tGT:PL\t%s:%d,%d,%d\n" % (chr_of[i] + 1, positions[rows[i]], rng.integers(5, 40), g, p[0], p[1], p[2]))
    print("setup %.1f s" % (time.perf_counter() - t0), flush=True)
    env = dict(os.environ, PYTHONPATH=ROOT)
    for db in ("db_packed.snpm", "db_int8.snpm"):
        for rep in range(3):
            if rep == 0:
                for f in (vcf + ".snpmatch.npz", vcf + ".snpmatch.stats.json"):
                    if os.path.exists(f):
                        os.remove(f)
            t0 = time.perf_counter()
            r = subprocess.run([sys.executable, "-m", "snpmatch_amd", "inbred", "-v", "-i", vcf, "-d", os.path.join(tmp, db), "-o", os.path.join(tmp, "out")],
                               env=env, capture_output=True, text=True)
            dt = time.perf_counter() - t0
            top = open(os.path.join(tmp, "out.matches.json")).read()[:0]
            print("%s run %d (%s): %.2f s rc %d" % (db, rep, "parses the VCF" if rep == 0 else "parse cache", dt, r.returncode), flush=True)
            if rep == 2:
                import datetime, re
                start = datetime.datetime.fromtimestamp(time.time() - dt)
                for line in r.stderr.splitlines():
                    m = re.match(r"(\d+-\d+-\d+ \d+:\d+:\d+),(\d+) - (.*)", line)
                    if m:
                        t = datetime.datetime.strptime(m.group(1), "%Y-%m-%d %H:%M:%S") + datetime.timedelta(milliseconds=int(m.group(2)))
                        print("   +%4d ms  %s" % ((t - start).total_seconds() * 1e3, m.group(3)[:110]))
                print("   +%4d ms  process gone" % (dt * 1e3))
    r = subprocess.run([sys.executable, "-X", "importtime", "-c", "import snpmatch_amd.cli, snpmatch_amd.core.snpmatch, snpmatch_amd.core.csmatch"], env=env, capture_output=True, text=True)
    lines = [l for l in r.stderr.splitlines() if "|" in l]
    top = sorted(lines, key=lambda l: -int(l.split("|")[1]))[:12]
    print("\n".join(top))
finally:
    shutil.rmtree(tmp, ignore_errors=True)
