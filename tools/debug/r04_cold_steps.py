"""the steps of a cold `inbred` run inside one fresh process, timed one by one (DB + sample made by tools/debug/r04_cold_cli.py's
recipe, kept in /dev/shm between the two processes of this script)"""
import os, sys, subprocess, tempfile, time, shutil
t_start = time.perf_counter()
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    db, vcf, out = sys.argv[2:5]
    marks = [("python + this script", time.perf_counter() - t_start)]
    def mark(name):
        marks.append((name, time.perf_counter() - t_start))
    import numpy as np
    mark("import numpy")
    from snpmatch_amd.core import snpmatch, parsers, snp_genotype
    mark("import snpmatch_amd.core")
    inp = parsers.ParseInputs(vcf)
    mark("ParseInputs (cache)")
    g = snp_genotype.Genotype(db, None)
    mark("Genotype(db)")
    from snpmatch_amd import engine
    ctx = engine.default_context()
    mark("library + HIP context")
    panel = g.panel()
    ctx.synchronize()
    mark("DB -> HBM")
    gt = snpmatch.Genotyper(inp, g, out, run_genotyper=False)
    gt.get_common_positions()
    mark("positions")
    res = gt.genotyper()
    mark("genotyper (one call)")
    gt.write_genotyper_output(res)
    mark("outputs")
    prev = 0.0
    for name, t in marks:
        print("   %-28s %6.1f ms" % (name, (t - prev) * 1e3))
        prev = t
    print("   %-28s %6.1f ms" % ("total before exit", prev * 1e3), flush=True)
    sys.exit(0)
import numpy as np
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
            p = pl[i].copy(); p[(0, 2, 1)[c]] = 0
            fh.write("Chr%d\t%d\t.\tC\tT\t40\t.\tDP=%d\tGT:PL\t%s:%d,%d,%d\n" % (chr_of[i] + 1, positions[rows[i]], rng.integers(5, 40), ("0/0", "1/1", "0/1")[c], p[0], p[1], p[2]))
    env = dict(os.environ, PYTHONPATH=ROOT)
    for db in ("db_packed.snpm", "db_int8.snpm"):
        for rep in range(3):
            t0 = time.perf_counter()
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", os.path.join(tmp, db), vcf, os.path.join(tmp, "out")], env=env, capture_output=True, text=True)
            print("%s run %d: %.0f ms wall, rc %d" % (db, rep, (time.perf_counter() - t0) * 1e3, r.returncode), flush=True)
            if rep == 2 or r.returncode:
                print(r.stdout + r.stderr[-600:])
finally:
    shutil.rmtree(tmp, ignore_errors=True)
