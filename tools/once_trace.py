"""phases of snpm_genotype_once[_coded] (SNPM_ONCE_TRACE) on 1135 x 11M int8, one 200k-SNP sample, in its three forms:
fused + zero-copy (default for coded samples), fused + copies, the first version's kernels and copies (SNPM_ONCE_FUSED=0)"""
import sys, time, os, numpy as np
sys.path.insert(0, '.')
os.environ["SNPM_ONCE_TRACE"] = "1"
from snpmatch_amd import engine, synth
n_snp, n_acc, n = 11_000_000, 1135, 200_000
rng = np.random.default_rng(5)
rows = np.sort(rng.choice(n_snp, size=n, replace=False)).astype(np.int64)
wei = synth.planted_sample(rng, synth.panel_rows(1001, rows, 416, 4)[:, 1], 0.02)[1]
n_in = 250000
sidx = np.sort(rng.choice(n_in, size=n, replace=False)).astype(np.int64)
wall = np.zeros((n_in, 3)); wall[sidx] = wei
tab = np.concatenate([engine.pl_table(256), [0.0]])
codes = engine.weight_codes(wall, tab)
forms = [("default", {}), ("zero-copy", {"SNPM_ONCE_ZEROCOPY": "1"}), ("copies", {"SNPM_ONCE_ZEROCOPY": "0"}), ("unfused", {"SNPM_ONCE_FUSED": "0"})]
only = os.environ.get("ONCE_FORMS") or (sys.argv[1] if len(sys.argv) > 1 else None)      # (rocprofv3 takes the program itself after --: the selection as an argument)
for name, env in forms:
    if only and name not in only.split(","):
        continue
    os.environ.update(env)
    ctx = engine.Context(0)
    for k in env:
        del os.environ[k]
    panel = engine.Panel(ctx, n_snp, n_acc); panel.fill_synthetic(1001)
    print("==", name, flush=True)
    for i in range(int(os.environ.get("ONCE_REPS", 6))):
        t0 = time.perf_counter()
        out = panel.genotype_once(rows, wall, sidx)
        print("plain wall %.3f ms top %d" % ((time.perf_counter() - t0) * 1e3, int(np.nanargmin(out["lik"]))), flush=True)
    for i in range(int(os.environ.get("ONCE_REPS", 6))):
        t0 = time.perf_counter()
        out = panel.genotype_once(rows, codes, sidx, table=tab)
        print("coded wall %.3f ms top %d" % ((time.perf_counter() - t0) * 1e3, int(np.nanargmin(out["lik"]))), flush=True)
    panel.free(); ctx.close()
