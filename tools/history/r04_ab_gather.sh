#!/bin/bash
# A/B of the gathered-row instantiations of k_fast (int8 panel 1135 x 11M): rows per prefetch group, parts per CU
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r04c; mkdir -p $out
show() { python -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1])
for f,v in d['formats'].items():
    for l in v['legs']:
        print('%-28s %-6s %-42s wall %.3f ms  kernel %.4f ms  %.0f GB/s  frac %.3f' % ('$2', f, l['leg'], l['wall_ms_per_call'], l['kernel_ms_per_call'], l['achieved_GBs'], l['frac_of_hbm_peak']))
"; }
for v in "" gg2 gg8; do
  for pc in 0 16 64; do
    lib=${v:+$PWD/tools/ab/libsnpmatch_hip_$v.so}
    SNPM_SEG_BLOCKS_PER_CU=$pc SNPMATCH_HIP_LIB=$lib timeout -k 10 300 python tools/bench_real_panel.py --formats ${FORMATS:-int8} > $out/ab_${v:-base}_pc$pc.json 2>/dev/null
    show $out/ab_${v:-base}_pc$pc.json "${v:-base} parts/CU=$pc"
  done
done | tee $out/ab_gather_G_parts.txt
python - <<'PY' | tee $out/time_batch_of_one.txt
import sys, time, numpy as np
sys.path.insert(0, '.')
from snpmatch_amd import engine, synth
ctx = engine.Context(0)
n_snp, n_acc, n = 11_000_000, 1135, 200_000
for packed in (False, True):
    panel = engine.Panel(ctx, n_snp, n_acc, packed=packed); panel.fill_synthetic(1001)
    rng = np.random.default_rng(5)
    rows = np.sort(rng.choice(n_snp, size=n, replace=False)).astype(np.int64)
    wei = synth.planted_sample(rng, synth.panel_rows(1001, rows, 416, 4)[:, 1], 0.02)[1]
    cat = (rows, wei, np.array([0, n], dtype=np.int64))
    engine.score_batch(panel, cat); ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): out = engine.score_batch(panel, cat)
    dt = (time.perf_counter() - t0) / 50
    pin = (ctx.pinned_empty(rows.shape, np.int64), ctx.pinned_empty(wei.shape, np.float64), cat[2]); pin[0][:] = rows; pin[1][:] = wei
    engine.score_batch(panel, pin)
    t0 = time.perf_counter()
    for _ in range(50): out = engine.score_batch(panel, pin)
    dtp = (time.perf_counter() - t0) / 50
    print("packed=%d  score_batch of ONE 200k-SNP sample: pageable inputs %.3f ms, pinned inputs %.3f ms (top %d)" % (packed, dt * 1e3, dtp * 1e3, int(np.nanargmin(out['lik'][0]))))
    panel.free()
PY
