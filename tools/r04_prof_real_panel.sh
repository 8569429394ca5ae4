#!/bin/bash
# rocprofv3 kernel stats of the real-panel legs (tools/bench_real_panel.py) -> gpurun_out/r04d/
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r04d; mkdir -p $out
for fmt in int8 packed; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_$fmt -- python3 tools/bench_real_panel.py --formats $fmt --reps 10 > $out/real_panel_$fmt.json 2> $out/real_panel_$fmt.err
  f=$(find $out/trace_$fmt -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp $f $out/real_panel_${fmt}_kernel_stats.csv
  t=$(find $out/trace_$fmt -name '*kernel_trace.csv' | head -1)
  [ -n "$t" ] && python tools/kernel_trace_summary.py $t > $out/real_panel_${fmt}_kernel_by_grid.txt
  rm -rf $out/trace_$fmt
done
python - <<'PY'
import sys, numpy as np
sys.path.insert(0, '.')
sys.argv = ['x']
import importlib.util
spec = importlib.util.spec_from_file_location("brp", "tools/bench_real_panel.py"); brp = importlib.util.module_from_spec(spec); spec.loader.exec_module(brp)
from snpmatch_amd import engine, synth
ctx = engine.Context(0)
g0, bounds, positions = brp.tair10_layout(brp.N_SNP)
rng = np.random.default_rng(5)
rows = np.sort(rng.choice(brp.N_SNP, size=200000, replace=False)).astype(np.int64)
wei = synth.planted_sample(rng, synth.panel_rows(1001, rows, 416, 4)[:, 1], 0.02)[1]
off = brp.window_offsets(g0, bounds, positions, rows)
panel = engine.Panel(ctx, brp.N_SNP, 1135); panel.fill_synthetic(1001)
q = engine.Query(panel, rows, wei)
q.run_windows(off, False, fast=True)
print("windows", len(off) - 1, "info", q.last_windows_info)
PY
