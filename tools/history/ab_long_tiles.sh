#!/bin/bash
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03d; mkdir -p $out
for rep in 1 2 3; do
  for v in -1 2000000; do
    SNPM_LONG_SCAN_ROWS=$v timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-alternatives --no-end-to-end 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('long_scan_rows=$v  ms/step %.3f  kernel %.3f ms  frac %.4f  e2e %.4f' % (d['ms_per_step'], r['avg_ms'], r['frac'], r['end_to_end_frac']))"
  done
done | tee $out/ab_long_tiles.txt
