#!/bin/bash
# Confirms that the default q4 block size (q4_waves_per_block) is the
# measured best among forced 4 / 7 / 8-wave blocks on a sweep of widths.
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03g; mkdir -p $out
for shape in "6144 40000000" "7000 40000000" "10000 40000000" "13312 30000000" "14336 30000000" "15360 30000000" "20000 20000000"; do
  set -- $shape
  for w in 0 4 7 8; do
    SNPM_FORCE_WPB=$w timeout -k 10 200 python bench.py --packed --n-acc $1 --n-snp $2 --mode fast --steps 6 --warmup 2 --no-cpu-baseline --no-alternatives --no-end-to-end 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('wpb=$w  %6s x %9s  %s %.3f ms  frac %.4f' % ('$1','$2', r['kernel'], r['avg_ms'], r['frac']))"
  done
done | tee $out/ab_q4_wpb4.txt
