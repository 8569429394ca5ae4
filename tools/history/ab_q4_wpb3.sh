#!/bin/bash
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03g; mkdir -p $out
for shape in "9216 30000000" "11264 30000000" "12288 30000000" "13312 30000000" "14336 30000000" "15360 30000000" "3000 50000000" "2048 50000000"; do
  set -- $shape
  for w in 3 4 5 6 7 8; do
    SNPM_FORCE_WPB=$w timeout -k 10 200 python bench.py --packed --n-acc $1 --n-snp $2 --mode fast --steps 5 --warmup 2 --no-cpu-baseline --no-alternatives --no-end-to-end 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('wpb=$w  %6s x %9s  %.3f ms  frac %.4f' % ('$1','$2', r['avg_ms'], r['frac']))"
  done
done | tee $out/ab_q4_wpb3.txt
