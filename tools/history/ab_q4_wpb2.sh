#!/bin/bash
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03g; mkdir -p $out
for shape in "8192 50000000" "16384 30000000" "24576 20000000" "7000 50000000" "6144 50000000"; do
  set -- $shape
  for w in 0 4 8; do
    SNPM_FORCE_WPB=$w timeout -k 10 200 python bench.py --packed --n-acc $1 --n-snp $2 --mode exact --steps 6 --warmup 2 --no-cpu-baseline --no-alternatives --no-end-to-end 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('wpb=$w  %6s x %9s  %s %.3f ms  frac %.4f  ok %s' % ('$1','$2', r['kernel'], r['avg_ms'], r['frac'], d['checks']['top_hit_is_planted']))"
  done
done | tee $out/ab_q4_wpb2.txt
