#!/bin/bash
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03c; mkdir -p $out
for rep in 1 2; do
for shape in "10000 20019000" "8192 24000000" "1252 50000000" "12500 16000000" "2500 50000000"; do
  set -- $shape
  for v in 128 192 248; do
    SNPMATCH_HIP_LIB=$PWD/tools/ab/libtile$v.so timeout -k 10 200 python bench.py --n-acc $1 --n-snp $2 --mode fast --steps 6 --warmup 2 --no-cpu-baseline --no-alternatives --no-end-to-end 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('tile=$v  %6s x %9s  kernel %.3f ms  frac %.4f  ok %s' % ('$1','$2', r['avg_ms'], r['frac'], d['checks']['top_hit_is_planted']))"
  done
done
done | tee $out/ab_tile_rows.txt
