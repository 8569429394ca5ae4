#!/bin/bash
# int8 k_fast at the width of the 1001 Genomes panel: block shapes and resident blocks
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03h; mkdir -p $out
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-28s %6s x %9s  %-8s %.3f ms  %.0f GB/s  frac %.4f' % ('$1','$2','$3', r['kernel'], r['avg_ms'], r['achieved'], r['frac']))"; }
for shape in "1135 40000000" "1135 11000000"; do
  set -- $shape
  common="--n-acc $1 --n-snp $2 --mode fast --steps 6 --warmup 2 --no-cpu-baseline --no-alternatives --no-end-to-end"
  for w in 0 1 2 3 4 6 8; do
    SNPM_FORCE_WPB=$w timeout -k 10 200 python bench.py $common 2>/dev/null | line wpb=$w $1 $2
  done
  for o in 2 3 4 6; do
    SNPM_OCC_CAP=$o timeout -k 10 200 python bench.py $common 2>/dev/null | line occ_cap=$o $1 $2
  done
  SNPM_LONG_SCAN_ROWS=1000000000 timeout -k 10 200 python bench.py $common 2>/dev/null | line tiles128 $1 $2
done | tee $out/ab_int8_1135.txt
