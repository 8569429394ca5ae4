#!/bin/bash
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03f; mkdir -p $out
for rep in 1 2; do
for shape in "1252 50000000" "2500 50000000" "5000 40000000" "12500 16000000"; do
  set -- $shape
  for cfg in "0 2000000" "4 2000000" "5 2000000" "4 -1" "0 -1"; do
    set -- $shape $cfg
    SNPM_OCC_CAP=$3 SNPM_LONG_SCAN_ROWS=$4 timeout -k 10 200 python bench.py --n-acc $1 --n-snp $2 --mode fast --steps 6 --warmup 2 --no-cpu-baseline --no-alternatives --no-end-to-end 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('cap=$3 long_rows=$4  %6s x %9s  kernel %.3f ms  frac %.4f' % ('$1','$2', r['avg_ms'], r['frac']))"
  done
done
done | tee $out/ab_5wave.txt
