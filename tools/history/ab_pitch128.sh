#!/bin/bash
# int8 panels: rows padded to 128 B (whole cache lines) instead of 256 B
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03h; mkdir -p $out
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-12s %6s x %9s  %-8s %.3f ms  %.0f GB/s  frac %.4f' % ('$1','$2','$3', r['kernel'], r['avg_ms'], r['achieved'], r['frac']))"; }
for shape in "1135 40000000" "1250 40000000" "10000 20000000" "5000 40000000" "2500 40000000" "12500 16000000" "700 60000000"; do
  set -- $shape
  common="--n-acc $1 --n-snp $2 --mode fast --steps 6 --warmup 2 --no-cpu-baseline --no-alternatives --no-end-to-end"
  for a in 256 128; do
    SNPM_PITCH_ALIGN=$a timeout -k 10 200 python bench.py $common 2>/dev/null | line align=$a $1 $2
  done
done | tee $out/ab_pitch128.txt
