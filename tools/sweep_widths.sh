#!/bin/bash
# k_fast / k_fast_packed_q4 / k_fast_bits on a sweep of panel widths (fast mode, dense scan, ~12-60 GB per panel)
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03h; mkdir -p $out
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-12s %6s x %9s  %-18s %.3f ms  %.0f GB/s  frac %.4f  step %.3f ms' % ('$1','$2','$3', r['kernel'], r['avg_ms'], r['achieved'], r['frac'], d['ms_per_step']))"; }
for shape in "256 100000000" "512 100000000" "1135 40000000" "1135 11000000" "2029 30000000" "3000 20000000" "5000 12000000"; do
  set -- $shape
  common="--n-acc $1 --n-snp $2 --mode fast --steps 6 --warmup 2 --no-cpu-baseline --no-alternatives --no-end-to-end"
  timeout -k 10 200 python bench.py $common 2>/dev/null | line int8 $1 $2
  timeout -k 10 200 python bench.py --packed $common 2>/dev/null | line packed-PL $1 $2
  timeout -k 10 200 python bench.py --packed --hard-calls $common 2>/dev/null | line packed-hard $1 $2
done | tee $out/sweep_widths.txt
