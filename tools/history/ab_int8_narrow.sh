#!/bin/bash
# int8 k_fast on panels of one or two waves: LUT tile size (248 rows on long scans vs 128) and resident blocks
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03j; mkdir -p $out
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-30s %6s x %9s  %-8s %.3f ms  %.0f GB/s  frac %.4f' % ('$1','$2','$3', r['kernel'], r['avg_ms'], r['achieved'], r['frac']))"; }
for shape in "256 100000000" "512 100000000" "700 60000000"; do
  set -- $shape
  common="--n-acc $1 --n-snp $2 --mode fast --steps 6 --warmup 2 --no-cpu-baseline --no-alternatives --no-end-to-end"
  SNPMATCH_X=1 timeout -k 10 200 python bench.py $common 2>/dev/null | line default $1 $2
  SNPM_LONG_SCAN_ROWS=100000000000 timeout -k 10 200 python bench.py $common 2>/dev/null | line tiles128 $1 $2
  for m in 2 4; do
    SNPM_PARTS_MULT=$m timeout -k 10 200 python bench.py $common 2>/dev/null | line parts_mult=$m $1 $2
    SNPM_LONG_SCAN_ROWS=100000000000 SNPM_PARTS_MULT=$m timeout -k 10 200 python bench.py $common 2>/dev/null | line tiles128,parts_mult=$m $1 $2
  done
  SNPM_FULL_OCCUPANCY=1 timeout -k 10 200 python bench.py $common 2>/dev/null | line full_occupancy $1 $2
done | tee $out/ab_int8_narrow.txt
