#!/bin/bash
# same-box A/B of k_fast_bits' grid order: the part as the fast block index (default) against the column block (rounds 1-3),
# on whole-row (SNPM_PACKED_SPLIT=0) and split packed panels
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r04n; mkdir -p $out
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-28s %6s x %9s  %-14s %.3f ms  frac %.4f' % ('$1','$2','$3', r['kernel'], r['avg_ms'], r['frac']))"; }
{
for shape in "1024 40000000" "1135 40000000" "1536 40000000" "2100 30000000" "3000 20000000" "3500 20000000" "4400 12000000" "8000 10000000" "10000 20000000"; do
  set -- $shape
  common="--n-acc $1 --n-snp $2 --mode fast --steps 6 --warmup 2 --no-cpu-baseline --no-alternatives --no-end-to-end --no-real-panel"
  for cfg in "0 0" "0 1" "1 0" "1 1"; do
    set -- $shape $cfg
    SNPM_PACKED_SPLIT=$3 SNPM_BITS_PART_FAST=$4 timeout -k 10 200 python bench.py --packed --hard-calls $common 2>/dev/null | line "split=$3 part_fast=$4" $1 $2
  done
done
} | tee $out/bits_grid_order_ab.txt
