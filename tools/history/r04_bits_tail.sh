#!/bin/bash
# same-box A/B of k_fast_bits on ragged packed panels: split layout on/off x light tail on/off
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r04n; mkdir -p $out
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-26s %6s x %9s  %-14s %.3f ms  frac %.4f' % ('$1','$2','$3', r['kernel'], r['avg_ms'], r['frac']))"; }
{
for shape in "1135 40000000" "1040 40000000" "2100 30000000" "3000 20000000" "4400 12000000"; do
  set -- $shape
  common="--n-acc $1 --n-snp $2 --mode fast --steps 6 --warmup 2 --no-cpu-baseline --no-alternatives --no-end-to-end --no-real-panel"
  for rep in 1 2; do
  for cfg in "0 0" "1 0" "1 1"; do
    set -- $shape $cfg
    SNPM_PACKED_SPLIT=$3 SNPM_BITS_LIGHT_TAIL=$4 timeout -k 10 200 python bench.py --packed --hard-calls $common 2>/dev/null | line "split=$3 light_tail=$4" $1 $2
  done
  done
done
} | tee $out/bits_light_tail_ab.txt
