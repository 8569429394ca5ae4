#!/bin/bash
# k_fast_packed_q4 block sizes again, under the parts rule (up to 16 parts per resident block)
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03j; mkdir -p $out
for shape in "10000 20000000" "12288 20000000" "5000 20000000" "6144 30000000" "20000 10000000"; do
  set -- $shape
  for w in 0 4 5 6 7 8; do
    SNPM_FORCE_WPB=$w timeout -k 10 200 python bench.py --packed --n-acc $1 --n-snp $2 --mode fast --steps 6 --warmup 2 --no-cpu-baseline --no-alternatives --no-end-to-end 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('wpb=$w  %6s x %9s  %.3f ms  frac %.4f  step %.3f' % ('$1','$2', r['avg_ms'], r['frac'], d['ms_per_step']))"
  done
done | tee $out/ab_q4_wpb5.txt
