#!/bin/bash
# k_fast_packed_q4: tiles a part scores in a row (Q4_RUN: 1, 2, 4 as shipped, 8) under the parts rule
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03j; mkdir -p $out
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-8s %6s x %9s  %-18s %.3f ms  frac %.4f  step %.3f' % ('$1','$2','$3', r['kernel'], r['avg_ms'], r['frac'], d['ms_per_step']))"; }
for shape in "10000 50000000" "8192 20000000" "1135 40000000" "2400 20000000"; do
  set -- $shape
  common="--n-acc $1 --n-snp $2 --mode fast --steps 6 --warmup 2 --no-cpu-baseline --no-alternatives --no-end-to-end --packed"
  for v in run1 run2 "" run8; do
    SNPMATCH_HIP_LIB=${v:+$PWD/tools/ab/libsnpmatch_hip_$v.so} timeout -k 10 300 python bench.py $common 2>/dev/null | line ${v:-run4} $1 $2
  done
done | tee $out/ab_q4_run.txt
