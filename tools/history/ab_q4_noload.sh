#!/bin/bash
# timing experiment: k_fast_packed_q4 with its row loads replaced by register arithmetic (wrong results): the floor set by
# table builds, lookups, counts and barriers alone
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03k; mkdir -p $out
for shape in "10000 50000000" "8192 20000000" "1135 40000000" "4096 20000000"; do
  set -- $shape
  for lib in "" tools/ab/libsnpmatch_hip_noload.so; do
    SNPMATCH_HIP_LIB=${lib:+$PWD/$lib} timeout -k 10 300 python bench.py --packed --n-acc $1 --n-snp $2 --mode fast --steps 6 --warmup 2 --no-cpu-baseline --no-alternatives --no-end-to-end 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-12s %6s x %9s  %.3f ms  frac %.4f' % ('${lib:+no-loads}','$1','$2', r['avg_ms'], r['frac']))"
  done
done | tee $out/ab_q4_noload.txt
