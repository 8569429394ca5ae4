#!/bin/bash
# timing experiment: k_fast_packed_q4 without the barrier between scoring and table build (wrong results): the most a
# second table set per block could save
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03k; mkdir -p $out
for shape in "10000 50000000" "8192 20000000" "1135 40000000" "4096 20000000"; do
  set -- $shape
  for lib in "" tools/ab/libsnpmatch_hip_onebar.so; do
    SNPMATCH_HIP_LIB=${lib:+$PWD/$lib} timeout -k 10 300 python bench.py --packed --n-acc $1 --n-snp $2 --mode fast --steps 6 --warmup 2 --no-cpu-baseline --no-alternatives --no-end-to-end 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-12s %6s x %9s  %.3f ms  frac %.4f' % ('${lib:+one-barrier}','$1','$2', r['avg_ms'], r['frac']))"
  done
done | tee $out/ab_q4_one_barrier.txt
