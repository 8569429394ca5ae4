#!/bin/bash
# the three fast kernels on wide panels (fast mode, dense scan)
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03k; mkdir -p $out
for shape in "4096 20000000" "6144 30000000" "8192 20000000" "10000 20000000" "12288 20000000" "16384 15000000" "20000 10000000"; do
  set -- $shape
  for fmt in "" "--packed" "--packed --hard-calls"; do
    timeout -k 10 200 python bench.py $fmt --n-acc $1 --n-snp $2 --mode fast --steps 6 --warmup 2 --no-cpu-baseline --no-alternatives --no-end-to-end 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-22s %6s x %9s  %-18s %.3f ms  frac %.4f  step %.3f ms' % ('${fmt:-int8}','$1','$2', r['kernel'], r['avg_ms'], r['frac'], d['ms_per_step']))"
  done
done | tee $out/sweep_wide_final.txt
