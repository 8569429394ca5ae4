#!/bin/bash
# after the register work on k_fast_packed_q4 (no spills): width sweep, the wide packed panel, the default bench line
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03h; mkdir -p $out
bash tools/sweep_widths.sh > /dev/null
cp $out/sweep_widths.txt $out/sweep_widths_nospill.txt
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-14s %6s x %9s  %-18s %.3f ms  %.0f GB/s  frac %.4f  step %.3f ms' % ('$1','$2','$3', r['kernel'], r['avg_ms'], r['achieved'], r['frac'], d['ms_per_step']))"; }
for shape in "10000 50000000" "8192 50000000" "16384 30000000" "6144 40000000"; do
  set -- $shape
  timeout -k 10 300 python bench.py --packed --n-acc $1 --n-snp $2 --mode fast --steps 6 --warmup 2 --no-cpu-baseline --no-alternatives --no-end-to-end 2>/dev/null | line packed-PL $1 $2
done | tee $out/wide_packed_nospill.txt
timeout -k 10 600 python bench.py --no-cpu-baseline > $out/bench_default_nospill.json 2>$out/bench_default_nospill.err
tail -c 1500 $out/bench_default_nospill.json
