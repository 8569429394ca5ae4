#!/bin/bash
# channel balance of narrow packed panels: row pitch 512 B (284 used) against 768 B (an odd number of 256-B blocks)
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03h; mkdir -p $out
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-22s %6s x %9s  %-18s %.3f ms  %.0f GB/s  frac %.4f  step %.3f' % ('$1','$2','$3', r['kernel'], r['avg_ms'], r['achieved'], r['frac'], d['ms_per_step']))"; }
for shape in "1135 40000000" "1040 40000000" "1536 40000000"; do
  set -- $shape
  common="--n-acc $1 --n-snp $2 --mode fast --steps 6 --warmup 2 --no-cpu-baseline --no-alternatives --no-end-to-end"
  for a in 256 768 1280; do
    SNPM_PITCH_ALIGN=$a timeout -k 10 200 python bench.py --packed --hard-calls $common 2>/dev/null | line hard-align$a $1 $2
    SNPM_PITCH_ALIGN=$a timeout -k 10 200 python bench.py --packed $common 2>/dev/null | line PL-align$a $1 $2
  done
  SNPM_PITCH_ALIGN=1280 timeout -k 10 200 python bench.py $common 2>/dev/null | line int8-align1280 $1 $2
  SNPM_PITCH_ALIGN=256 timeout -k 10 200 python bench.py $common 2>/dev/null | line int8-align256 $1 $2
done | tee $out/ab_pitch_odd.txt
