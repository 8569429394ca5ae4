#!/bin/bash
# the bench panel's row pitch (10 000 accessions: 10 240 B as shipped) against other paddings
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03k; mkdir -p $out
for a in 256 10112 10496 10752 11008 11264 12032; do
  SNPM_PITCH_ALIGN=$a timeout -k 10 200 python bench.py --n-acc 10000 --n-snp 17000000 --mode fast --steps 6 --warmup 2 --no-cpu-baseline --no-alternatives --no-end-to-end 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('pitch-align=%-6s  %.3f ms  frac %.4f (algorithmic bytes)' % ('$a', r['avg_ms'], r['frac']))"
done | tee $out/ab_pitch_10000.txt
