#!/bin/bash
# What a split layout (main part at a 256-B-multiple pitch + the ragged tail of a row at its own narrow pitch) could give a
# narrow packed panel, measured WITHOUT building it: the 1135-accession panel as two panels, 1024 accessions at the default
# 256-B pitch and the 111-accession tail at a 32-B pitch (SNPM_PITCH_ALIGN=32), each scanned on its own.  A split-layout
# kernel would take between max(t_main, t_tail) and t_main + t_tail.  Also narrow panels at narrow pitches (<= 1024 accessions).
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r04g; mkdir -p $out
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-22s %6s x %9s  %-18s %.3f ms  %.0f GB/s  frac %.4f' % ('$1','$2','$3', r['kernel'], r['avg_ms'], r['achieved'], r['frac']))"; }
run() {  # label align n_acc n_snp extra
  common="--n-acc $3 --n-snp $4 --mode fast --steps 6 --warmup 2 --no-cpu-baseline --no-alternatives --no-end-to-end --no-real-panel"
  SNPM_PITCH_ALIGN=$2 timeout -k 10 200 python bench.py --packed --hard-calls $common 2>/dev/null | line "hard-$1" $3 $4
  SNPM_PITCH_ALIGN=$2 timeout -k 10 200 python bench.py --packed $common 2>/dev/null | line "PL-$1" $3 $4
}
{
run align256 256 1135 40000000
run align256 256 1024 40000000
run tail-align32 32 111 40000000
run tail-align32 32 111 160000000
run align256 256 512 100000000
run align128 128 512 100000000
run align256 256 256 100000000
run align64 64 256 100000000
run align256 256 128 200000000
run align32 32 128 200000000
} | tee $out/split_layout_potential.txt
