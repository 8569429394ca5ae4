#!/bin/bash
# rocprofv3 kernel stats of the shared-row batch scan (tools/time_shared.py) -> gpurun_out/$1 (default r05b)
# usage: tools/r05_prof_shared.sh [outdir] [time_shared.py arguments ...]
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${1:-r05b}; shift || true
mkdir -p $out
args="${*:-64 200000 5}"
tag=$(echo $args | tr ' ' '_')
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_$tag -- python3 tools/time_shared.py $args > $out/time_shared_$tag.log 2> $out/time_shared_$tag.err
echo "rc=$?"
f=$(find $out/trace_$tag -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp $f $out/shared_${tag}_kernel_stats.csv
t=$(find $out/trace_$tag -name '*kernel_trace.csv' | head -1)
[ -n "$t" ] && python tools/kernel_trace_summary.py $t > $out/shared_${tag}_kernel_by_grid.txt
rm -rf $out/trace_$tag
grep -v "^W2026\|^E2026" $out/time_shared_$tag.log
