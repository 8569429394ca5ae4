#!/bin/bash
# refresh of the packed-panel lines after the k_fast_packed_q4 tuning (same commands as tools/measure_r02b.sh)
set -uo pipefail
out=gpurun_out/r02b; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --packed > $out/bench_packed_full.json 2> $out/bench_packed.err; echo "rc=$?"
timeout -k 10 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --packed --hard-calls > $out/bench_packed_hardcalls_full.json 2> $out/bench_packed_hc.err; echo "rc=$?"
timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --mode strict --packed --n-snp 6250000 > $out/bench_strict_packed_10k_x_6250k.json 2> $out/bench_strict_packed.err; echo "rc=$?"
rm -rf $out/prof_packed
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_packed -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --packed > $out/bench_packed_under_rocprof.json 2> $out/prof_packed.err; echo "rc=$?"
find $out -name "*.db" -delete 2>/dev/null
echo done
