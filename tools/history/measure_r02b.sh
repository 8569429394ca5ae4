#!/bin/bash
# Round-2 evidence, second pass (after the k_strict4 / k_fast_bits / k_fast_packed_q4 changes): bench lines of the default
# job and of the strict and packed variants, rocprofv3 kernel stats of the default bench and of the strict / packed runs.
set -uo pipefail
out=gpurun_out/r02b; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
step() { echo "== $*"; }
step "default bench (N=1: whole 10k x 50M job in slabs)"
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $out/bench_n1.json 2> $out/bench_n1.err; echo "rc=$?"
step "strict mode, int8 and packed, 10k x 6.25M"
timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --mode strict --n-snp 6250000 > $out/bench_strict_10k_x_6250k.json 2> $out/bench_strict.err; echo "rc=$?"
timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --mode strict --packed --n-snp 6250000 > $out/bench_strict_packed_10k_x_6250k.json 2> $out/bench_strict_packed.err; echo "rc=$?"
step "packed panel: whole job resident, PL weights and hard calls"
timeout -k 10 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --packed > $out/bench_packed_full.json 2> $out/bench_packed.err; echo "rc=$?"
timeout -k 10 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --packed --hard-calls > $out/bench_packed_hardcalls_full.json 2> $out/bench_packed_hc.err; echo "rc=$?"
step "rocprofv3 kernel stats: default bench, strict int8, packed PL, packed hard calls"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_bench -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-end-to-end > $out/bench_n1_under_rocprof.json 2> $out/prof_bench.err; echo "rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_strict -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --mode strict --n-snp 6250000 > $out/bench_strict_under_rocprof.json 2> $out/prof_strict.err; echo "rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_packed -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --packed > $out/bench_packed_under_rocprof.json 2> $out/prof_packed.err; echo "rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_packed_hc -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --packed --hard-calls > $out/bench_packed_hc_under_rocprof.json 2> $out/prof_packed_hc.err; echo "rc=$?"
step "read-only ceiling and the pattern-only build of k_fast (if present)"
timeout -k 10 200 python tools/read_ceiling.py > $out/read_ceiling.txt 2>&1; echo "rc=$?"
if [ -f tools/ab/libsnpmatch_pattern.so ]; then
  timeout -k 10 200 python tools/ab/ab_fast.py tools/ab/libsnpmatch_pattern.so > $out/pattern_only_k_fast.txt 2>&1; echo "rc=$?"
  timeout -k 10 200 python tools/ab/ab_fast.py snpmatch_amd/libsnpmatch_hip.so >> $out/pattern_only_k_fast.txt 2>&1; echo "rc=$?"
fi
find $out -name "*.db" -delete 2>/dev/null
echo done
