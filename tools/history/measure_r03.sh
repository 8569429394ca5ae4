#!/bin/bash
# Round-3 evidence: the default bench line, the same command under rocprofv3 (kernel stats + launches reduced per slab
# shape), the N>1 code path of bench.py on one GPU (RCCL communicator of one rank inside the library).
set -uo pipefail
out=gpurun_out/r03; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
echo "== default bench"
timeout -k 10 500 python bench.py --steps 20 --warmup 3 > $out/bench_n1.json 2> $out/bench_n1.err; echo "rc=$?"
echo "== rocprofv3 kernel trace of the same job"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_bench -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-end-to-end --no-alternatives > $out/bench_n1_under_rocprof.json 2> $out/prof_bench.err; echo "rc=$?"
python tools/kernel_trace_by_shape.py --phase-marker k_synth $out/prof_bench > $out/bench_kernel_by_shape.csv
find $out/prof_bench -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/bench_kernel_stats.csv
echo "== bench.py --force-dist (world 1, nccl): the gather through snpm_group_*"
timeout -k 10 300 python bench.py --steps 5 --warmup 1 --force-dist --no-cpu-baseline --no-alternatives --no-end-to-end --n-snp 6250000 > $out/bench_force_dist_c_abi.json 2> $out/bench_force_dist.err; echo "rc=$?"
find $out -name "*.db" -delete 2>/dev/null
rm -rf $out/prof_bench
echo done
