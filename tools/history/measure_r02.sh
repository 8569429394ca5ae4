#!/bin/bash
# Round-2 evidence run (on the GPU box, from the repo root): bench lines of every shape DESIGN.md quotes, rocprofv3 kernel
# stats of the default bench, PMC traffic of the dominant slab shape.  Everything lands under gpurun_out/r02/.
set -uo pipefail
out=gpurun_out/r02; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
step() { echo "== $*"; }
step "default bench (N=1: whole 10k x 50M job in slabs)"
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $out/bench_n1.json 2> $out/bench_n1.err; echo "rc=$?"
step "per-GPU shapes of N=2,4,8 on one GPU"
for na in 5000 2500 1252; do
  timeout -k 10 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --n-acc $na > $out/bench_shape_${na}x50M.json 2> $out/bench_shape_${na}.err; echo "$na rc=$?"
done
step "config 5 slab 12500 x 16M"
timeout -k 10 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --n-acc 12500 --n-snp 16000000 > $out/bench_config5_slab_12500x16M.json 2> $out/bench_config5.err; echo "rc=$?"
step "strict mode (reference order everywhere), 10k x 6.25M"
timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --mode strict --n-snp 6250000 > $out/bench_strict_10k_x_6250k.json 2> $out/bench_strict.err; echo "rc=$?"
step "packed panel: whole job resident"
timeout -k 10 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --packed > $out/bench_packed_full.json 2> $out/bench_packed.err; echo "rc=$?"
timeout -k 10 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --packed --hard-calls > $out/bench_packed_hardcalls_full.json 2> $out/bench_packed_hc.err; echo "rc=$?"
step "small-query / pipeline / batch timings"
timeout -k 10 200 python tools/time_small_query.py > $out/time_small_query.log 2>&1; echo "rc=$?"
timeout -k 10 200 python tools/time_inbred.py > $out/time_inbred.log 2>&1; echo "rc=$?"
timeout -k 10 200 python tools/time_f1_pairs.py > $out/time_f1_pairs.log 2>&1; echo "rc=$?"
timeout -k 10 300 python tools/time_batch.py 64 4 > $out/time_batch_64.log 2>&1; echo "rc=$?"
step "rocprofv3 kernel stats of the default bench"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_bench -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-end-to-end > $out/bench_n1_under_rocprof.json 2> $out/prof_bench.err; echo "rc=$?"
step "PMC traffic of the dominant slab shape"
PMC_KEY=slab bash tools/collect_pmc.sh slab_10000x20019000 10000 20019000 > $out/pmc_slab.log 2>&1; echo "rc=$?"
echo done
