#!/bin/bash
# Round-3 evidence, final pass: smoke, the default bench line, the same job under rocprofv3 (pooled stats + per-slab summary),
# the packed variants, strict mode, bench.py's N>1 path with the library's communicator, PMC traffic of the dominant shapes.
set -uo pipefail
out=gpurun_out/${SNPM_MEASURE_TAG:-r03b}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
echo "== smoke"; timeout -k 10 300 python __graft_entry__.py smoke > $out/smoke.log 2>&1; echo "rc=$?"
echo "== default bench"
timeout -k 10 500 python bench.py --steps 20 --warmup 3 > $out/bench_n1.json 2> $out/bench_n1.err; echo "rc=$?"
echo "== rocprofv3 kernel trace of the same job"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_bench -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-end-to-end --no-alternatives > $out/bench_n1_under_rocprof.json 2> $out/prof_bench.err; echo "rc=$?"
python tools/kernel_trace_by_shape.py --phase-marker k_synth $out/prof_bench > $out/bench_kernel_by_shape.csv
find $out/prof_bench -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/bench_kernel_stats.csv
rm -rf $out/prof_bench
echo "== packed panel (PL weights, hard calls), strict mode"
timeout -k 10 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-alternatives --packed > $out/bench_packed_full.json 2> $out/bench_packed.err; echo "rc=$?"
timeout -k 10 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-alternatives --packed --hard-calls > $out/bench_packed_hardcalls_full.json 2> $out/bench_packed_hc.err; echo "rc=$?"
timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-alternatives --mode strict --n-snp 6250000 > $out/bench_strict_10k_x_6250k.json 2> $out/bench_strict.err; echo "rc=$?"
echo "== bench.py --force-dist (world 1, nccl)"
timeout -k 10 300 python bench.py --steps 5 --warmup 1 --force-dist --no-cpu-baseline --no-alternatives --no-end-to-end --n-snp 6250000 > $out/bench_force_dist_c_abi.json 2> $out/bench_force_dist.err; echo "rc=$?"
echo "== PMC traffic: the bench slab, the config-5 slab, the packed panel"
bash tools/collect_pmc.sh slab_10000x20019000 10000 20019000 > $out/pmc_slab.log 2>&1; echo "rc=$?"
cp gpurun_out/pmc_slab_10000x20019000/pmc_traffic.json profiles/pmc_traffic.json 2>/dev/null
bash tools/collect_pmc.sh slab_12500x16342000 12500 16342000 > $out/pmc_config5.log 2>&1; echo "rc=$?"
cp gpurun_out/pmc_slab_12500x16342000/pmc_traffic.json profiles/pmc_traffic.json 2>/dev/null
PMC_PACKED=1 bash tools/collect_pmc.sh packed_q4_10000x20000000 10000 20000000 > $out/pmc_packed.log 2>&1; echo "rc=$?"
cp gpurun_out/pmc_packed_q4_10000x20000000/pmc_traffic.json profiles/pmc_traffic.json 2>/dev/null
cp profiles/pmc_traffic.json $out/pmc_traffic.json
find gpurun_out -name "*.db" -delete 2>/dev/null
rm -rf gpurun_out/pmc_*/fetch gpurun_out/pmc_*/write
echo done
