#!/bin/bash
# Round 4, the round's last code: GPU suite, smoke, the default bench line, rocprofv3 kernel stats of the same job, PMC traffic of
# the headline slab (stamped with the library's build id), the one-call path's phases, the real-panel legs under rocprofv3.
# Run on the GPU box from the repo root; everything lands in gpurun_out/r04z/ and is copied to profiles/r04_*_final* by hand.
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r04z; mkdir -p $out
if [ "${SKIP_TESTS:-0}" != 1 ]; then
  echo "== gpu tests"; timeout -k 10 700 python -m pytest tests -x -q -m gpu > $out/gpu_tests.log 2>&1; echo "rc=$?"; tail -2 $out/gpu_tests.log
fi
echo "== smoke"; timeout -k 10 300 python __graft_entry__.py smoke > $out/smoke.log 2>&1; echo "rc=$?"
echo "== default bench"; timeout -k 10 500 python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "rc=$?"
echo "== rocprofv3 kernel trace of the same job"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_bench -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-end-to-end --no-alternatives --no-real-panel > $out/bench_under_rocprof.json 2> $out/prof_bench.err; echo "rc=$?"
python tools/kernel_trace_by_shape.py --phase-marker k_synth $out/prof_bench > $out/bench_kernel_by_shape.csv
find $out/prof_bench -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/bench_kernel_stats.csv
rm -rf $out/prof_bench
if [ "${SKIP_PMC:-0}" != 1 ]; then
echo "== PMC traffic of the headline slab"
bash tools/collect_pmc.sh slab_10000x20019000 10000 20019000 > $out/pmc_headline.log 2>&1; echo "rc=$?"; tail -3 $out/pmc_headline.log
cp gpurun_out/pmc_slab_10000x20019000/pmc_traffic.json $out/pmc_traffic.json 2>/dev/null
fi
echo "== one-call path: phases of its forms, GPU timeline"
python tools/debug/r04_once_trace.py 2>&1 | grep -v amdgpu > $out/once_forms.txt; grep "coded wall" $out/once_forms.txt | head -8 | tail -3
bash tools/debug/r04_once_timeline.sh > /dev/null 2>&1; cp gpurun_out/r04t/once_timeline.txt $out/once_timeline.txt 2>/dev/null
echo "== warm inbred / cross runs"
{ echo "== int8 panel"; python tools/time_inbred.py 2>&1 | grep -v amdgpu; echo "== 2-bit packed panel"; PACKED=1 python tools/time_inbred.py 2>&1 | grep -v amdgpu; } > $out/time_inbred.txt; grep rep1 $out/time_inbred.txt
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04z/bench_default.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("bench: %.3f ms/step  %.4g cmp/s  k_fast frac %.4f  all slabs %.4f  e2e %.4f  traffic %s" % (d["ms_per_step"], d["value"], r["frac"], r["all_slabs_frac"], r["end_to_end_frac"], r["traffic"]))
print("alternatives:", {k: (v.get("ms_per_step"), v.get("kernel_avg_ms")) for k, v in d.get("alternatives", {}).items()})
for f, v in d.get("real_panel", {}).get("formats", {}).items():
    for l in v["legs"]:
        print("  %-6s %-44s wall %.3f ms kernel %.4f ms frac %.3f" % (f, l["leg"], l["wall_ms_per_call"], l["kernel_ms_per_call"], l["frac_of_hbm_peak"]))
PY
