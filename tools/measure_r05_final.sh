#!/bin/bash
# Round 5, the round's last code: GPU suite, smoke, the default bench line (with its real_panel / staging children), rocprofv3 kernel
# stats of the same job, PMC traffic of the headline slab (stamped with the library's build id), the real-panel legs and the
# shared-row scan under rocprofv3, the one-call path's timeline, and the figures DESIGN.md still quoted from round 3: the per-GPU
# shards of N = 8 / 4 / 2, k_strict4 on 10 000 x 6.25M, the configs[4] share looped in 7 slabs, the packed full job.
# Run on the GPU box from the repo root; everything lands in gpurun_out/r05z/ (copied to profiles/r05_* by hand).
# PARTS selects what runs: tests smoke bench prof pmc panel shared once shards strict config4 packed (default: all)
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r05z; mkdir -p $out
PARTS=${PARTS:-"tests smoke bench prof pmc panel shared once shards strict config4 packed"}
has() { [[ " $PARTS " == *" $1 "* ]]; }
if has tests; then echo "== gpu tests"; timeout -k 10 900 python -m pytest tests -q -m gpu > $out/gpu_tests.log 2>&1; echo "rc=$?"; tail -2 $out/gpu_tests.log; fi
if has smoke; then echo "== smoke"; timeout -k 10 300 python __graft_entry__.py smoke > $out/smoke.log 2>&1; echo "rc=$?"; tail -1 $out/smoke.log; fi
if has bench; then echo "== default bench"; timeout -k 10 700 python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "rc=$?"; fi
if has prof; then
  echo "== rocprofv3 kernel trace of the same job"
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_bench -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-end-to-end --no-alternatives --no-real-panel --no-staging > $out/bench_under_rocprof.json 2> $out/prof_bench.err; echo "rc=$?"
  python tools/kernel_trace_by_shape.py --phase-marker k_synth $out/prof_bench > $out/bench_kernel_by_shape.csv
  find $out/prof_bench -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/bench_kernel_stats.csv
  rm -rf $out/prof_bench
fi
if has pmc; then
  echo "== PMC traffic of the headline slab"
  bash tools/collect_pmc.sh slab_10000x20019000 10000 20019000 > $out/pmc_headline.log 2>&1; echo "rc=$?"; tail -3 $out/pmc_headline.log
  cp gpurun_out/pmc_slab_10000x20019000/pmc_traffic.json $out/pmc_traffic.json 2>/dev/null
fi
if has panel; then
  echo "== real-panel legs under rocprofv3"
  for fmt in int8 packed; do
    timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_$fmt -- python3 tools/bench_real_panel.py --formats $fmt --reps 10 > $out/real_panel_$fmt.json 2> $out/real_panel_$fmt.err
    f=$(find $out/trace_$fmt -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $out/real_panel_${fmt}_kernel_stats.csv
    t=$(find $out/trace_$fmt -name '*kernel_trace.csv' | head -1); [ -n "$t" ] && python tools/kernel_trace_summary.py $t > $out/real_panel_${fmt}_kernel_by_grid.txt
    rm -rf $out/trace_$fmt
  done
fi
if has shared; then
  echo "== shared-row scan under rocprofv3 (int8, packed)"
  bash tools/r05_prof_shared.sh r05z 64 200000 10 | grep -v "^rc"
  bash tools/r05_prof_shared.sh r05z 64 200000 10 0 1 | grep -v "^rc"
  echo "== shared-row scan: batch sizes, digits, overlap"
  { for b in 8 16 32 64 128 256; do echo "-- B=$b"; python tools/time_shared.py $b 200000 6 2>&1 | grep "pass\|scan"; done
    for dg in 5 6 7; do echo "-- digits=$dg (B=64)"; python tools/time_shared.py 64 200000 6 $dg 2>&1 | grep "scan"; done
    for drop in 0.3 0.6 0.8; do echo "-- each sample lacks $drop of the marker set (B=64)"; python tools/time_shared.py 64 200000 6 0 0 $drop 2>&1 | grep "pass\|scan"; done
  } > $out/shared_sweeps.txt 2>&1; tail -4 $out/shared_sweeps.txt
fi
if has once; then
  echo "== one-call path: GPU timeline"
  bash tools/once_timeline.sh > /dev/null 2>&1; cp gpurun_out/once_timeline/once_timeline.txt $out/once_timeline.txt 2>/dev/null; tail -7 $out/once_timeline.txt
fi
if has shards; then
  echo "== per-GPU shards of N = 8 / 4 / 2"
  for shape in "1250 8" "2500 4" "5000 2"; do
    set -- $shape
    timeout -k 10 400 python bench.py --n-acc $1 --n-snp 50000000 --steps 10 --warmup 2 --no-cpu-baseline --no-alternatives --no-real-panel --no-staging > $out/bench_shard_${1}x50M_n$2.json 2>$out/bench_shard_$1.err
    python - <<PY
import json
d=json.loads(open("$out/bench_shard_${1}x50M_n$2.json").read().strip().splitlines()[-1]); r=d['roofline']
print("%5d x 50M (one of $2 GPUs): step %.3f ms  k_fast %.3f ms  frac %.4f  e2e frac %s  reeval %s  slabs %s  -> N=$2 job value ~ %.3e" % ($1, d['ms_per_step'], r['avg_ms'], r['frac'], r.get('end_to_end_frac'), d['checks']['strict_reevaluations'], d['config']['slabs'], 1e4*5e7/(d['ms_per_step']*1e-3)))
PY
  done | tee $out/shard_shapes.txt
fi
if has strict; then
  echo "== k_strict4: 10 000 x 6.25M in reference order"
  timeout -k 10 400 python bench.py --n-snp 6250000 --mode strict --steps 5 --warmup 1 --no-cpu-baseline --no-alternatives --no-real-panel --no-staging > $out/bench_strict_10k_x_6250k.json 2> $out/bench_strict.err; echo "rc=$?"
fi
if has config4; then
  echo "== configs[4] share: 12 500 x 100M looped in slabs"
  timeout -k 10 600 python bench.py --n-acc 12500 --n-snp 100000000 --steps 3 --warmup 1 --no-cpu-baseline --no-alternatives --no-real-panel --no-staging --no-end-to-end > $out/bench_config5_looped.json 2> $out/bench_config5.err; echo "rc=$?"
fi
if has packed; then
  echo "== packed panel, whole job resident: PL weights / hard calls"
  timeout -k 10 400 python bench.py --packed --steps 8 --warmup 2 --no-cpu-baseline --no-alternatives --no-real-panel --no-staging > $out/bench_packed_full.json 2> $out/bench_packed.err; echo "rc=$?"
  timeout -k 10 400 python bench.py --packed --hard-calls --steps 8 --warmup 2 --no-cpu-baseline --no-alternatives --no-real-panel --no-staging > $out/bench_packed_hardcalls_full.json 2> $out/bench_packed_hard.err; echo "rc=$?"
fi
python - <<'PY'
import json, os
o = "gpurun_out/r05z/"
def last(p):
    try: return json.loads(open(o + p).read().strip().splitlines()[-1])
    except Exception as e: return None
d = last("bench_default.json")
if d:
    r = d["roofline"]
    print("bench: %.3f ms/step  %.4g cmp/s  k_fast frac %.4f  all slabs %.4f  e2e %.4f  traffic %s  build %s" % (d["ms_per_step"], d["value"], r["frac"], r["all_slabs_frac"], r["end_to_end_frac"], r["traffic"], d.get("library_build_id")))
    for f, v in d.get("real_panel", {}).get("formats", {}).items():
        for l in v["legs"]:
            print("  %-6s %-44s wall %.3f ms kernel %.4f ms frac %.3f %s" % (f, l["leg"], l["wall_ms_per_call"], l["kernel_ms_per_call"], l["frac_of_hbm_peak"], ("%.0f samples/s" % l["samples_per_s"]) if l.get("samples_per_s") else ""))
    st = d.get("staging", {})
    print("  staging:", [(l["leg"], round(l.get("int8_GBs", 0), 1)) for l in st.get("legs", [])], "ceiling", st.get("hipMemcpyAsync_ceiling_GBs"), "overlap", st.get("overlap", {}).get("efficiency"))
for name in ("bench_strict_10k_x_6250k.json", "bench_config5_looped.json", "bench_packed_full.json", "bench_packed_hardcalls_full.json"):
    d = last(name)
    if d: print("%s: %.3f ms/step %.4g cmp/s kernel %s %.3f ms frac %.4f" % (name, d["ms_per_step"], d["value"], d["roofline"]["kernel"], d["roofline"]["avg_ms"], d["roofline"]["frac"]))
PY
