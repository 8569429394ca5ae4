#!/bin/bash
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r04f; mkdir -p $out
timeout -k 10 600 python -m pytest tests -x -q -m gpu > $out/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 $out/gpu_tests.log
timeout -k 10 400 python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc=$?"
timeout -k 10 200 python bench.py --force-dist --n-snp 6250000 --steps 5 --warmup 1 --no-cpu-baseline > $out/bench_force_dist.json 2> $out/bench_force_dist.err; echo "force-dist rc=$?"
SNPM_BENCH_SIMULATE_STUCK_JOIN=0 timeout -k 10 200 python bench.py --force-dist --group-timeout 5 --n-snp 6250000 --steps 5 --warmup 1 --no-cpu-baseline > $out/bench_stuck_join.json 2> $out/bench_stuck_join.err; echo "stuck-join rc=$? (3 expected)"
python - <<'PY'
import json
for f in ("bench_default", "bench_force_dist", "bench_stuck_join"):
    try:
        d = json.loads(open("gpurun_out/r04f/%s.json" % f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "no line:", e); continue
    print(f, "value %.4g ms/step %.3f frac %.4f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"]), d["config"].get("collective"), d["checks"], d.get("per_rank"))
    if "real_panel" in d:
        rp = d["real_panel"]
        if "error" in rp: print("  real_panel error", rp["error"])
        else:
            for fm, v in rp["formats"].items():
                for l in v["legs"]:
                    print("   %-6s %-44s wall %.3f kernel %.4f ms frac %.3f" % (fm, l["leg"], l["wall_ms_per_call"], l["kernel_ms_per_call"], l["frac_of_hbm_peak"]))
    print("  traffic", d["roofline"]["traffic"], d["roofline"]["traffic_source"])
PY
