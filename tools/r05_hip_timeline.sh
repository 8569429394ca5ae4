#!/bin/bash
# host-side timeline (HIP API calls + kernels) of the LAST shared-row call of tools/time_shared.py -> gpurun_out/$1/shared_hip_timeline.txt
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${1:-r05m}; shift || true
mkdir -p $out
args="${*:-64 200000 3}"
timeout -k 10 500 rocprofv3 --hip-trace --kernel-trace --output-format csv -d $out/hipt -- python3 tools/time_shared.py $args > $out/hipt.log 2> $out/hipt.err
echo "rc=$?"
api=$(find $out/hipt -name '*hip_api_trace.csv' | head -1)
ker=$(find $out/hipt -name '*kernel_trace.csv' | head -1)
python3 - "$api" "$ker" > $out/shared_hip_timeline.txt <<'PY'
import csv, sys
api = list(csv.DictReader(open(sys.argv[1])))
ker = list(csv.DictReader(open(sys.argv[2])))
ev = []
for r in api:
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "api", r["Function"]))
for r in ker:
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "gpu", r["Kernel_Name"].split("(")[0][:60]))
ev.sort()
# the last k_sh_mark marks the last shared call; print everything from 300 us before it to the end of the following k_likelihood + 300 us
marks = [e for e in ev if e[2] == "gpu" and "k_sh_mark" in e[3]]
t0 = marks[-1][0] - 150_000
liks = [e for e in ev if e[2] == "gpu" and "k_likelihood" in e[3] and e[0] > marks[-1][0]]
t1 = liks[-1][1] + 400_000
for s, e, kind, name in ev:
    if t0 <= s <= t1:
        print("%9.1f us  %-4s %8.1f us  %s" % ((s - t0) / 1e3, kind, (e - s) / 1e3, name))
PY
rm -rf $out/hipt
tail -120 $out/shared_hip_timeline.txt
