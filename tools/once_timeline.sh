#!/bin/bash
# GPU timeline (kernels + copies) of the last snpm_genotype_once[_coded] calls of tools/once_trace.py
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/once_timeline; mkdir -p $out
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out/trace -- python3 tools/once_trace.py ${ONCE_TIMELINE_FORM:-default} > $out/once_trace.txt 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys
out = sys.argv[1]
ev = []
for f in glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:90], "grid %s wg %s" % (r.get("Grid_Size_X", r.get("Grid_Size")), r.get("Workgroup_Size_X", r.get("Workgroup_Size")))))
for f in glob.glob(out + "/trace/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", ""), ""))
ev.sort()
# calls are separated by gaps > 150 us; print the last two groups of each half
groups, cur = [], []
for e in ev:
    if cur and e[0] - cur[-1][1] > 150_000:
        groups.append(cur); cur = []
    cur.append(e)
if cur: groups.append(cur)
with open(out + "/once_timeline.txt", "w") as fh:
    for gi in (len(groups) - 10, len(groups) - 9, len(groups) - 2, len(groups) - 1):
        if gi < 0: continue
        g = groups[gi]
        fh.write("== call group %d: %d events, span %.1f us\n" % (gi, len(g), (g[-1][1] - g[0][0]) / 1e3))
        for s, e, name, extra in g:
            fh.write("  +%7.1f us  %6.1f us  %s  %s\n" % ((s - g[0][0]) / 1e3, (e - s) / 1e3, name, extra))
PY
rm -rf $out/trace
tail -5 $out/once_trace.txt
