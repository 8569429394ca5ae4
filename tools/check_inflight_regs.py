#!/usr/bin/env python3
"""Static check of the kernels that issue LDS reads from inline asm (k_scan_few): between a
ds_read_b128 and the s_waitcnt that covers it, no other instruction may touch its destination registers (the
compiler does not know they are in flight).  Input: the gfx950 .s file written by SNPM_SAVE_TEMPS=1 ./build_lib.sh.
LDS reads return in order, so after `s_waitcnt lgkmcnt(N)` at most the N most recent reads are still pending
(scalar loads also count towards lgkmcnt, which only makes the real wait stricter than assumed here)."""
import re
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "build/snpm_api-hip-amdgcn-amd-amdhsa-gfx950.s"
reg_range = re.compile(r"v\[(\d+):(\d+)\]")
reg_one = re.compile(r"(?<![\w\[])v(\d+)\b")


def regs_of(text):
    out = set()
    for a, b in reg_range.findall(text):
        out.update(range(int(a), int(b) + 1))
    for a in reg_one.findall(text):
        out.add(int(a))
    return out


bad = 0
checked = 0
name = None
pending = []          # list of register sets, oldest first
for line in open(path):
    m = re.match(r"^(_ZN4snpm\w+):", line)
    if m:
        name, pending = m.group(1), []
        continue
    if name is None or not ("k_scan_few" in name):
        continue
    t = line.strip()
    if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
        if t.endswith(":"):
            # a label: control flow may merge here; reads still pending at a label are carried conservatively
            pass
        continue
    t = t.split(";")[0].strip()
    op = t.split()[0]
    if op == "s_endpgm":
        name = None
        continue
    if op == "ds_read_b128":
        dst = regs_of(t.split(",")[0])
        addr = regs_of(",".join(t.split(",")[1:]))
        for p in pending:
            if p & (dst | addr):
                bad += 1
                print("%s: LDS read touches in-flight registers: %s" % (name, t))
        pending.append(dst)
        checked += 1
        continue
    if op == "s_waitcnt":
        m = re.search(r"lgkmcnt\((\d+)\)", t)
        if m:
            n = int(m.group(1))
            pending = pending[len(pending) - n:] if n else []
        continue
    if op in ("s_barrier",):
        continue
    used = regs_of(t)
    for p in pending:
        if p & used:
            bad += 1
            print("%s: instruction touches in-flight registers: %s" % (name, t))
print("checked %d asm LDS reads, %d violations" % (checked, bad))
sys.exit(1 if bad else 0)
