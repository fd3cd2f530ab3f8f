#!/usr/bin/env python3
"""Static view of one kernel's ISA: instructions per basic block by class (VALU / SALU / DS / VMEM / SMEM / branch), so that the tile loop
of a record kernel can be read without a GPU.  scripts/isa_blocks.py <file.s> <mangled-name-substring> [--dump LABEL]"""
import re, sys
src, pat = sys.argv[1], sys.argv[2]
dump = sys.argv[4] if len(sys.argv) > 4 and sys.argv[3] == "--dump" else None
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*" + re.escape(pat) + r"\S*:", l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
blocks, cur = [], ["entry", []]
for l in lines[start + 1:end]:
    m = re.match(r"^(\.LBB\S+):", l)
    if m:
        blocks.append(cur); cur = [m.group(1), []]
        continue
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."): continue
    cur[1].append(t)
blocks.append(cur)
def cls(op):
    if op.startswith("v_"): return "VALU"
    if op.startswith("ds_"): return "DS"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")): return "VMEM"
    if op.startswith(("s_load", "s_buffer_load", "s_store")): return "SMEM"
    if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")): return "BR"
    if op.startswith(("s_waitcnt", "s_nop")): return "WAIT"
    return "SALU"
tot = {}
for name, ins in blocks:
    c = {}
    for t in ins:
        k = cls(t.split()[0]); c[k] = c.get(k, 0) + 1; tot[k] = tot.get(k, 0) + 1
    tgt = [t.split()[-1] for t in ins if t.startswith(("s_cbranch", "s_branch"))]
    if dump is None:
        print("%-14s %4d  %s  -> %s" % (name, len(ins), " ".join("%s=%d" % kv for kv in sorted(c.items())), ",".join(tgt)))
    elif name == dump:
        print("\n".join(ins))
if dump is None: print("TOTAL", tot)
