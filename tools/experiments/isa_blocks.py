#!/usr/bin/env python3
"""Per-basic-block instruction mix of one kernel in a hipcc -S listing (static view for VALU budgeting).
usage: isa_blocks.py file.s mangled_kernel_name [min_instrs]"""
import re, sys
path, name = sys.argv[1], sys.argv[2]
min_n = int(sys.argv[3]) if len(sys.argv) > 3 else 8
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(name + ":"))
blocks, cur = [], ["entry", {}]
tot = {}
for l in lines[start + 1:]:
    s = l.strip()
    if s.startswith(".Lfunc_end"):
        break
    m = re.match(r"^(\.LBB\d+_\d+):", s)
    if m:
        blocks.append(cur); cur = [m.group(1), {}]; continue
    if not s or s.startswith(";") or s.startswith("."):
        continue
    op = s.split()[0]
    kind = ("valu" if op.startswith("v_") else "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_"))
            else "branch" if op.startswith(("s_cbranch", "s_branch")) else "wait" if op.startswith(("s_waitcnt", "s_nop", "s_barrier")) else "salu")
    cur[1][kind] = cur[1].get(kind, 0) + 1
    tot[kind] = tot.get(kind, 0) + 1
blocks.append(cur)
print("total", tot, "blocks", len(blocks))
for b, c in blocks:
    n = sum(c.values())
    if n >= min_n:
        print(f"{b:12s} n={n:4d} " + " ".join(f"{k}={v}" for k, v in sorted(c.items())))
