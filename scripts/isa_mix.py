#!/usr/bin/env python3
"""Static instruction mix of a kernel's loops from the compiler's assembly (hipcc -S --cuda-device-only):
per basic block of every loop: VALU / SALU / branch / VMEM / SMEM / LDS / waitcnt counts, and the totals per loop.

    python3 scripts/isa_mix.py k_frame.s 'k_shadowILb0E'  [> profiles/r3_isa_mix.txt]

Blocks are the compiler's (.LBBx_y labels and the fall-through '; %bb.N' markers); the loop a block belongs to is
taken from the 'in Loop: Header=BBx_y Depth=d' comments.  A static count is not the dynamic mix (blocks under an
exec mask are skipped when no lane needs them), but it is what one wave issues for a visit whose lanes take every path.
"""
import collections
import re
import sys

path, pat = sys.argv[1], sys.argv[2]
lines = open(path).read().splitlines()
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(pat) + r"\w*:", l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))


def classify(op):
    if op.startswith("s_cbranch") or op in ("s_branch", "s_setpc_b64", "s_endpgm"):
        return "branch"
    if op.startswith("s_waitcnt") or op == "s_nop":
        return "wait/nop"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "SMEM"
    if op.startswith("s_"):
        return "SALU"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "VMEM"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith("v_"):
        return "VALU"
    return "other"


blocks = []   # (label, loop header, depth, Counter)
cur = ["entry", None, 0, collections.Counter()]
for l in lines[start + 1:end]:
    s = l.strip()
    m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", l) or re.match(r"^; (%bb\.\d+):\s*(;.*)?$", s)
    if m:
        blocks.append(tuple(cur))
        hdr = re.search(r"Header=(BB\d+_\d+) Depth=(\d+)", l)
        own = re.search(r"Loop Header: Depth=(\d+)", l)
        label = m.group(1)
        if own:
            cur = [label, label.lstrip("."), int(own.group(1)), collections.Counter()]
        elif hdr:
            cur = [label, "L" + hdr.group(1), int(hdr.group(2)), collections.Counter()]
        else:
            cur = [label, None, 0, collections.Counter()]
        continue
    if not s or s.startswith((";", ".")) or s.endswith(":"):
        continue
    op = s.split()[0]
    cur[3][classify(op)] += 1
blocks.append(tuple(cur))

cols = ["VALU", "SALU", "branch", "VMEM", "SMEM", "LDS", "wait/nop"]
print("kernel %s (%s): %d basic blocks" % (pat, path.split("/")[-1], len(blocks)))
print("%-12s %-12s %5s | " % ("block", "loop", "depth") + " ".join("%8s" % c for c in cols))
tot = collections.defaultdict(collections.Counter)
for label, hdr, depth, c in blocks:
    if not sum(c.values()):
        continue
    print("%-12s %-12s %5d | " % (label, hdr or "-", depth) + " ".join("%8d" % c[k] for k in cols))
    if hdr:
        tot[(hdr.lstrip("L").lstrip("."), depth)].update(c)
print()
print("totals per loop (all blocks of the loop, every path taken once):")
for (hdr, depth), c in sorted(tot.items()):
    print("loop %-10s depth %d | " % (hdr, depth) + " ".join("%8d" % c[k] for k in cols) + "   total %d" % sum(c.values()))
