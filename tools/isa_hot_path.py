"""Instruction count of the HOT path through the interior-step loop (the last depth-2 loop) of a fill kernel's ISA:
every conditional branch falls through, except the head's "not a ghost-block boundary" branch (taken) and guarded
blocks that hold rare work (error flag, strip change, partner wait): those are skipped.   python tools/isa_hot_path.py k.s"""
import collections, re, sys
lines = open(sys.argv[1]).read().split("\n")
lab = {m.group(1): i for i, l in enumerate(lines) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
hdr = [re.search(r"Header=(BB\d+_\d+) Depth=2", l) for l in lines]
last = [h.group(1) for h in hdr if h][-1]
i = lab[".L" + last]
RARE = ("global_atomic", "v_mul_lo_u32", "s_sleep", "flat_load", "flat_store", "s_setprio", "global_load_lds")
ops = collections.Counter()
seen_head_branch = False
steps = 0
while steps < 5000:
    steps += 1
    l = lines[i].strip()
    i += 1
    if not l or l.startswith(";") or l.startswith(".") :
        continue
    op = l.split()[0]
    ops[op] += 1
    if op == "s_branch":
        tgt = l.split()[1]
        if tgt == ".L" + last:
            break
        i = lab[tgt]
        continue
    if op.startswith("s_cbranch"):
        tgt = l.split()[1]
        if tgt not in lab:
            continue
        if op == "s_cbranch_scc1" and not seen_head_branch:  # g & (BLK-1) != 0: no block boundary work
            seen_head_branch = True
            i = lab[tgt]
            continue
        j = lab[tgt]
        if j > i:
            block = "\n".join(lines[i:j])
            if any(r in block for r in RARE):
                i = j
cls = lambda p: sum(c for o, c in ops.items() if o.startswith(p))
print(f"hot path: total {sum(ops.values())}  VALU {cls('v_')}  DS {cls('ds_')}  SALU {cls('s_')}  VMEM {cls('global_')}")
for o, c in ops.most_common(int(sys.argv[2]) if len(sys.argv) > 2 else 14):
    print(f"  {c:4d} {o}")
