"""Op histogram of the interior-step loop (the last depth-2 loop) of a fill kernel's ISA.  Counts
every instruction between the loop's first and last block, rare paths (block prefetch, partner
wait) included: a comparison figure between builds, not a cycle count."""
import collections, re, sys
lines = open(sys.argv[1]).read().split("\n")
lab = [(i, l) for i, l in enumerate(lines) if re.match(r"^\.LBB\d+_\d+:", l)]
hdrs = [re.search(r"Header=(BB\d+_\d+) Depth=2", l) for _, l in lab]
last = [h.group(1) for h in hdrs if h][-1]
idx = [k for k, (i, l) in enumerate(lab) if f"Header={last} Depth=2" in l or f"Header={last} Depth=3" in l]
start, end = lab[idx[0]][0], lab[idx[-1] + 1][0]
ops = collections.Counter()
for l in lines[start:end]:
    l = l.strip()
    if not l or l.startswith(";") or l.startswith("."):
        continue
    ops[l.split()[0]] += 1
cls = lambda p: sum(c for o, c in ops.items() if o.startswith(p))
print(f"lines {start}-{end}: total {sum(ops.values())}  VALU {cls('v_')}  DS {cls('ds_')}  SALU {cls('s_')}  VMEM {cls('global_')}")
for o, c in ops.most_common(int(sys.argv[2]) if len(sys.argv) > 2 else 12):
    print(f"  {c:4d} {o}")
