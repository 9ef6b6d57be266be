"""Which scalar values does the specialised step kernel spill?  Reads the disassembly that tools/spec_resources.py keeps
(KEEP_ASM=path) and lists, per spilled value, the instruction that produced it and how often it is fetched back
(static counts of v_writelane / v_readlane on the spill registers).  Usage: spill_report.py asm [top]"""
import collections, re, sys
lines = open(sys.argv[1]).read().split("\n")
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
pat = re.compile(r"\s+(\S+)\s+(.*?)\s+//")
spill_regs = collections.Counter()
for l in lines:
    m = pat.match(l)
    if m and m.group(1) == "v_writelane_b32":
        spill_regs[m.group(2).split(",")[0].strip()] += 1
regs = {r for r, c in spill_regs.items()}
defs, last, cnt, where = {}, {}, collections.Counter(), collections.defaultdict(list)
for i, l in enumerate(lines):
    m = pat.match(l)
    if not m:
        continue
    op, a = m.group(1), [x.strip() for x in m.group(2).split(",")]
    if op == "v_writelane_b32":
        last[(a[0], a[2])] = (i + 1, defs.get(a[1], (None, ""))[1][:64])
    elif op == "v_readlane_b32" and a[1] in regs and a[2].isdigit():
        w = last.get((a[1], a[2]))
        cnt[w] += 1
        where[w].append(i + 1)
    elif a and a[0].startswith("s"):
        mm = re.match(r"s\[(\d+):(\d+)\]", a[0])
        names = [f"s{k}" for k in range(int(mm.group(1)), int(mm.group(2)) + 1)] if mm else [a[0]]
        for n in names:
            defs[n] = (i + 1, l.split("//")[0].strip())
print("spill registers:", dict(spill_regs), "| writes", sum(spill_regs.values()), "| reads", sum(cnt.values()))
for k, v in sorted(cnt.items(), key=lambda x: -x[1])[:top]:
    print(f"{v:4d} reads  written at line {k[0] if k else '?'}: {k[1] if k else ''}   first/last read {where[k][0]}/{where[k][-1]}")
