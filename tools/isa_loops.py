"""tools/isa_loops.py <file.s> <kernel-substring> -- per-loop instruction mix of one kernel in a hipcc -S listing
(basic blocks grouped by the loop-header comments LLVM emits), innermost hot loop first."""
import collections, re, sys
s = open(sys.argv[1]).read()
key = sys.argv[2]
m = re.search(r"^(_Z\w*%s\w*):" % re.escape(key), s, re.M)
k = s[m.start():]
k = k[:k.index(".Lfunc_end")]
print(m.group(1))
blocks = []  # (label, depth, header-of, counter)
cur = None
for l in k.split("\n"):
    t = l.strip()
    mm = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", t)
    if mm:
        cur = [mm.group(1), mm.group(2) or "", collections.Counter()]
        blocks.append(cur)
        continue
    if not t or t.startswith(";") or t.startswith(".") or cur is None:
        continue
    cur[2][t.split()[0]] += 1
tot = collections.Counter()
for lab, com, c in blocks:
    n = sum(c.values())
    if n >= 40:
        valu = sum(v for o, v in c.items() if o.startswith("v_"))
        print("%-12s n=%4d valu=%4d nop=%3d mad=%3d rdlane=%3d wrlane=%3d scratch=%2d ds=%3d gl=%2d | %s" % (
            lab, n, valu, c["s_nop"], c["v_mad_u64_u32"], c["v_readlane_b32"], c["v_writelane_b32"],
            sum(v for o, v in c.items() if "scratch" in o), sum(v for o, v in c.items() if o.startswith("ds_")),
            sum(v for o, v in c.items() if o.startswith("global_")), com[:70]))
