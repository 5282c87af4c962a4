"""tools/isa_stats.py -- static instruction mix of the main loop of k_uncor_fast<7,5,7,7> (device asm)."""
import collections, re, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "em_model_manned_bayes_amd", "csrc", "emgpu_kernels_fast.hip")
extra = sys.argv[1:]
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-ffp-contract=off", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-o", "/tmp/fast.s", src] + extra, stderr=subprocess.DEVNULL)
s = open("/tmp/fast.s").read()
k = s[s.index("_ZN5emgpu12k_uncor_fastILi7ELi2ELi4ELi2EEEv9EmgpuPlan8EmgpuRunNS_8FastArgsE:"):]
k = k[:k.index(".Lfunc_end")]
# find the main loop header: the loop with most mads
loops = collections.defaultdict(collections.Counter)
cur = None
for l in k.split("\n"):
    t = l.strip()
    m = re.match(r"^(\.LBB\d+_\d+):\s*;\s*(.*)$", t)
    if m:
        c = m.group(2)
        h = re.search(r"Header: Depth=1", c)
        if "This Loop Header: Depth=1" in c:
            cur = m.group(1)[1:]
        elif "Header=" in c or "Parent Loop" in c or "Inner Loop" in c:
            mm = re.search(r"(?:Header=|Parent Loop )(BB\d+_\d+)", c)
            cur = mm.group(1) if mm and "Depth=1" in c or mm else cur
        else:
            cur = None
        continue
    if re.match(r"^\.LBB\d+_\d+:", t):
        cur = None
        continue
    if not t or t.startswith(";") or t.startswith("."):
        continue
    if cur:
        loops[cur][t.split()[0]] += 1
best = max(loops.items(), key=lambda kv: kv[1]["global_store_dwordx4"])
c = best[1]
valu = sum(v for o, v in c.items() if o.startswith("v_"))
salu = sum(v for o, v in c.items() if o.startswith("s_"))
print("loop", best[0], "total", sum(c.values()), "VALU", valu, "SALU", salu, "mad", c["v_mad_u64_u32"], "nop", c["s_nop"],
      "branches", sum(v for o, v in c.items() if "branch" in o), "ds", sum(v for o, v in c.items() if o.startswith("ds_")),
      "global", sum(v for o, v in c.items() if o.startswith("global_")))
print(c.most_common(28))
for line in s.split("\n"):
    if re.search(r"\.(vgpr_count|sgpr_count|group_segment_fixed_size|private_segment_fixed_size|sgpr_spill_count|vgpr_spill_count):", line):
        print(line.strip(), end="; ")
    if "k_uncor_fastILi7ELi2ELi4ELi4" in line and ".name" in line:
        break
print()
