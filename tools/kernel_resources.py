"""tools/kernel_resources.py <file.hip> [substring] -- registers, spills, scratch, LDS and occupancy of every kernel in a translation
unit (hipcc -Rpass-analysis=kernel-resource-usage, gfx950), one line per kernel."""
import re, subprocess, sys, os
src = sys.argv[1]
key = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950", "-c", src, "-o", "/dev/null",
                      "-Rpass-analysis=kernel-resource-usage"] + sys.argv[3:], capture_output=True, text=True).stderr
cur = None
rows = []
for l in out.splitlines():
    m = re.search(r"remark: (?:\S+ )?\s*Function Name: (\S+)", l)
    if m:
        cur = {"name": m.group(1)}; rows.append(cur); continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", l)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
for r in rows:
    if key in r["name"]:
        name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(.*", "", name)
        print("%-70s VGPR %3d  AGPR %3d  spillV %3d  spillS %3d  scratch %4d  LDS %6d  waves/SIMD %d" % (
            name[-70:], r.get("VGPRs", -1), r.get("AGPRs", -1), r.get("VGPRs Spill", -1), r.get("SGPRs Spill", -1), r.get("ScratchSize", -1),
            r.get("LDS Size", -1), r.get("Occupancy", -1)))
