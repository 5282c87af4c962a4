"""tools/summarize_profiles.py TAG -- condense gpurun_out/prof_TAG (rocprofv3 CSVs) into
profiles/TAG_*: the kernel-stats table, per-launch PMC values of the dominant kernel and a JSON
summary that bench.py reads for roofline.traffic."""
import collections
import csv
import glob as _glob
import json
import os
import sys

class glob:  # gpurun merges a call's files into what earlier calls left behind: only the newest file of a kind in a directory counts
    @staticmethod
    def glob(pattern):
        best = {}
        for f in _glob.glob(pattern):
            key = (os.path.dirname(f), os.path.basename(f).split("_", 1)[-1])   # <pid>_kernel_stats.csv -> kernel_stats.csv
            if key not in best or os.path.getmtime(f) > os.path.getmtime(best[key]):
                best[key] = f
        return sorted(best.values())

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)

summary = {"tag": tag}
# kernel stats (rocprofv3 --kernel-trace --stats)
for f in glob.glob(os.path.join(src, "kt", "*", "*_kernel_stats.csv")):
    rows = list(csv.DictReader(open(f)))
    with open(os.path.join(dst, tag + "_kernel_stats.csv"), "w") as o:
        w = csv.writer(o)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in rows:
            w.writerow([r["Name"][:120], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]])
    ours = [r for r in rows if "emgpu::" in r["Name"]]   # (bench.py's write-ceiling calibration runs torch fill kernels after the timed region)
    top = ours[0] if ours else rows[0]
    summary["kernel"] = top["Name"]
    summary["kernel_calls"] = int(top["Calls"])
    summary["kernel_avg_ms"] = float(top["AverageNs"]) / 1e6
    summary["kernel_pct_of_gpu_time"] = float(top["Percentage"])
for f in glob.glob(os.path.join(src, "bench_plain.json")):
    try:
        summary["bench_line"] = json.loads(open(f).read().strip().split("\n")[-1])
    except Exception:
        pass
timed_steps = int(summary.get("bench_line", {}).get("steps", 10) or 10)   # the pass's own --steps (the line says how many it timed)
# every launch of the dominant kernel, in time order: the first five (clocks still ramping) are bench.py's warm-up,
# the rest are its timed region -- their mean is what bench.py's roofline.avg_launch_ms must agree with
for f in glob.glob(os.path.join(src, "kt", "*", "*_kernel_trace.csv")):
    rows = [r for r in csv.DictReader(open(f)) if summary.get("kernel") and r["Kernel_Name"] == summary["kernel"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ms = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
    # round 5: bench.py launches more before its timed region (pre-warm, placement candidates, warm-up, settle steps); the timed region is
    # the LAST `--steps` launches of the pass (default 10: the kernel-trace pass runs bench.py's defaults with --telemetry-s 0, nothing follows)
    summary["kernel_launches_in_the_pass"] = len(ms)
    summary["kernel_launch_ms"] = [round(x, 4) for x in ms[-40:]]
    if len(ms) > timed_steps + 5:
        summary["kernel_avg_ms_timed_region"] = sum(ms[-timed_steps:]) / float(timed_steps)
        summary["timed_steps"] = timed_steps
    elif len(ms) > 5:
        summary["kernel_avg_ms_timed_region"] = sum(ms[5:]) / len(ms[5:])   # (rounds 1-4: 5 warm-up launches, then the timed ones)
# per-dispatch resource usage from the kernel trace
for f in glob.glob(os.path.join(src, "kt", "*", "*_kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"] == summary.get("kernel"):
            summary["dispatch"] = {k: r[k] for k in ("Workgroup_Size", "Grid_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count") if k in r}
            break
# PMC passes
pmc = collections.defaultdict(list)
for d in ("pmc_write", "pmc_fetch", "pmc_sq1", "pmc_sq2"):
    for f in glob.glob(os.path.join(src, d, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"] == summary.get("kernel"):
                pmc[r["Counter_Name"]].append(float(r["Counter_Value"]))
summary["pmc_per_launch"] = {k: sum(v) / len(v) for k, v in pmc.items()}
pl = summary["pmc_per_launch"]
if "SQ_THREAD_CYCLES_VALU" in pl and pl.get("SQ_ACTIVE_INST_VALU"):
    # lanes active per cycle in which the vector unit executes (thread-cycles / (64 x busy cycles)): 1.0 = no divergence, no idle lanes
    summary["valu_lanes_active"] = pl["SQ_THREAD_CYCLES_VALU"] / (64.0 * pl["SQ_ACTIVE_INST_VALU"])
if "WRITE_SIZE" in pmc:
    # MI355X_MICROARCH.md "HBM": WRITE_SIZE (KiB) is exact for 16-B-per-lane streaming stores;
    # FETCH_SIZE (KiB) reports 1/2 of the bytes of wide coalesced reads on gfx950 -> doubled.
    wr = summary["pmc_per_launch"]["WRITE_SIZE"] * 1024
    rd = summary["pmc_per_launch"].get("FETCH_SIZE", 0.0) * 1024 * 2
    summary["hbm_write_bytes_per_launch"] = wr
    summary["hbm_read_bytes_per_launch_corrected"] = rd
    summary["hbm_traffic_bytes_per_launch"] = wr + rd
json.dump(summary, open(os.path.join(dst, tag + "_summary.json"), "w"), indent=1, sort_keys=True)
print(json.dumps({k: v for k, v in summary.items() if k != "bench_line"}, indent=1)[:3000])
