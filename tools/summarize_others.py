"""tools/summarize_others.py TAG -- profiles/TAG_other_kernels.csv from gpurun_out/others_TAG (tools/profile_others.sh):
one row per workload with the dominant emgpu kernel's rocprofv3 --stats line."""
import csv, os, sys
import glob as _glob
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
src = os.path.join(ROOT, "gpurun_out", "others_" + tag)
rows = []
for d in sorted(glob.glob(os.path.join(src, "*", ""))):
    name = os.path.basename(os.path.dirname(d))
    for f in glob.glob(os.path.join(d, "*", "*_kernel_stats.csv")):
        for r in csv.DictReader(open(f)):
            if "emgpu::" in r["Name"]:
                rows.append([name, r["Name"][:110], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"], r["Percentage"]])
with open(os.path.join(ROOT, "profiles", tag + "_other_kernels.csv"), "w") as o:
    w = csv.writer(o)
    w.writerow(["workload", "kernel", "calls", "average_ns", "min_ns", "max_ns", "percent_of_gpu_time"])
    w.writerows(rows)
for r in rows:
    print(r[0], r[1][:60], "avg %.3f ms min %.3f ms" % (float(r[3]) / 1e6, float(r[4]) / 1e6))
