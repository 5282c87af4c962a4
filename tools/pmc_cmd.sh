#!/bin/bash
# tools/pmc_cmd.sh KERNEL_SUBSTRING "COUNTERS..." program args... -- one rocprofv3 --pmc pass over any python tool,
# prints the per-launch averages of the counters for kernels whose name contains KERNEL_SUBSTRING
K="$1"; shift
C="$1"; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmcc
timeout 150 rocprofv3 --pmc $C --output-format csv -d gpurun_out/pmcc -- python3 "$@" > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmcc/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "$K" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("$K", {k: "%.5g" % (sum(v)/len(v)) for k, v in sorted(agg.items())})
PY
