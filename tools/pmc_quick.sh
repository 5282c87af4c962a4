#!/bin/bash
# tools/pmc_quick.sh "<bench args>" COUNTERS... : one rocprofv3 --pmc pass over bench.py (bounded by `timeout`: an unsupported counter
# can hang the profiler), prints per-launch averages of the sampling kernels
ARGS="$1"; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmcq
timeout ${PMC_TIMEOUT:-150} rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmcq -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --no-host-path --prewarm-s 0 --placement-candidates 1 $ARGS > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmcq/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_uncor_fast" in r["Kernel_Name"] or "k_dbn" in r["Kernel_Name"] or "k_terminal_propagate" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("$ARGS", {k: "%.5g" % (sum(v)/len(v)) for k, v in sorted(agg.items())})
PY
