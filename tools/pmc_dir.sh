#!/bin/bash
# tools/pmc_dir.sh DIR KERNEL "COUNTERS" -- one PMC pass of DIR's own bench.py + library (a checkout made by tools/ab_checkout.sh, or .)
D="$1"; K="$2"; C="$3"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT/$D"
rm -rf /tmp/pmcd
rocprofv3 --pmc $C --output-format csv -d /tmp/pmcd -- python3 bench.py --no-cpu-baseline --no-other-configs --steps 2 --warmup 1 > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("/tmp/pmcd/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "$K" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("$D", "$K", {k: "%.5g" % (sum(v)/len(v)) for k, v in sorted(agg.items())})
PY
