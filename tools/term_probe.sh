#!/bin/bash
# tools/term_probe.sh [bench args] -- kernel trace of the terminal config + the lane-activity counters of k_terminal_propagate
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/term_probe
rm -rf $OUT; mkdir -p $OUT
B="--config terminal --no-cpu-baseline --steps 5 --warmup 2 $*"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 bench.py $B > $OUT/kt.log 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/kt/*/*_kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:6]:
        print("%-90s calls %4s avg %10.3f ms  %5s%%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e6, r["Percentage"]))
PY
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc1 -- python3 bench.py $B --steps 2 --warmup 1 > $OUT/pmc1.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmcw -- python3 bench.py $B --steps 2 --warmup 1 > $OUT/pmcw.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmcf -- python3 bench.py $B --steps 2 --warmup 1 > $OUT/pmcf.log 2>&1
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$OUT/pmc*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_terminal_propagate" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
a = {k: sum(v) / len(v) for k, v in agg.items()}
print({k: "%.5g" % v for k, v in sorted(a.items())})
if "SQ_THREAD_CYCLES_VALU" in a and "SQ_ACTIVE_INST_VALU" in a:
    print("lanes active per VALU cycle: %.3f" % (a["SQ_THREAD_CYCLES_VALU"] / (64.0 * a["SQ_ACTIVE_INST_VALU"])))
PY
tail -2 $OUT/kt.log | cut -c1-300
