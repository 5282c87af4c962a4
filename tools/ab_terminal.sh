#!/bin/bash
# tools/ab_terminal.sh A.so B.so ... -- interleaved timing of k_terminal_propagate builds on ONE box (N encounters, default 2 M), then
# (PMC=1) the HBM counters of each build's propagation kernel in their own passes
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
N=${N:-2000000}
for rep in 1 2 3; do
  for v in "$@"; do
    ms=$(EMGPU_LIB=$PWD/$v timeout 120 python bench.py --config terminal --n $N --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --no-host-path --verbose-line 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read()); print(l['roofline']['avg_step_ms'])")
    echo "rep $rep $v $ms"
  done
done
if [ -n "${PMC:-}" ]; then
  for v in "$@"; do
    for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
      rm -rf gpurun_out/pmcab
      EMGPU_LIB=$PWD/$v timeout 200 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmcab -- python3 bench.py --config terminal --n $N --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --no-host-path --verbose-line > /dev/null 2>&1
      python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmcab/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_terminal_propagate" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("$v", {k: "%.4g" % (sum(x)/len(x)) for k, x in sorted(agg.items())})
PY
    done
  done
fi
