#!/bin/bash
# tools/scale_curve.sh [bench args] -- the 1 / 2 / 4 / 8-GPU lines of bench.py back to back on ONE node, in the shape of the driver's SCALE_rNN.json
# (VERDICT r5 next #7c: no 8-GPU node has been available to this build; the day one is, this is the one command).  Each N is launched the way the
# driver launches it (python -m torch.distributed.run, one rank per GPU, RCCL), its line goes to gpurun_out/scale/n<N>.json, its DETAIL record (every
# candidate trace's time per rank 0, telemetry) to n<N>.detail.json, and every rank's global index ranges to ranges_n<N>/ (--ranges-out: the
# ranges of a step must be disjoint and gap-free across the ranks -- checked below).  Scaling efficiency is NOT computed here: the driver does that.
#   STEPS (default 20), WARMUP (5), GPUS ("1 2 4 8"), PORT (29511)
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/scale
mkdir -p $OUT
have=$(python - <<'PY'
import sys
sys.path.insert(0, ".")
import bench
print(bench.visible_gpu_count())
PY
)
echo "visible GPUs: $have"
for N in ${GPUS:-1 2 4 8}; do
  if [ "$N" -gt "$have" ]; then echo "n=$N: skipped (only $have GPU(s) visible)"; continue; fi
  rm -rf $OUT/ranges_n$N
  if [ "$N" -eq 1 ]; then
    python bench.py --gpus 1 --steps ${STEPS:-20} --warmup ${WARMUP:-5} --no-other-configs --no-host-path --ranges-out $OUT/ranges_n$N --detail-out $OUT/n$N.detail.json "$@" > $OUT/n$N.json 2> $OUT/n$N.err
  else
    HSA_ENABLE_IPC_MODE_LEGACY=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port ${PORT:-29511} \
      bench.py --gpus $N --steps ${STEPS:-20} --warmup ${WARMUP:-5} --ranges-out $OUT/ranges_n$N --detail-out $OUT/n$N.detail.json "$@" > $OUT/n$N.json 2> $OUT/n$N.err
  fi
  echo "n=$N rc=$? $(tail -1 $OUT/n$N.json | cut -c1-240)"
done
python - "$OUT" <<'PY'
import glob, json, os, sys
out = sys.argv[1]
rows = []
for f in sorted(glob.glob(os.path.join(out, "n*.json"))):
    if f.endswith(".detail.json"):
        continue
    try:
        l = json.loads(open(f).read().strip().split("\n")[-1])
    except Exception as e:
        print(f, "no line:", e); continue
    n = l["n_gpus"]
    r = l["roofline"]
    rows.append({"n_gpus": n, "value": l["value"], "unit": l["unit"], "ms_per_step": l["ms_per_step"], "per_gpu": l["value"] / n,
                 "rank0_avg_launch_ms": r.get("avg_launch_ms"), "rank0_placement": r.get("placement"), "rank0_first_allocation_ms": r.get("first_allocation_ms"),
                 "box_state": l["config"].get("box_state"), "sclk_mhz": r.get("sclk_mhz"), "socket_power_w": r.get("socket_power_w")})
    # the ranks' ranges of every step: disjoint, gap-free, n x n_gpus units
    rs = [json.load(open(g)) for g in sorted(glob.glob(os.path.join(out, "ranges_n%d" % n, "rank*.json")))]
    if rs:
        steps = sorted({x["step"] for r_ in rs for x in r_["ranges"]})
        for k in steps:
            iv = sorted((x["first"], x["first"] + x["n"]) for r_ in rs for x in r_["ranges"] if x["step"] == k)
            assert all(a[1] == b[0] for a, b in zip(iv, iv[1:])), ("ranges of step", k, iv)
        rows[-1]["ranges_checked_steps"] = len(steps)
json.dump({"lines": rows}, open(os.path.join(out, "scale_curve.json"), "w"), indent=1)
for r in rows:
    print("n=%d  %.4g %s  %.3f ms/step  per GPU %.4g  rank 0 launch %s ms  placement %s" % (r["n_gpus"], r["value"], r["unit"], r["ms_per_step"], r["per_gpu"],
                                                                                       r["rank0_avg_launch_ms"], (r["rank0_placement"] or {}).get("ms")))
PY
