#!/bin/bash
# tools/state_hunt.sh -- does THIS box show the slow state?  Six short headline processes; when one of them reads slow, the placement and
# stream probes are run in the same call (the state follows the process, profiles/r05_queue_probe.txt), and the A/B of the headline variants.
cd "$GRAFT_REPO_ROOT"
REPS="1 2 3" STEPS=10 bash tools/ab_state.sh tools/ab/head_base.so tools/ab/head_base.so 2>&1 | tee /tmp/hunt.txt
if grep -q "slow\|below" /tmp/hunt.txt; then
  echo "== slow state seen: probes"
  CYCLES=2 python tools/placement_probe2.py 5 2>&1 | grep cycle
  CYCLES=2 python tools/placement_probe2.py 5 2>&1 | grep cycle
  CYCLES=1 python tools/stream_probe.py 4 2>&1 | grep cycle
  REPS="1 2 3 4 5 6" STEPS=10 bash tools/ab_state.sh tools/ab/head_base.so tools/ab/head_karg.so
else
  echo "== this box is fast in every process"
fi
