#!/bin/bash
# tools/profile_round.sh TAG -- every committed profile of a round in one gpurun call: the headline and BASELINE.json configs[2..4] through
# tools/profile_bench.sh (kernel trace + the HBM and SQ counter passes + the plain line), then the other workloads' kernel stats
# (tools/profile_others.sh).  Condense with: for t in TAG TAG_per_step TAG_cor TAG_cor_v2p1_like TAG_mixed TAG_terminal; do python tools/summarize_profiles.py $t; done;
# python tools/summarize_others.py TAG
TAG=${1:-r04}
cd "$GRAFT_REPO_ROOT"
bash tools/profile_bench.sh $TAG
bash tools/profile_bench.sh ${TAG}_per_step --config uncor_per_step
bash tools/profile_bench.sh ${TAG}_cor --config cor
bash tools/profile_bench.sh ${TAG}_cor_v2p1_like --config cor_v2p1_like
bash tools/profile_bench.sh ${TAG}_mixed --config mixed
bash tools/profile_bench.sh ${TAG}_terminal --config terminal
bash tools/profile_others.sh $TAG
