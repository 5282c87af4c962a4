#!/bin/bash
# tools/check_philox10.sh -- the Random123-default generator kept alive: builds the library (-DEMGPU_PHILOX_ROUNDS=10) and the oracle
# (-DEM_PHILOX_ROUNDS=10) side by side with the shipped 7-round ones (tools/ab/philox10.so, tools/ab/libem_oracle_philox10.so) and runs the
# live GPU-vs-oracle parity tests on them (the committed golden vectors are the 7-round build's and are not part of this run).
# Build here (no GPU needed), run the second half through gpurun:  bash tools/check_philox10.sh build;  gpurun -- 'bash tools/check_philox10.sh run'
set -e
cd "$(dirname "$0")/.."
if [ "$1" != "run" ]; then
  bash tools/build_variant.sh philox10 -DEMGPU_PHILOX_ROUNDS=10
  gcc -O2 -fPIC -std=c11 -ffp-contract=off -fno-fast-math -fopenmp -DEM_PHILOX_ROUNDS=10 -shared -o tools/ab/libem_oracle_philox10.so oracle/em_oracle.c -lm
fi
if [ "$1" != "build" ]; then
  export EMGPU_LIB=$PWD/tools/ab/philox10.so EM_ORACLE_LIB=$PWD/tools/ab/libem_oracle_philox10.so
  python -c "
from em_model_manned_bayes_amd import _lib as L
import sys; sys.path.insert(0, 'oracle'); import oracle as O
print(L.lib().emgpu_version().decode(), '| oracle rounds', O.philox_rounds())"
  python -m pytest tests/test_gpu_parity.py -q -k "uncor_sample_matches_oracle or event_lists_from_the_fast_kernel or per_step_mode or terminal_propagation_matches or plain_dbn_sample or random_models or dense_only_step or bn_sample_function" 2>&1 | tail -4
fi
