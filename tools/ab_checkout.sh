#!/bin/bash
# tools/ab_checkout.sh <commit> <name> -- export <commit> into tools/ab/<name>/ (git-ignored) and build
# its library there, for tools/ab_bench.sh.
set -e
cd "$(dirname "$0")/.."
rm -rf "tools/ab/$2"; mkdir -p "tools/ab/$2"
git archive "$1" -- bench.py em_model_manned_bayes_amd include models oracle | tar -x -C "tools/ab/$2"
make -C "tools/ab/$2/em_model_manned_bayes_amd/csrc" -j4 > /dev/null 2>&1
ls -la "tools/ab/$2/em_model_manned_bayes_amd/libemgpu.so"
