#!/bin/bash
TAG=${1:-r03h}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
export FAV_E2E_VARIANTS="default:-png_encoder gpu;two_cores:-png_encoder gpu -writers 2;host:-png_encoder host"
python scripts/e2e.py 600 > $O/e2e_$TAG.log 2>&1
taskset -c 0-1 python scripts/e2e.py 600 > $O/e2e_taskset_$TAG.log 2>&1
cat $O/e2e_$TAG.log | cut -c1-1000; echo "--- taskset -c 0-1"; cat $O/e2e_taskset_$TAG.log | cut -c1-1000
