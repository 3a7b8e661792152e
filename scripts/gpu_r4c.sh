#!/bin/bash
TAG=${1:-r04c}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
export FAV_E2E_VARIANTS="s4:-structure 1 FAV_LOOP_TRACE=1;s3:-structure 0 FAV_LOOP_TRACE=1"
timeout 600 python scripts/e2e.py 300 > $O/e2e_$TAG.log 2>&1
cut -c1-200 $O/e2e_$TAG.log | grep -v "thread CPU"
