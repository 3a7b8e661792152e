#!/bin/bash
TAG=${1:-r03i}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
for v in 0 8 9 10 5; do timeout 60 scripts/runtime_thread_bench.bin $v; done 2>&1 | tee $O/runtime_thread_$TAG.log
