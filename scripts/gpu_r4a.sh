#!/bin/bash
# CLI with two frames of look-ahead in the 4-argument mode: tests, then file -> PNG rates (300 frames and sustained)
TAG=${1:-r04a}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
(timeout 900 python -m pytest tests/test_gpu_cli.py -q -x --timeout 600 2>&1 | tail -6) > $O/test_$TAG.log; cat $O/test_$TAG.log
export FAV_E2E_VARIANTS="s4:-structure 1;s3:-structure 0;s4_b:-structure 1;s4_host:-structure 1 -png_encoder host -png_level 1"
timeout 600 python scripts/e2e.py 300 > $O/e2e_$TAG.log 2>&1
cut -c1-330 $O/e2e_$TAG.log | grep -v "thread CPU"
export FAV_E2E_VARIANTS="s4_long:-structure 1;s3_long:-structure 0;s4_two_cpus:-structure 1"
timeout 600 python scripts/e2e.py 1500 > $O/e2e_${TAG}_long.log 2>&1
cut -c1-330 $O/e2e_${TAG}_long.log | grep -v "thread CPU"
