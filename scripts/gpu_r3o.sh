#!/bin/bash
TAG=${1:-r03o}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
(timeout 900 python -m pytest tests/test_gpu_cli.py -m gpu -q --timeout 600 2>&1 | tail -8) > $O/test_cli_$TAG.log
export FAV_E2E_VARIANTS="default:-png_encoder gpu;again:-png_encoder gpu;structure1:-png_encoder gpu -structure 1;host:-png_encoder host"
python scripts/e2e.py 300 > $O/e2e_$TAG.log 2>&1
taskset -c 0-1 python scripts/e2e.py 300 > $O/e2e_taskset_$TAG.log 2>&1
cat $O/test_cli_$TAG.log; cut -c1-700 $O/e2e_$TAG.log; echo "--- taskset"; cut -c1-700 $O/e2e_taskset_$TAG.log
