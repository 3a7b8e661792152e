#!/bin/bash
TAG=${1:-r03r}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
(timeout 600 python -m pytest tests/test_gpu_cli.py -m gpu -q --timeout 600 2>&1 | tail -4) > $O/test_cli_$TAG.log
export FAV_E2E_VARIANTS="a:-png_encoder gpu;b:-png_encoder gpu;c:-png_encoder gpu"
python scripts/e2e.py 1500 > $O/e2e_$TAG.log 2>&1
taskset -c 0-1 python scripts/e2e.py 1500 > $O/e2e_taskset_$TAG.log 2>&1
cat $O/test_cli_$TAG.log; cut -c1-620 $O/e2e_$TAG.log | grep -v thread; echo "--- taskset"; cut -c1-620 $O/e2e_taskset_$TAG.log | grep -v thread
