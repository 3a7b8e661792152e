#!/bin/bash
# Round-3 GPU call G: PNG tests with the stored-row form, writer CPU split, HW queue count
TAG=${1:-r03g}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
(timeout 600 python -m pytest tests/test_gpu_png.py tests/test_gpu_cli.py -m gpu -q --timeout 300 2>&1 | tail -12) > $O/test_png_$TAG.log
export FAV_E2E_VARIANTS="default:-png_encoder gpu;hwq8:GPU_MAX_HW_QUEUES=8 -png_encoder gpu;hwq2:GPU_MAX_HW_QUEUES=2 -png_encoder gpu;w4:-png_encoder gpu -writers 4"
python scripts/e2e.py 600 > $O/e2e_$TAG.log 2>&1
cat $O/test_png_$TAG.log; cat $O/e2e_$TAG.log | cut -c1-1000
