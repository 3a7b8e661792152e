#!/bin/bash
TAG=${1:-r03l}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
(timeout 900 python -m pytest tests/test_gpu_cli.py tests/test_gpu_png.py -m gpu -q --timeout 600 2>&1 | tail -12) > $O/test_cli_$TAG.log
(timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -k "two_networks or stream_vs_oracle or stream_options or lookahead or image_model" 2>&1 | tail -8) > $O/test_par_$TAG.log
export FAV_E2E_VARIANTS="default:-png_encoder gpu;structure1:-png_encoder gpu -structure 1;host:-png_encoder host"
python scripts/e2e.py 600 > $O/e2e_$TAG.log 2>&1
taskset -c 0-1 python scripts/e2e.py 600 > $O/e2e_taskset_$TAG.log 2>&1
cat $O/test_cli_$TAG.log $O/test_par_$TAG.log; cat $O/e2e_$TAG.log | cut -c1-1000; echo "--- taskset -c 0-1"; cat $O/e2e_taskset_$TAG.log | cut -c1-1000
