#!/bin/bash
TAG=${1:-r03p}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_png.py -m gpu -q --timeout 600 -k "lookahead or png or stream_options or two_networks" 2>&1 | tail -8) > $O/test_par_$TAG.log
(timeout 900 python -m pytest tests/test_gpu_cli.py -m gpu -q --timeout 600 2>&1 | tail -8) > $O/test_cli_$TAG.log
export FAV_E2E_VARIANTS="structure0:-png_encoder gpu -structure 0;structure1:-png_encoder gpu -structure 1;structure1_again:-png_encoder gpu -structure 1"
python scripts/e2e.py 600 > $O/e2e_$TAG.log 2>&1
taskset -c 0-1 python scripts/e2e.py 600 > $O/e2e_taskset_$TAG.log 2>&1
cat $O/test_par_$TAG.log $O/test_cli_$TAG.log; cut -c1-760 $O/e2e_$TAG.log; echo "--- taskset"; cut -c1-760 $O/e2e_taskset_$TAG.log
