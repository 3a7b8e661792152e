#!/bin/bash
# Round-3 GPU call F: where does the host CPU go?  e2e legs under runtime wait-policy settings
TAG=${1:-r03f}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
export FAV_E2E_VARIANTS="default:-png_encoder gpu;active0:ROC_ACTIVE_WAIT_TIMEOUT=0 -png_encoder gpu;active50:ROC_ACTIVE_WAIT_TIMEOUT=50 -png_encoder gpu;mwaitx:HSA_ENABLE_MWAITX=1 -png_encoder gpu;nointr:HSA_ENABLE_INTERRUPT=0 -png_encoder gpu;nodirect:AMD_DIRECT_DISPATCH=0 -png_encoder gpu;cpuwait0:ROC_CPU_WAIT_FOR_SIGNAL=0 -png_encoder gpu"
python scripts/e2e.py 600 > $O/e2e_$TAG.log 2>&1
(timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -k "bf16" 2>&1 | tail -5) > $O/test_par_$TAG.log
cat $O/e2e_$TAG.log | cut -c1-900; cat $O/test_par_$TAG.log
