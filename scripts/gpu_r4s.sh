#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_cli.py tests/test_gpu_parity.py -q -x --timeout 600 -k "cli or stylize or canonical or stream or odd or shared or two_networks" 2>&1 | tail -4)
export FAV_E2E_VARIANTS="s3:-structure 0 FAV_LOOP_TRACE=2;s3_b:-structure 0;s3_nowarm:-structure 0 FAV_NO_WARMUP=1;s4:-structure 1;s3_c:-structure 0;s3_nowarm_b:-structure 0 FAV_NO_WARMUP=1"
timeout 600 python scripts/e2e.py 300 2>&1 | grep -E "loop trace frame [1-3]:| wall" | cut -c1-210
