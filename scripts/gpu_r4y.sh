#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
(timeout 900 python -m pytest tests/test_gpu_parity.py -q -x --timeout 600 -k "canonical or min_filter or assemble or warp or stream_options or image_model or odd" 2>&1 | tail -3)
B="python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-extra --no-e2e"
for i in 1 2; do FAV_FOLD_DBG=15 timeout 300 $B 2>gpurun_out/y.err | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fps', j['value'], 'last layer', j['roofline']['per_kernel_ms_tflops'].get('1'))"; grep FOLDDBG gpurun_out/y.err; done
