#!/bin/bash
# 4-argument mode through the CLI: does the number of hardware queues matter (five streams: compute, upload, download, two side queues)?
TAG=${1:-r03x}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
export FAV_E2E_VARIANTS="s4:-structure 1;s4_q8:-structure 1 GPU_MAX_HW_QUEUES=8;s4_q6:-structure 1 GPU_MAX_HW_QUEUES=6;s3:-structure 0;s3_q8:-structure 0 GPU_MAX_HW_QUEUES=8;s4_b:-structure 1;s4_q8_b:-structure 1 GPU_MAX_HW_QUEUES=8"
timeout 600 python scripts/e2e.py 300 > $O/e2e_$TAG.log 2>&1
cat $O/e2e_$TAG.log | cut -c1-700
B="python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-e2e"
for q in 4 8; do
  GPU_MAX_HW_QUEUES=$q timeout 300 $B 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench GPU_MAX_HW_QUEUES=$q', j['value'], j['extra'])"
done
