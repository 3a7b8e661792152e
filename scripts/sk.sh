#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for m in 0 1 2; do
  echo "== FAV_SK=$m"; FAV_SK=$m timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 300 -k "conv or network or stream_vs or 640" 2>&1 | tail -2
  FAV_SK=$m timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['avg_launch_us'], d['roofline']['achieved'], d['ms_per_step'], d['roofline']['conv_stack_ms_per_frame'])"
done
