#!/bin/bash
TAG=${1:-r04g}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
B="python bench.py --structure 1 --steps 60 --warmup 10 --no-cpu-baseline --no-extra --no-e2e"
run() { local name=$1; shift
  (env "$@" timeout 300 $B 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$name', 'fps', j['value'], 'wino_us', j['roofline'].get('avg_launch_us'))") 2>&1 | tail -1 | cut -c1-400; }
run dyn_tail16 X=1
run static FAV_NO_DYNAMIC=1
run dyn_tail32 FAV_WINO_TAIL=32
run dyn_tail8 FAV_WINO_TAIL=8
run dyn_tail0 FAV_WINO_TAIL=0
run dyn_tail16_b X=1
run static_b FAV_NO_DYNAMIC=1
