#!/bin/bash
TAG=${1:-r04j}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra --no-e2e"
for v in "X=1" "FAV_NO_LAZY_JOIN=1" "X=2" "FAV_NO_LAZY_JOIN=1"; do
  echo "== $v"; env $v FAV_WINO_DBG=21 timeout 300 $B 2>$O/dbg.err | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fps', j['value'], 'wino_us', j['roofline'].get('avg_launch_us'))"; grep WINODBG $O/dbg.err
done
