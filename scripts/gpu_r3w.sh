#!/bin/bash
# two-block Winograd form: bit identity with the one-block form, then A/B of the frame rate and the in-kernel timeline
TAG=${1:-r03w}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
(timeout 900 python -m pytest tests/test_gpu_parity.py -q -x --timeout 600 -k "two_block or winograd or canonical_network" 2>&1 | tail -8) > $O/test_$TAG.log
cat $O/test_$TAG.log
B="python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-extra --no-e2e"
run() { # name, env...
  local name=$1; shift
  (env "$@" timeout 300 $B 2> $O/ab_${TAG}_$name.err | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$name', 'fps', j['value'], 'frac', j['roofline']['frac'], 'wino_us', j['roofline'].get('avg_launch_us'))") 2>&1 | tail -2
  grep WINODBG $O/ab_${TAG}_$name.err | head -3
}
run one_block FAV_WINO_ONE_BLOCK=1 FAV_WINO_DBG=25
run two_default FAV_WINO_DBG=25
run two_nodelay FAV_WINO_PHASE_NS=0 FAV_WINO_DBG=25
run two_5us FAV_WINO_PHASE_NS=5000
run two_16us FAV_WINO_PHASE_NS=16000
run one_block_b FAV_WINO_ONE_BLOCK=1
run two_default_b X=1
