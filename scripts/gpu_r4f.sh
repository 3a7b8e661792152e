#!/bin/bash
# dynamic work distribution next to the side queues: tests, then the 4-argument mode in HBM and through the CLI, with and without
TAG=${1:-r04f}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
(timeout 900 python -m pytest tests/test_gpu_parity.py -q -x --timeout 600 -k "lookahead or structure or prefetch or four_arg or 4arg" 2>&1 | tail -5) > $O/test_$TAG.log; cat $O/test_$TAG.log
B="python bench.py --structure 1 --steps 60 --warmup 10 --no-cpu-baseline --no-extra --no-e2e"
run() { local name=$1; shift
  (env "$@" timeout 300 $B 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$name', 'fps', j['value'], 'wino_us', j['roofline'].get('avg_launch_us'), {k: v for k, v in j.get('per_kernel_ms', {}).items()} if 'per_kernel_ms' in j else '')") 2>&1 | tail -1 | cut -c1-400; }
run dynamic X=1
run static FAV_NO_DYNAMIC=1
run dynamic_b X=1
run static_b FAV_NO_DYNAMIC=1
export FAV_E2E_VARIANTS="s4_dyn:-structure 1;s4_static:-structure 1 FAV_NO_DYNAMIC=1;s3:-structure 0"
timeout 600 python scripts/e2e.py 1000 > $O/e2e_$TAG.log 2>&1
cut -c1-120 $O/e2e_$TAG.log | grep -v "thread CPU"
