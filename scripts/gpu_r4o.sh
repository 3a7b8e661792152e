#!/bin/bash
# first layer with F(2x2,3x3) over its nine 3x3 blocks: parity tests, then A/B against the 1-D form
TAG=${1:-r04o}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
(timeout 900 python -m pytest tests/test_gpu_parity.py -q -x --timeout 600 -k "canonical or first or image_model or networks or odd" 2>&1 | tail -8) > $O/test_$TAG.log; cat $O/test_$TAG.log
B="python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-extra --no-e2e"
run() { local name=$1; shift
  (env "$@" timeout 300 $B 2> $O/ab_${TAG}_$name.err | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
pk = j['roofline']['per_kernel_ms_tflops']
print('$name', 'fps', j['value'], 'first layer [ms, TFLOP/s]', pk.get('16'), pk.get('6'), 'parity', j.get('parity'))") 2>&1 | tail -1 | cut -c1-400; }
run first2d X=1
run first1d FAV_FIRST_1D=1
run first2d_b X=1
run first1d_b FAV_FIRST_1D=1
