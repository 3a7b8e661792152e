#!/bin/bash
# iir passes as blocks of four waves instead of one: mask parity tests, then the 4-argument mode in HBM and through the CLI
TAG=${1:-r04h}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
(timeout 900 python -m pytest tests/test_gpu_parity.py -q -x --timeout 600 -k "structure or consistency or lookahead or sequential" 2>&1 | tail -5) > $O/test_$TAG.log; cat $O/test_$TAG.log
B="python bench.py --structure 1 --steps 60 --warmup 10 --no-cpu-baseline --no-extra --no-e2e"
run() { local name=$1; shift
  (env "$@" timeout 300 $B 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$name', 'fps', j['value'], 'wino_us', j['roofline'].get('avg_launch_us'))") 2>&1 | tail -1 | cut -c1-400; }
run iir256 X=1
run iir64 FAV_IIR_BLOCK=64
run iir256_cus6 FAV_SIDE_CUS=6
run iir256_cus4 FAV_SIDE_CUS=4
run iir256_b X=1
run iir64_b FAV_IIR_BLOCK=64
export FAV_E2E_VARIANTS="s4_iir256:-structure 1;s4_iir64:-structure 1 FAV_IIR_BLOCK=64;s4_iir256_b:-structure 1"
timeout 600 python scripts/e2e.py 1000 > $O/e2e_$TAG.log 2>&1
cut -c1-120 $O/e2e_$TAG.log | grep -v "thread CPU"
