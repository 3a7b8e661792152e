#!/bin/bash
# pending residual joins in the Winograd staging: LDS-DMA layout check, bit identity with the launched joins, frame rate A/B
TAG=${1:-r04i}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 60 scripts/ldstest.bin
(timeout 900 python -m pytest tests/test_gpu_parity.py -q -x --timeout 600 -k "pending_residual or winograd or canonical_network or canonical_1280" 2>&1 | tail -8) > $O/test_$TAG.log; cat $O/test_$TAG.log
B="python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-extra --no-e2e"
run() { local name=$1; shift
  (env "$@" timeout 300 $B 2> $O/ab_${TAG}_$name.err | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$name', 'fps', j['value'], 'frac', j['roofline']['frac'], 'wino_us', j['roofline'].get('avg_launch_us'), 'parity', j.get('parity', {}).get('max_abs'))") 2>&1 | tail -1 | cut -c1-300; }
run lazy X=1
run eager FAV_NO_LAZY_JOIN=1
run lazy_b X=1
run eager_b FAV_NO_LAZY_JOIN=1
