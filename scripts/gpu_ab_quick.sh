#!/bin/bash
# A/B of env switches in the timed configuration on one box: each item of $AB (and DEFAULT) twice, alternating
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
for rep in 1 2; do for kv in DEFAULT=1 $AB; do
  (env $kv timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra --no-e2e 2>&1 | tail -1 | python -c "
import sys, json
try:
    j = json.loads(sys.stdin.read()); r = j['roofline']
    print('$kv', j['value'], 'fps', r['per_kernel_ms_tflops'])
except Exception as e: print('$kv', 'FAILED', e)") | tee -a $O/abq_${1:-x}.log
done; done
