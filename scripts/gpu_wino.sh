#!/bin/bash
# Winograd bring-up on the GPU box: targeted parity tests, timeline of one launch, A/B against the direct kernel
TAG=${1:-w}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
(timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -x -k "${KEXPR:-winograd or canonical or stale or tiny_network}" 2>&1 | tail -15) > $O/wtest_$TAG.log
cat $O/wtest_$TAG.log
for kv in ${ABS:-FAV_NO_WINO=1 FAV_WINO_DBG=7 FAV_NO_WINO=1 FAV_WINO_DBG=18}; do
  (env $kv timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra --no-e2e 2>$O/werr_$TAG.log | tail -1 | python -c "
import sys, json
try:
    j = json.loads(sys.stdin.read()); r = j['roofline']
    print('$kv', j['value'], 'fps', 'dom_us', r['avg_launch_us'], 'conv_ms', r['conv_stack_ms_per_frame'], r['per_kernel_ms_tflops'])
except Exception as e: print('$kv', 'FAILED', e)"; grep WINODBG $O/werr_$TAG.log) 2>&1 | tee -a $O/wab_$TAG.log
done
if [ -n "$PARITY" ]; then timeout 600 python bench.py --steps 20 --warmup 3 --no-extra --no-e2e 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('parity', j.get('parity'), 'value', j['value'])" | tee -a $O/wab_$TAG.log; fi
