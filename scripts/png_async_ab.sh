#!/bin/bash
# A/B: the PNG encode in front of the next frame on the compute queue (default) against the stream's encoder queue (--png-async),
# in-HBM rate, both checker modes; then the CLI with -png_overlap 0 / 1 on the 300-frame clip (scripts/e2e.py prints the rates)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
Q="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra --no-e2e"
line() { python -c "
import sys, json
l = [x for x in sys.stdin.read().splitlines() if x.startswith('{')]
if not l: print('$1 NO BENCH LINE'); sys.exit()
j = json.loads(l[-1]); print('$1', j['value'], 'fps  ms/step', j['ms_per_step'])
"; }
{
for rep in 1 2 3; do
  timeout 300 $Q 2>/dev/null | line "4-arg, compute queue "
  timeout 300 $Q --png-async 2>/dev/null | line "4-arg, encoder queue "
done
for rep in 1 2; do
  timeout 300 $Q --structure 0 2>/dev/null | line "3-arg, compute queue "
  timeout 300 $Q --structure 0 --png-async 2>/dev/null | line "3-arg, encoder queue "
done
} 2>&1 | tee $O/png_async_ab_${1:-x}.log
echo "--- bin/fav_stylize, 600 frames file -> PNG, -png_overlap 0 / 1" | tee -a $O/png_async_ab_${1:-x}.log
for rep in 1 2; do
FAV_E2E_VARIANTS="4arg_compute_queue:-structure 1 -png_overlap 0;4arg_encoder_queue:-structure 1 -png_overlap 1;3arg_compute_queue:-png_overlap 0;3arg_encoder_queue:-png_overlap 1" timeout 900 python scripts/e2e.py 600 2>&1 | grep -v "thread CPU\|loop trace" | cut -c1-260 | tee -a $O/png_async_ab_${1:-x}.log
done
