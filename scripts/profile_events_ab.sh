#!/bin/bash
# what the bench's own per-convolution HIP events cost `value`: every 4th step (default) / every 8th / none
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
Q="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra --no-e2e"
line() { python -c "
import sys, json
l = [x for x in sys.stdin.read().splitlines() if x.startswith('{')]
j = json.loads(l[-1]); print('$1', j['value'], 'fps  ms/step', j['ms_per_step'])
"; }
for rep in 1 2 3; do
  timeout 300 $Q 2>/dev/null | line "events on every 4th step "
  timeout 300 $Q --profile-every 8 2>/dev/null | line "events on every 8th step "
  timeout 300 $Q --no-profile 2>/dev/null | line "no events                "
done 2>&1 | tee $O/profile_events_ab_${1:-x}.log
