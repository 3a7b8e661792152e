#!/bin/bash
# what the HIP events of the roofline block cost in the timed region: every step / every 4th step (default) / none
cd ${GRAFT_REPO_ROOT:-.}
for v in "--profile-every 1" "" "--no-profile" "--profile-every 1" "" "--no-profile"; do
  timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra --no-e2e $v 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); r = j.get('roofline', {})
print('[$v]', j['value'], 'fps  conv stack', r.get('conv_stack_ms_per_frame'), 'ms  dominant', r.get('avg_launch_us'), 'us  frac', r.get('frac'), r.get('launches'))"
done
