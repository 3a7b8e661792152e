#!/bin/bash
# A/B of several builds of libfav on ONE box: scripts/ab_lib.sh <a.so> <b.so> ... (alternates them with the in-tree library, two rounds)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; ARGS=(); for a in "$@"; do ARGS+=("$(readlink -f $a)"); done; set -- "${ARGS[@]}"
cd $R/fast-artistic-videos_amd; cp libfav.so /tmp/libfav_cur.so
for rep in 1 2; do for v in /tmp/libfav_cur.so "$@"; do
  cp $v libfav.so
  (cd $R && timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra --no-e2e 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); r = j['roofline']
k = r['per_kernel_ms_tflops']
print('$v', j['value'], 'fps  wino us', r.get('avg_launch_us'), ' join/no-join', r.get('avg_launch_us_with_join'), r.get('avg_launch_us_without_join'), {i: k.get(i) for i in ('728', '729', '528', '529') if i in k})")
done; done
cp /tmp/libfav_cur.so libfav.so
