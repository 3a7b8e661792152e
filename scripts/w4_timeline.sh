#!/bin/bash
# in-kernel timelines of the F(4x4) launches (diagnostic library): bash scripts/w4_timeline.sh [<launch number> ...]
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for n in ${@:-25 26}; do
  FAV_AMD_LIB=$R/fast-artistic-videos_amd/libfav_diag.so FAV_WINO_DBG=$n timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra --no-e2e 2>&1 | grep WINO4DBG | sed "s/^/launch $n: /"
done
