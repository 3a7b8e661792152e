#!/bin/bash
# in-kernel timeline of the last layer (diagnostic library): bash scripts/fold_timeline.sh [<launch number>]
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
FAV_AMD_LIB=$R/fast-artistic-videos_amd/libfav_diag.so FAV_FOLD_DBG=${1:-20} timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra --no-e2e 2>&1 | grep -i "FOLD"
