#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
FAV_FOLD_DBG=12 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-e2e 2>&1 >/dev/null | grep FOLDDBG
