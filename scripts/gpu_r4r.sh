#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export FAV_E2E_VARIANTS="s3:-structure 0 FAV_LOOP_TRACE=2"
timeout 600 python scripts/e2e.py 300 2>&1 | cut -c1-260
