#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 600 -k "lookahead or stream_vs" 2>&1 | tail -2
for cfg in "--structure 0 --lookahead 0" "--structure 0 --lookahead 1" "--structure 1 --lookahead 0" "--structure 1 --lookahead 1"; do
  echo -n "$cfg : "; python bench.py --steps 30 --warmup 4 --no-cpu-baseline $cfg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done
