#!/bin/bash
# needs a tuning build: make -C fast-artistic-videos_amd clean all CXXEXTRA=-DFAV_ABLATIONS (the ablation instances give wrong results by design)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for a in ${ABLS:-0 1 2 3 4 5}; do
  echo -n "ABL=$a "; FAV_ABL=$a python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['avg_launch_us'], d['roofline']['achieved'], d['ms_per_step'])"
done
