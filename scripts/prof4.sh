#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_s4 -o s4 -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --structure 1 --lookahead ${LA:-0} > $O/prof_s4.log 2>&1
python - <<PY
import csv,glob
f = glob.glob("$O/prof_s4/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print(f"{r['Name'][:70]:70s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.2f} pct={r['Percentage']}")
PY
