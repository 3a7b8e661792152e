#!/bin/bash
# Run on the GPU box (via gpurun): parity tests, smoke, bench, rocprofv3 kernel stats.  usage: gpu_check.sh <tag> [quick]
TAG=${1:-x}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
if [ "$2" != "quick" ]; then
  (python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -40) > $O/test_$TAG.log
  (python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3) > $O/smoke_$TAG.log
fi
(python bench.py --steps 40 --warmup 5 2>&1 | tail -2) > $O/bench_$TAG.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$TAG -o $TAG -- python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra --no-e2e > $O/prof_$TAG.log 2>&1
cd $R
for f in $(find $O/prof_$TAG -name "*kernel_stats.csv"); do cp $f $O/kernel_stats_$TAG.csv; done
cat $O/test_$TAG.log $O/smoke_$TAG.log 2>/dev/null | tail -25
cat $O/bench_$TAG.log
python - <<PY
import csv,sys
try:
    rows=list(csv.DictReader(open("$O/kernel_stats_$TAG.csv")))
    for r in rows[:24]:
        print(f"{r['Name'][:90]:90s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.2f} pct={r['Percentage']}")
except Exception as e: print("no stats", e)
PY
