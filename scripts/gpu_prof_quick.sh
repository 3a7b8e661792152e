#!/bin/bash
# quick look: rocprofv3 kernel stats of the timed configuration only (no tests, no CPU legs)
TAG=${1:-x}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/profq_$TAG -o $TAG -- python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra --no-e2e > $O/profq_$TAG.log 2>&1
cd $R
for f in $(find $O/profq_$TAG -name "*kernel_stats.csv"); do cp $f $O/kernel_stats_$TAG.csv; done
rm -rf $O/profq_$TAG
grep '^{' $O/profq_$TAG.log | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('fps', j['value'], j['roofline']['per_kernel_ms_tflops'])"
python - <<PY
import csv
rows=list(csv.DictReader(open("$O/kernel_stats_$TAG.csv")))
for r in rows[:${2:-20}]:
    print(f"{r['Name'][:100]:100s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.2f} pct={r['Percentage']}")
PY
