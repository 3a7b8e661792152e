#!/bin/bash
# rocprofv3 kernel stats of the 4-argument (image-structure) mode as the timed configuration: where the side queues' time goes
TAG=${1:-x}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof4_$TAG -o $TAG -- python $R/bench.py --structure 1 --steps 40 --warmup 5 --no-cpu-baseline --no-extra --no-e2e > $O/prof4_$TAG.log 2>&1
cd $R
for f in $(find $O/prof4_$TAG -name "*kernel_stats.csv"); do cp $f $O/kernel_stats_4arg_$TAG.csv; done
rm -rf $O/prof4_$TAG
tail -1 $O/prof4_$TAG.log | cut -c1-400
python - <<PY
import csv
rows=list(csv.DictReader(open("$O/kernel_stats_4arg_$TAG.csv")))
for r in rows[:40]:
    print(f"{r['Name'][:100]:100s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.2f} tot_ms={float(r['TotalDurationNs'])/1e6:8.2f} pct={r['Percentage']}")
PY
