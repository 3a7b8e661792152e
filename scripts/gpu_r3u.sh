#!/bin/bash
TAG=${1:-r03u}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$TAG -o $TAG -- python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra --no-e2e > $O/prof_$TAG.log 2>&1
cd $R
for f in $(find $O/prof_$TAG -name "*kernel_stats.csv"); do cp $f $O/kernel_stats_$TAG.csv; done
rm -rf $O/prof_$TAG
bash scripts/gpu_pmc.sh $TAG > $O/pmc_$TAG.log 2>&1
rm -rf $O/pmc_${TAG}_*/
tail -30 $O/pmc_$TAG.log | cut -c1-400
python - <<PY
import csv
rows=list(csv.DictReader(open("$O/kernel_stats_$TAG.csv")))
for r in rows[:20]:
    print(f"{r['Name'][:90]:90s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.2f} pct={r['Percentage']}")
PY
