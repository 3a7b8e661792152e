#!/bin/bash
# Round-2 GPU check (run on the GPU box through gpurun): parity suite, smoke, full bench line, A/B of tuning switches, rocprofv3 stats.
# usage: gpu_r2.sh <tag> [notests] ; extra A/B env settings in $AB (space-separated VAR=VAL items, each run separately)
TAG=${1:-x}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
if [ "$2" != "notests" ]; then
  (timeout 1200 python -m pytest tests -m gpu -q --timeout 900 -x 2>&1 | tail -30) > $O/test_$TAG.log
  (timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3) > $O/smoke_$TAG.log
fi
timeout 900 python bench.py --steps 40 --warmup 5 --force-dist > $O/bench_$TAG.log 2> $O/bench_$TAG.err; tail -5 $O/bench_$TAG.err
for kv in $AB; do
  for rep in 1 2; do
    (env $kv timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra --no-e2e 2>&1 | tail -1 | python -c "
import sys, json
try:
    j = json.loads(sys.stdin.read()); r = j['roofline']
    print('$kv', j['value'], 'fps', 'dom_us', r['avg_launch_us'], 'frac', r['frac'], 'conv_ms', r['conv_stack_ms_per_frame'], r['per_kernel_ms_tflops'])
except Exception as e: print('$kv', 'FAILED', e)") >> $O/ab_$TAG.log
  done
done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$TAG -o $TAG -- python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra --no-e2e > $O/prof_$TAG.log 2>&1
cd $R
for f in $(find $O/prof_$TAG -name "*kernel_stats.csv"); do cp $f $O/kernel_stats_$TAG.csv; done
rm -rf $O/prof_$TAG
echo "=== tests"; cat $O/test_$TAG.log $O/smoke_$TAG.log 2>/dev/null | tail -25
echo "=== bench"; cat $O/bench_$TAG.log
echo "=== ab"; cat $O/ab_$TAG.log 2>/dev/null
python - <<PY
import csv
try:
    rows=list(csv.DictReader(open("$O/kernel_stats_$TAG.csv")))
    for r in rows[:20]:
        print(f"{r['Name'][:90]:90s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.2f} pct={r['Percentage']}")
except Exception as e: print("no stats", e)
PY
