#!/bin/bash
# Round-3 GPU call N: full suite, smoke, default bench line, rocprofv3 kernel stats, PMC HBM passes, 60-frame free-running clip, VR bench
TAG=${1:-r03n}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
(timeout 1500 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -40) > $O/test_$TAG.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3) > $O/smoke_$TAG.log
timeout 900 python bench.py > $O/bench_$TAG.log 2> $O/bench_$TAG.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$TAG -o $TAG -- python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra --no-e2e > $O/prof_$TAG.log 2>&1
cd $R
for f in $(find $O/prof_$TAG -name "*kernel_stats.csv"); do cp $f $O/kernel_stats_$TAG.csv; done
rm -rf $O/prof_$TAG
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_${TAG}_$C -o p -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra --no-e2e > $O/pmc_${TAG}_$C.log 2>&1
done
cd $R
python - <<PY
import csv, glob, collections, json, re
out = {}
for f in sorted(glob.glob("$O/pmc_${TAG}_*/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        for c, v in d.items():
            out.setdefault(k, {})[c] = {"mean": sum(v) / len(v), "n": len(v)}
json.dump(out, open("$O/pmc_hbm_$TAG.json", "w"), indent=1, sort_keys=True)
for k in sorted(out):
    if "FETCH_SIZE" in out[k] and "WRITE_SIZE" in out[k]:
        print(k[:70], "FETCH_KB %.0f WRITE_KB %.0f" % (out[k]["FETCH_SIZE"]["mean"], out[k]["WRITE_SIZE"]["mean"]))
PY
rm -rf $O/pmc_${TAG}_FETCH_SIZE $O/pmc_${TAG}_WRITE_SIZE
(timeout 500 python scripts/parity_clip.py --config 3 --frames 60 --no-teacher --budget-s 330 --out $O/parity_c3_freerun.json 2>&1 | tail -4) > $O/parity_c3_$TAG.log
(timeout 300 python scripts/vr_bench.py 2>&1 | tail -5) > $O/vr_bench_$TAG.log
echo "=== tests"; cat $O/test_$TAG.log; cat $O/smoke_$TAG.log
echo "=== bench"; tail -3 $O/bench_$TAG.err; cut -c1-600 $O/bench_$TAG.log
echo "=== c3"; cat $O/parity_c3_$TAG.log | cut -c1-600
echo "=== vr"; cat $O/vr_bench_$TAG.log
python - <<PY
import csv
try:
    rows=list(csv.DictReader(open("$O/kernel_stats_$TAG.csv")))
    for r in rows[:22]:
        print(f"{r['Name'][:90]:90s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.2f} pct={r['Percentage']}")
except Exception as e: print("no stats", e)
PY
