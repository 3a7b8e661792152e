#!/bin/bash
# Round-3 GPU call C: PNG kernels (second version), odd sizes / channel counts, CLI, bench + rocprofv3
TAG=${1:-r03d}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
(timeout 600 python -m pytest tests/test_gpu_png.py -m gpu -q --timeout 300 2>&1 | tail -30) > $O/test_png_$TAG.log
(timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -k "not_powers or not_multiples or stream or fused_prior or mask or consistency or min_filter or config1 or temporal" 2>&1 | tail -30) > $O/test_par_$TAG.log
(timeout 900 python -m pytest tests/test_gpu_cli.py -m gpu -q --timeout 600 2>&1 | tail -30) > $O/test_cli_$TAG.log
timeout 1500 python bench.py --steps 40 --warmup 5 > $O/bench_$TAG.log 2> $O/bench_$TAG.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$TAG -o $TAG -- python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra --no-e2e > $O/prof_$TAG.log 2>&1
cd $R
for f in $(find $O/prof_$TAG -name "*kernel_stats.csv"); do cp $f $O/kernel_stats_$TAG.csv; done
rm -rf $O/prof_$TAG
echo "=== hostreg"; timeout 120 scripts/hostreg_bench.bin 2>&1 | tee $O/hostreg_$TAG.log
echo "=== png tests"; cat $O/test_png_$TAG.log
echo "=== parity tests"; cat $O/test_par_$TAG.log
echo "=== cli tests"; cat $O/test_cli_$TAG.log
echo "=== bench"; tail -5 $O/bench_$TAG.err; python - <<PY
import json
try:
    j=json.loads(open("$O/bench_$TAG.log").read().strip().splitlines()[-1])
    print("value", j["value"], "e2e", j.get("end_to_end_fps"))
    for k,v in j.get("e2e",{}).items():
        if isinstance(v,dict): print(k, {a:b for a,b in v.items() if a in ("fps","steady_state_fps","host_cpu_ms_per_frame","cpu_ms_per_frame_loaders","cpu_ms_per_frame_writers","cpu_ms_per_frame_main","wait_loader_s","wait_png_pool_s","png_writers","error","shader_clock")})
    print(json.dumps(j.get("roofline_hbm")))
except Exception as e: print("bench parse failed", e)
PY
python - <<PY
import csv
try:
    rows=list(csv.DictReader(open("$O/kernel_stats_$TAG.csv")))
    for r in rows[:24]:
        print(f"{r['Name'][:90]:90s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.2f} pct={r['Percentage']}")
except Exception as e: print("no stats", e)
PY
