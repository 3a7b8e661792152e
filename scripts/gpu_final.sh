#!/bin/bash
# what the driver runs at round end: the GPU suite, smoke(), the default bench line
TAG=${1:-final}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
(timeout 1500 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -15) > $O/test_$TAG.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3) > $O/smoke_$TAG.log
S=$(date +%s); timeout 900 python bench.py > $O/bench_$TAG.log 2> $O/bench_$TAG.err; E=$(date +%s)
echo "bench wall seconds: $((E-S))" >> $O/bench_$TAG.err
cat $O/test_$TAG.log $O/smoke_$TAG.log; tail -3 $O/bench_$TAG.err
python - <<PY
import json
j=json.loads(open("$O/bench_$TAG.log").read().strip().splitlines()[-1])
print("value", j["value"], "end_to_end_fps", j.get("end_to_end_fps"), "frac", j["roofline"]["frac"], "extra", j["extra"])
for k,v in j.get("e2e",{}).items():
    if isinstance(v,dict): print(k, {a:b for a,b in v.items() if a in ("fps","steady_state_fps","host_cpu_ms_per_frame","frames","error")})
PY
