#!/bin/bash
TAG=${1:-r03m}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
(timeout 900 python -m pytest tests/test_gpu_png.py tests/test_gpu_cli.py -m gpu -q --timeout 600 -x 2>&1 | tail -25) > $O/test_png_$TAG.log
timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --quick-e2e > $O/bench_$TAG.log 2> $O/bench_$TAG.err
cat $O/test_png_$TAG.log; tail -3 $O/bench_$TAG.err
python - <<PY
import json
try:
    j=json.loads(open("$O/bench_$TAG.log").read().strip().splitlines()[-1])
    print("value", j["value"], "e2e", j.get("end_to_end_fps"), "png bytes", j["extra"].get("png_bytes_per_frame_in_timed_region"))
    print(j.get("e2e",{}).get("gpu_png"))
except Exception as e: print("bench parse failed", e)
PY
