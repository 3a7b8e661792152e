#!/bin/bash
# Round-3 GPU call E: option / two-stream / bf16-lookahead tests, CLI + VR CLI (GPU PNG), e2e legs with thread CPU report
TAG=${1:-r03e}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
(timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -k "stream_options or two_networks or bf16 or shared_device or lookahead" 2>&1 | tail -30) > $O/test_par_$TAG.log
(timeout 900 python -m pytest tests/test_gpu_cli.py tests/test_gpu_png.py -m gpu -q --timeout 600 2>&1 | tail -30) > $O/test_cli_$TAG.log
timeout 1500 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > $O/bench_$TAG.log 2> $O/bench_$TAG.err
python scripts/e2e.py > $O/e2e_$TAG.log 2>&1
echo "=== parity tests"; cat $O/test_par_$TAG.log
echo "=== cli tests"; cat $O/test_cli_$TAG.log
echo "=== bench"; tail -5 $O/bench_$TAG.err; python - <<PY
import json
try:
    j=json.loads(open("$O/bench_$TAG.log").read().strip().splitlines()[-1])
    print("value", j["value"], "e2e", j.get("end_to_end_fps"))
    for k,v in j.get("e2e",{}).items():
        if isinstance(v,dict): print(k, {a:b for a,b in v.items() if a in ("fps","steady_state_fps","host_cpu_ms_per_frame","cpu_ms_per_frame_loaders","cpu_ms_per_frame_writers","cpu_ms_per_frame_main","wait_loader_s","wait_png_pool_s","png_writers","error")})
except Exception as e: print("bench parse failed", e)
PY
echo "=== e2e.py"; tail -12 $O/e2e_$TAG.log
