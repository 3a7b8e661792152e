#!/bin/bash
# Round-3 GPU call A (parity): warp fixtures from the reference's own kernel, the new pinned-warp tests, config 2 as written,
# the 1504^2 VR faces, the free-running 1280x720 clip, a short bench.   usage: gpu_r3a.sh <tag>
TAG=${1:-r03a}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
(timeout 300 python tests/golden/make_golden.py --warp --out $O/golden 2>&1 | tail -8) > $O/golden_$TAG.log
cp $O/golden/warp_*.npz tests/golden/ 2>/dev/null
(timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -x -s -k "warp or fused_prior or stream or config2 or min_filter" 2>&1 | tail -30) > $O/test_$TAG.log
(timeout 600 python -m pytest tests/test_gpu_vr.py -m gpu -q --timeout 600 -x -s -k "config5" 2>&1 | tail -12) > $O/test_vr_$TAG.log
(timeout 300 python -m pytest tests/test_cpu_oracle.py -q -k "warp" 2>&1 | tail -5) > $O/test_cpu_$TAG.log
(timeout 700 python scripts/parity_clip.py --config 3 --frames 60 --no-teacher --budget-s 420 --out $O/parity_c3_freerun.json 2>&1 | tail -70) > $O/parity_c3_$TAG.log
timeout 600 python bench.py --steps 40 --warmup 5 > $O/bench_$TAG.log 2> $O/bench_$TAG.err
echo "=== golden"; cat $O/golden_$TAG.log
echo "=== tests"; cat $O/test_$TAG.log; cat $O/test_vr_$TAG.log; cat $O/test_cpu_$TAG.log
echo "=== c3"; tail -5 $O/parity_c3_$TAG.log
echo "=== bench"; tail -3 $O/bench_$TAG.err; cat $O/bench_$TAG.log | cut -c1-1500
