#!/bin/bash
# where does the CLI's 4-argument mode lose against the in-HBM loop?  per-kernel durations of both under rocprofv3
TAG=${1:-r03z}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export FAV_E2E_VARIANTS="s4:-structure 1;s3:-structure 0"
export FAV_E2E_WRAP="rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_${TAG}_{name} -o t --"
timeout 600 python $R/scripts/e2e.py 300 > $O/e2e_$TAG.log 2>&1
cut -c1-200 $O/e2e_$TAG.log | grep -v "thread CPU"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_${TAG}_bench4 -o t -- python $R/bench.py --structure 1 --steps 60 --warmup 10 --no-cpu-baseline --no-extra --no-e2e > $O/bench4_$TAG.log 2>&1
tail -1 $O/bench4_$TAG.log | cut -c1-200
for n in s4 s3 bench4; do
  f=$(find $O/p_${TAG}_$n -name "*kernel_stats.csv" | head -1); cp $f $O/kstats_${TAG}_$n.csv
  t=$(find $O/p_${TAG}_$n -name "*kernel_trace.csv" | head -1); cp $t $O/ktrace_${TAG}_$n.csv
  rm -rf $O/p_${TAG}_$n
done
ls -la $O/ktrace_${TAG}_*.csv
