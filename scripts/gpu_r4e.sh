#!/bin/bash
# host-side launch times next to GPU execution times (same clock): kernel trace + HIP runtime trace of the CLI's 4-argument mode
TAG=${1:-r04e}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export FAV_E2E_VARIANTS="s4:-structure 1"
export FAV_E2E_WRAP="rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d $O/p_${TAG}_{name} -o t --"
timeout 600 python $R/scripts/e2e.py 200 > $O/e2e_$TAG.log 2>&1
cut -c1-200 $O/e2e_$TAG.log | grep -v "thread CPU" | head -5
find $O/p_${TAG}_s4 -name "*.csv" | head
for f in $(find $O/p_${TAG}_s4 -name "*kernel_trace.csv"); do cp $f $O/ktrace_$TAG.csv; done
for f in $(find $O/p_${TAG}_s4 -name "*hip_api_trace.csv"); do python - "$f" "$O/hiptrace_$TAG.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = [r for r in rows if r.get("Function", "") in ("hipLaunchKernel", "hipModuleLaunchKernel", "hipExtLaunchKernel", "hipMemcpyAsync", "hipEventQuery", "hipEventRecord", "hipStreamQuery")]
w = csv.DictWriter(open(sys.argv[2], "w"), fieldnames=list(rows[0].keys())); w.writeheader()
# keep a window of 12 frames' worth of calls from the middle of the run to stay small
t = sorted(int(r["Start_Timestamp"]) for r in keep)
lo = t[len(t) // 2]; hi = lo + 30_000_000
for r in keep:
    if lo <= int(r["Start_Timestamp"]) <= hi: w.writerow(r)
PY
done
rm -rf $O/p_${TAG}_s4
ls -la $O/*$TAG*
