#!/bin/bash
# kernel-to-kernel gaps of the timed configuration (rocprofv3 kernel trace): which boundaries cost more than the ~2 us minimum?
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/gaps -o g -- python $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extra --no-e2e > $O/gaps.log 2>&1
cd $R
python - <<PY
import csv, glob, re, collections
f = glob.glob("$O/gaps/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
def short(n): return re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "").replace("void ", "").replace("fav::", ""))[:40]
# last full frame: find the last prep_input_kernel
idx = [i for i, r in enumerate(rows) if "prep_input" in r["Kernel_Name"]]
a, b = idx[-3], idx[-2]
tot_gap = 0; tot_k = 0
agg = collections.defaultdict(list)
for i in range(a, b):
    r, n = rows[i], rows[i + 1]
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    gap = (int(n["Start_Timestamp"]) - int(r["End_Timestamp"])) / 1e3
    tot_gap += gap; tot_k += dur
    agg[(short(r["Kernel_Name"]), short(n["Kernel_Name"]))].append(gap)
    print(f"{short(r['Kernel_Name']):42s} {dur:8.2f} us   gap to next {gap:6.2f}")
print("frame: kernels %.1f us, gaps %.1f us over %d launches" % (tot_k, tot_gap, b - a))
PY
