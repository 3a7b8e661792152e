#!/bin/bash
# A/B of experimental variants of the F(4x4) kernel (FAV_W4_VAR of the diagnostic library; kernels_wino4.hip says what each bit does):
#   bash scripts/w4_var_ab.sh <tag> <var> [<var> ...]      alternating quick benches, whole-frame parity of the last variant, timelines
TAG=${1:-x}; shift
VARS=${@:-0 8}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
D=$R/fast-artistic-videos_amd/libfav_diag.so
Q="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra --no-e2e"
line() { python -c "
import sys, json
l = [x for x in sys.stdin.read().splitlines() if x.startswith('{')]
if not l: print('$1 NO BENCH LINE'); sys.exit()
j = json.loads(l[-1]); r = j.get('roofline', {})
print('$1', j['value'], 'fps  ms/step', j['ms_per_step'], ' F(4x4) us/launch', r.get('avg_launch_us'), 'frac', r.get('frac'), ' parity', j.get('parity'))
"; }
{
for rep in 1 2 3; do
  for V in $VARS; do FAV_AMD_LIB=$D FAV_W4_VAR=$V timeout 300 $Q 2>/dev/null | line "FAV_W4_VAR=$V "; done
done
for V in $VARS; do LAST=$V; done
echo "--- whole-frame parity against the oracle, FAV_W4_VAR=$LAST (1280x720, 4-argument mode)"
FAV_AMD_LIB=$D FAV_W4_VAR=$LAST timeout 600 python bench.py --steps 10 --warmup 2 --no-extra --no-e2e 2>/dev/null | line "FAV_W4_VAR=$LAST "
echo "--- in-kernel timelines (launch 25 of each instantiation: whole-unit launches)"
for V in $VARS; do FAV_AMD_LIB=$D FAV_W4_VAR=$V FAV_WINO_DBG=25 timeout 300 $Q 2>&1 | grep WINO4DBG | sed "s/^/FAV_W4_VAR=$V : /"; done
} 2>&1 | tee $O/w4_var_ab_${TAG}.log
