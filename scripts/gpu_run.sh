#!/bin/bash
# The ONE runner for GPU-box sessions (replaces the per-experiment gpu_r*.sh scripts of rounds 1-3):
#   gpurun --timeout 900 -- 'bash scripts/gpu_run.sh <tag> <step> [<step> ...]'
# Every step writes under gpurun_out/ with the tag in the name; summaries worth keeping are copied into profiles/ by hand.
# Steps:
#   test[:<pytest -k expression>]    pytest -m gpu (whole suite, or a selection)
#   testfile:<file>[:<k>]            one test file
#   smoke                            __graft_entry__.smoke()
#   bench[:<extra bench.py flags>]   the driver's default command line (+ flags, '+' for spaces)
#   quick                            in-HBM rate only (no CPU baseline, no e2e, no extras); quick3: the 3-argument checker mode (--structure 0)
#   prof                             rocprofv3 --kernel-trace --stats of the timed configuration (4-argument mode since round 6) -> <tag>_kernel_stats.csv
#   prof4 / prof3                    the same with --structure 1 / --structure 0
#   pmc                              scripts/gpu_pmc.sh <tag>  (separate --pmc passes)
#   gaps                             scripts/gaps.sh
#   ab:<env=val>[,<env=val>...][;<env=val>...]   quick bench alternating default / each ';'-separated set of environment settings, two rounds
#   ab4:... / ab3:...                the same with --structure 1 / --structure 0 spelled out
#   ablib:<lib.so>[,<lib.so>...]     scripts/ab_lib.sh (alternating library builds)
#   e2e[:<frames>]                   scripts/e2e.py (the CLI's own timing breakdown)
#   c3dump                           GPU side of the 300-frame config-3 free-running parity run (contractive checkpoint)
#   parity:<config>:<frames>[:gain]  scripts/parity_clip.py on the box (teacher-forced + free-running)
#   vr[:<flags>]                     scripts/vr_bench.py (e.g. vr:--arch+2x)
#   wide[:<flags>]                   scripts/wide_bench.py (a checkpoint with more filters; e.g. wide:--arch+1.5x)
#   profwide[:<flags>]               rocprofv3 --kernel-trace --stats of scripts/wide_bench.py -> <tag>_wide_kernel_stats.csv
#   sh:<command>                     anything else ('+' for spaces)
TAG=${1:-x}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
QUICK="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra --no-e2e"
summ() { python -c "
import sys, json
l = [x for x in sys.stdin.read().splitlines() if x.startswith('{')]
if not l: print('NO BENCH LINE'); sys.exit()
j = json.loads(l[-1]); r = j.get('roofline', {})
print('$1', j['value'], 'fps  ms/step', j['ms_per_step'], ' wino us', r.get('avg_launch_us'), 'frac', r.get('frac'), ' per-kernel', r.get('per_kernel_ms_tflops'))
e = j.get('extra', {}); print('   extra', {k: v for k, v in e.items() if 'frames_per_s' in k})
if 'e2e' in j: print('   e2e', {k: (v.get('fps'), v.get('png_mismatch_frames'), v.get('png_missing_frames')) for k, v in j['e2e'].items() if isinstance(v, dict) and 'fps' in v}, 'mismatch total', j.get('png_mismatch_frames'), 'checked', j.get('png_checked_frames'))
"; }
for STEP in "$@"; do
  K=${STEP%%:*}; A=""; [[ "$STEP" == *:* ]] && A=${STEP#*:}; A=${A//+/ }
  echo "=== [$TAG] $K $A"
  case $K in
    test)     if [ -n "$A" ]; then timeout 3000 python -m pytest tests -m gpu -x -q --timeout 900 -k "$A" 2>&1 | tee $O/test_${TAG}.log | tail -15
              else timeout 3000 python -m pytest tests -m gpu -x -q --timeout 900 2>&1 | tee $O/test_${TAG}.log | tail -15; fi ;;
    testfile) F=${A%%:*}; KX=""; [[ "$A" == *:* ]] && KX=${A#*:}
              timeout 3000 python -m pytest tests/$F -m gpu -x -q -s --timeout 900 ${KX:+-k "$KX"} 2>&1 | tee $O/test_${TAG}_${F%.py}.log | tail -25 ;;
    smoke)    timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tee $O/smoke_${TAG}.log | tail -3 ;;
    bench)    timeout 1500 python bench.py $A > $O/bench_${TAG}.log 2> $O/bench_${TAG}.err; summ bench < $O/bench_${TAG}.log; tail -3 $O/bench_${TAG}.err ;;
    quick|quick3) X=""; [ $K = quick3 ] && X="--structure 0"
              for i in 1 2; do timeout 300 $QUICK $X $A 2>/dev/null | tee -a $O/${K}_${TAG}.log | summ $K; done ;;
    prof|prof4|prof3)
              X=""; [ $K = prof4 ] && X="--structure 1"; [ $K = prof3 ] && X="--structure 0"
              (cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_$K -o p -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-e2e $X > $O/prof_${TAG}_$K.log 2>&1)
              cp $(ls $O/prof_${TAG}_$K/*kernel_stats.csv | head -1) $O/${TAG}_${K}_kernel_stats.csv 2>/dev/null
              python - <<PY
import csv
rows = list(csv.DictReader(open("$O/${TAG}_${K}_kernel_stats.csv")))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:28]:
    print("%-90s calls %5s avg %9.2f us  %5.1f %%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
              ;;
    pmc)      bash scripts/gpu_pmc.sh $TAG 2>&1 | tail -40 ;;
    gaps)     bash scripts/gaps.sh 2>&1 | tail -70 ;;
    ab|ab4|ab3) X=""; [ $K = ab4 ] && X="--structure 1"; [ $K = ab3 ] && X="--structure 0"
              for rep in 1 2; do
                timeout 300 $QUICK $X 2>/dev/null | summ "default "
                for V in ${A//;/ }; do (export ${V//,/ }; timeout 300 $QUICK $X 2>/dev/null | summ "$V "); done
              done 2>&1 | tee $O/${K}_${TAG}.log ;;
    ablib)    bash scripts/ab_lib.sh ${A//,/ } 2>&1 | tee $O/ablib_${TAG}.log ;;
    e2e)      timeout 900 python scripts/e2e.py ${A:-300} 2>&1 | tee $O/e2e_${TAG}.log | tail -12 ;;
    c3dump)   timeout 600 python scripts/parity_clip.py --config 3 --frames 300 --gain 0.05 --gpu-dump $O/c3_contractive_gpu.npz 2>&1 | tail -3 ;;
    parity)   IFS=: read C N G <<< "$A"
              timeout 2400 python scripts/parity_clip.py --config $C --frames $N --gain ${G:-1.0} --out $O/parity_c${C}_${TAG}.json 2>&1 | tail -2 ;;
    vr)       timeout 900 python scripts/vr_bench.py $A 2>&1 | tee -a $O/vr_${TAG}.log | tail -5 ;;
    wide)     timeout 900 python scripts/wide_bench.py $A 2>&1 | tee -a $O/wide_${TAG}.log | tail -5 ;;
    profwide) (cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_wide -o p -- python $R/scripts/wide_bench.py --steps 10 --warmup 3 $A > $O/prof_${TAG}_wide.log 2>&1)
              cp $(ls $O/prof_${TAG}_wide/*kernel_stats.csv | head -1) $O/${TAG}_wide_kernel_stats.csv 2>/dev/null
              python - <<PY
import csv
rows = list(csv.DictReader(open("$O/${TAG}_wide_kernel_stats.csv")))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:24]:
    print("%-90s calls %5s avg %9.2f us  %5.1f %%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
              tail -1 $O/prof_${TAG}_wide.log | cut -c1-600 ;;
    sh)       bash -c "$A" 2>&1 | tail -40 ;;
    *)        echo "unknown step $K" ;;
  esac
done
