cd $GRAFT_REPO_ROOT
for v in ${SIDES:-8 16 24 32 8 16}; do
  FAV_SIDE_CUS=$v timeout 300 python bench.py --structure 1 --steps 40 --warmup 5 --no-cpu-baseline --no-extra --no-e2e 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('SIDE_CUS=$v', j['value'], 'fps (4-arg, look-ahead)')"
done
