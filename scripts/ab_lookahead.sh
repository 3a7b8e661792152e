cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -q -x -k "stream or cli or stylize or lookahead or 1280 or config1" 2>&1 | tail -3
for v in "--lookahead 0" "--lookahead 1" "--lookahead 0" "--lookahead 1"; do
  timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra --no-e2e $v 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('[$v]', j['value'], 'fps')"
done
