cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
(timeout 1400 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -12) > $O/test_w2.log; cat $O/test_w2.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2)
timeout 600 python bench.py --steps 40 --warmup 5 --no-extra --no-e2e 2>/dev/null | tail -1 > $O/bench_w2.log
python -c "
import json; j=json.loads(open('$O/bench_w2.log').read()); print('value', j['value'], 'parity', j.get('parity'), 'cpu', j.get('cpu_baseline'))"
