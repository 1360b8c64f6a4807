timeout 900 python -m pytest tests/test_gpu_level1.py tests/test_gpu_fit.py tests/test_gpu_fit_capi.py -x -q -k "solveforc or lambda or probe or fit" 2>&1 | tail -3
python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('C3', d['value'], d['phases_s']['lambda'])
for e in d['other_kernels']:
    if 'sf_probe' in e['kernel'] or 'deriv_rows' in e['kernel']: print(e['kernel'][:30], e['achieved'], e['avg_launch_us'])"
python bench.py --config C2 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('C2', d['value'], d['phases_s'])"
