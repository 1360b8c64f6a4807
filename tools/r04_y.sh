timeout 1200 python -m pytest tests/test_gpu_dist_world.py tests/test_gpu_fit.py -x -q -k "world or dist" 2>&1 | tail -5
python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C3', d['value'], d['phases_s'])"
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --force-dist 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C3 force-dist', d['value'], d['comm_nranks'], d['phases_s'])"
