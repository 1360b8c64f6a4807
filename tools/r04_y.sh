O=gpurun_out/r04h; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_dist_world.py tests/test_gpu_level1.py -x -q -k "world or eigen" 2>&1 | tail -5
for cfg in "5000 10" "20000 20"; do BIGKRLS_VERBOSE=1 timeout 300 python tools/eig_once.py $cfg 2>&1 | grep "^rep\|back-transform stage 2" | tail -3; done
python bench.py --config C4 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C4', d['value'], d['phases_s'])"
python bench.py --config C4 --steps 3 --warmup 1 --no-cpu-baseline --force-dist 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C4 force-dist', d['value'], d['comm_nranks'], d['phases_s'])"
