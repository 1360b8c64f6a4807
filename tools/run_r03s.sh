O=gpurun_out/r03s; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $O/gpu_tests.log
python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_C3.json
python bench.py --config C4 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_C4.json
python bench.py --config C2 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_C2.json
for f in $O/bench_*.json; do python -c "
import json; d=json.load(open('$f')); print('$f', d['value'], d['roofline']['frac'], [ (o['kernel'][:12], o.get('avg_launch_us')) for o in d['other_kernels'][:3]])"; done
