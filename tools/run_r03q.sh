O=gpurun_out/r03q; mkdir -p $O
python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_C3_10steps_box2.json
python bench.py --config C4 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_C4_box2.json
python bench.py --config C5 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_C5_box2.json
python bench.py --config C2 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_C2_box2.json
for f in $O/*.json; do python -c "
import json; d=json.load(open('$f')); print('$f', d['value'], d['roofline']['frac'])"; done
