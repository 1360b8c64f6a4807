# quick GPU check: tests + smoke + C2/C3 bench lines   (EVID=tag)
O=gpurun_out/${EVID:-r05a}; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q --durations=8 > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
python bench.py --config C2 --steps 10 --warmup 3 --no-cpu-baseline 2>$O/bench_C2.err | tail -1 > $O/bench_C2.json
python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>$O/bench_C3.err | tail -1 > $O/bench_C3_10steps.json
tail -4 $O/gpu_tests.log; tail -2 $O/smoke.log
for f in $O/bench_*.json; do python -c "
import json
d=json.load(open('$f')); r=d['roofline']; print('$f', d['value'], r['kernel'][:40], r['frac'], r.get('fit_frac'), d['phases_s'])"; done
